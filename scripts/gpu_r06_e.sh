#!/bin/bash
# round 6, GPU pass E: the complex-row and complex two-pass kernels: parity tests, then the inverse-transform timings of profiles/r05_inverse.txt again
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "complex_rows or complex_slabs or inverse" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
P="timeout 300 python scripts/prof.py call"
{
$P ifft 16384,4096,complex64 dim=x
$P ifft 16384,4096,complex64 dim=x --env XRFTHIP_CROWS=0
$P ifft 32768,2048,complex64 dim=x --env XRFTHIP_CROWS=0
$P ifft 131072,1024,complex64 dim=x --env XRFTHIP_CROWS=0
$P ifft 8192,8192,complex64 dim=x
$P ifft 4096,16384,complex64 dim=x
$P ifft 32768,2048,complex64 dim=x
$P fft 16384,4096,complex64 dim=x
$P ifft 16,4096,4096,complex64 dim=y,x
$P ifft 64,2048,2048,complex64 dim=y,x
$P ifft 64,1024,1024,complex64 dim=y,x
$P fft 16,4096,4096,complex64 dim=y,x
$P power_spectrum 16,4096,4096,complex64 dim=y,x
$P ifft 131072,1024,complex64 dim=x
$P ifft 1024,65536,complex64 dim=x
$P ifft 64,1440,720,complex128 dim=y,x
} > $O/inverse.txt 2>&1
grep -v "amdgpu\|Warn" $O/inverse.txt | cut -c1-260
