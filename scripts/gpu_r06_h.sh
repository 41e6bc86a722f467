#!/bin/bash
# round 6, GPU pass H: the c2r row pass with the partners through the LDS: parity, timings
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "half_spectra or complex_rows or complex_slabs or inverse" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
P="timeout 300 python scripts/prof.py call"
{
$P ifft 16,4096,2049,complex64 dim=y real_dim=x
$P ifft 64,2048,1025,complex64 dim=y real_dim=x
$P ifft 64,1024,513,complex64 dim=y real_dim=x
$P ifft 16384,2049,complex64 dim=x real_dim=x
$P ifft 131072,513,complex64 dim=x real_dim=x
} > $O/irfft.txt 2>&1
grep -v "amdgpu\|Warn" $O/irfft.txt | grep "GFFT\|Error" | cut -c1-220
