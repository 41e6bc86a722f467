#!/bin/bash
# round 6, GPU pass L: inverse transforms of grid lengths on the table kernels (complex columns of any count in fastm_yonly, irfft rows in fastm_xonly)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06l; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "inverse or short_contiguous or one_axis_not_contiguous" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
P="timeout 300 python scripts/prof.py call"
{
$P ifft 64,1440,361,complex128 dim=y,x real_dim=x
$P ifft 64,1440,720,complex128 dim=y,x
$P ifft 64,1440,361,complex64 dim=y,x real_dim=x
$P ifft 64,1440,720,complex64 dim=y,x
$P ifft 92160,361,complex128 dim=x real_dim=x
$P ifft 64,1440,361,complex128 dim=y
$P ifft 32,2000,1001,complex128 dim=y,x real_dim=x
$P ifft 16,2160,2161,complex64 dim=y,x real_dim=x
} > $O/inv.txt 2>&1
grep -v "amdgpu\|Warn" $O/inv.txt | grep "GFFT\|Error\|==" | cut -c1-250
