#!/bin/bash
# round 6, GPU pass K: fasts_power_kernel with the next slab's loads in flight beside the stores (a resident set walking the slabs) against one workgroup per slab
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06k; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "walked" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
P="timeout 300 python scripts/prof.py call"
{
for G in 0 256 -1 0 256; do
echo "== XRFTHIP_FASTS_GRID=$G"
export XRFTHIP_FASTS_GRID=$G
$P power_spectrum 4096,256,256,float32 dim=y,x detrend=linear window=hann
$P power_spectrum 4096,256,256,float32 dim=y,x
$P isotropic_power_spectrum 4096,256,256,float32 dim=y,x detrend=linear window=hann
done
for G in 0 512 1024; do
echo "== XRFTHIP_FASTS_GRID=$G"
export XRFTHIP_FASTS_GRID=$G
$P power_spectrum 8192,256,128,float32 dim=y,x detrend=linear window=hann
$P power_spectrum 16384,128,128,float32 dim=y,x detrend=linear window=hann
done
for G in 0 2048 3072; do
echo "== XRFTHIP_FASTS_GRID=$G"
export XRFTHIP_FASTS_GRID=$G
$P power_spectrum 16384,128,128,float32 dim=y,x detrend=linear window=hann
$P power_spectrum 65536,64,64,float32 dim=y,x detrend=linear window=hann
done
} > $O/fasts.txt 2>&1
grep -v "amdgpu\|Warn" $O/fasts.txt | grep "GFFT\|Error\|==" | cut -c1-200
