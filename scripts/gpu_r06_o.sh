#!/bin/bash
# round 6, GPU pass O: a start stagger for the resident set of 256 x 256 slab workgroups (fasts_power_kernel)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06o; mkdir -p $O
export TMPDIR=/tmp
P="timeout 300 python scripts/prof.py call"
{
for REP in 1 2; do
for S in 0 513 514 515 516 769 770 771 1025 1026 1027; do
export XRFTHIP_FASTS_STAGGER=$S
echo "== stagger=$S"
$P power_spectrum 4096,256,256,float32 dim=y,x detrend=linear window=hann --reps 20
$P power_spectrum 4096,256,256,float32 dim=y,x --reps 20
done
export XRFTHIP_FASTS_STAGGER=0 XRFTHIP_FASTS_GRID=0
echo "== one workgroup per slab"
$P power_spectrum 4096,256,256,float32 dim=y,x detrend=linear window=hann --reps 20
$P power_spectrum 4096,256,256,float32 dim=y,x --reps 20
unset XRFTHIP_FASTS_GRID
done
} > $O/fasts.txt 2>&1
grep -v "amdgpu\|Warn" $O/fasts.txt | grep "GFFT\|Error\|==" | cut -c1-40,95-250
