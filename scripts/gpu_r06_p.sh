#!/bin/bash
# round 6, GPU pass P: resident set + stagger for real rows of 8192 samples (fastr2_kernel<16, 8>, four workgroups per CU)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06p; mkdir -p $O
export TMPDIR=/tmp
P="timeout 300 python scripts/prof.py call"
{
for REP in 1 2; do
for CFG in "0 0" "1024 0" "1024 513" "1024 769" "1024 514" "1024 1025" "768 0" "768 769"; do
set -- $CFG
export XRFTHIP_FASTR_GRID=$1 XRFTHIP_FASTR_STAGGER=$2
echo "== grid=$1 stagger=$2"
$P fft 8192,8192,float32 dim=x --reps 20
$P dft 8192,8192,float32 dim=x --reps 20
$P power_spectrum 8192,8192,float32 dim=x detrend=linear window=hann --reps 20
done; done
} > $O/rows.txt 2>&1
