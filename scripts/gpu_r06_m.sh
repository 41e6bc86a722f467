#!/bin/bash
# round 6, GPU pass M: resident sets + start stagger for the register-resident row kernels below 65536 samples (fastr2_kernel); every setting three times, interleaved
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06m; mkdir -p $O
export TMPDIR=/tmp
P="timeout 300 python scripts/prof.py call"
{
for REP in 1 2 3; do
for S in 0 769 770 1025 1026 1282; do
export XRFTHIP_FASTR_GRID=256 XRFTHIP_FASTR_STAGGER=$S
echo "== 32768: grid=256 stagger=$S"
$P fft 2048,32768,float32 dim=x --reps 20
$P dft 2048,32768,float32 dim=x --reps 20
$P power_spectrum 2048,32768,float32 dim=x detrend=linear window=hann --reps 20
done
for S in 0 513 769 770 1025 1026; do
export XRFTHIP_FASTR_GRID=512 XRFTHIP_FASTR_STAGGER=$S
echo "== 16384: grid=512 stagger=$S"
$P fft 4096,16384,float32 dim=x --reps 20
$P power_spectrum 4096,16384,float32 dim=x detrend=linear window=hann --reps 20
done
export XRFTHIP_FASTR_GRID=0 XRFTHIP_FASTR_STAGGER=0
echo "== 16384: grid=0 stagger=0"
$P fft 4096,16384,float32 dim=x --reps 20
$P power_spectrum 4096,16384,float32 dim=x detrend=linear window=hann --reps 20
echo "== 32768: grid=0 stagger=0"
$P fft 2048,32768,float32 dim=x --reps 20
$P power_spectrum 2048,32768,float32 dim=x detrend=linear window=hann --reps 20
done
} > $O/rows.txt 2>&1
grep -v "amdgpu\|Warn" $O/rows.txt | grep "GFFT\|Error\|==" | cut -c1-40,95-250
