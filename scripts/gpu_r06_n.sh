#!/bin/bash
# round 6, GPU pass N: default resident set + stagger rules of the 32768 / 16384-sample row kernels: parity, then the defaults against one workgroup per row
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06n; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "long_rows_walked or fourstep_1d or config2" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
P="timeout 300 python scripts/prof.py call"
{
for REP in 1 2; do
for E in "" "XRFTHIP_FASTR_GRID=0 XRFTHIP_FASTR_STAGGER=0"; do
echo "== ${E:-defaults}"
env $E $P fft 2048,32768,float32 dim=x --reps 20
env $E $P dft 2048,32768,float32 dim=x --reps 20
env $E $P power_spectrum 2048,32768,float32 dim=x detrend=linear window=hann --reps 20
env $E $P fft 4096,16384,float32 dim=x --reps 20
env $E $P dft 4096,16384,float32 dim=x --reps 20
env $E $P power_spectrum 4096,16384,float32 dim=x detrend=linear window=hann --reps 20
done; done
} > $O/rows.txt 2>&1
grep -v "amdgpu\|Warn" $O/rows.txt | grep "GFFT\|Error\|==" | cut -c1-40,95-250
