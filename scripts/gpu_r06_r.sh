#!/bin/bash
# round 6, GPU pass R: ONE long complex sequence per batch entry (2^16 .. 2^20 points) through the two passes of fasty_c2c.h on the [n / 256][256] view (four-step form):
# parity, then against the generic four-step passes (XRFTHIP_FASTYC=0)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06r; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "complex_rows_in_one_pass or complex_slabs or inverse_transforms" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
P="timeout 300 python scripts/prof.py call"
{
for E in "" "XRFTHIP_FASTYC=0"; do
echo "== ${E:-four-step on fasty_c2c}"
env $E $P ifft 1024,65536,complex64 dim=x --reps 10
env $E $P fft 1024,65536,complex64 dim=x --reps 10
env $E $P power_spectrum 1024,65536,complex64 dim=x --reps 10
env $E $P ifft 256,262144,complex64 dim=x --reps 10
env $E $P ifft 64,1048576,complex64 dim=x --reps 10
done
} > $O/fs.txt 2>&1
grep -v "amdgpu\|Warn" $O/fs.txt | grep "GFFT\|Error\|==" | cut -c1-250
