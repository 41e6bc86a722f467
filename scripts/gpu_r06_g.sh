#!/bin/bash
# round 6, GPU pass G: the split library + the c2r pipeline: the whole GPU suite (timed), smoke, then the irfft timings
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06g; mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s); timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $O/pytest_gpu.txt 2>&1; echo "wall $(( $(date +%s) - T0 )) s" >> $O/pytest_gpu.txt; tail -18 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
P="timeout 300 python scripts/prof.py call"
{
$P ifft 16,4096,2049,complex64 dim=y real_dim=x
$P ifft 64,2048,1025,complex64 dim=y real_dim=x
$P ifft 64,1024,513,complex64 dim=y real_dim=x
$P ifft 16384,2049,complex64 dim=x real_dim=x
$P ifft 131072,513,complex64 dim=x real_dim=x
} > $O/irfft.txt 2>&1
grep -v "amdgpu\|Warn" $O/irfft.txt | grep "GFFT\|\] \|Error" | cut -c1-220
