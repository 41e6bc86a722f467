#!/bin/bash
# round 6, GPU pass F: the whole GPU suite as the driver runs it (timed), then the long random sweeps
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06f; mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s); timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu.txt 2>&1; echo "wall $(( $(date +%s) - T0 )) s" >> $O/pytest_gpu.txt; tail -35 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
