#!/bin/bash
# the whole -m gpu suite + smoke() on the box, with the wall time (what the driver runs at round end)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/suite; mkdir -p $O
T0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
echo "wall $(( $(date +%s) - T0 )) s" | tee -a $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
