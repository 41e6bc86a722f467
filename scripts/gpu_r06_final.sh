#!/bin/bash
# round 6, closing pass on the final tree: the evidence pass (scripts/gpu_profile.sh), the long GPU sweep of the test suite, a seeded random differential sweep
cd "$(dirname "$0")/.." || exit 1
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=r06 bash scripts/gpu_profile.sh 2>&1 | tail -30
O=gpurun_out/r06; mkdir -p $O
T0=$(date +%s)
XRFT_GPU_SWEEP=long timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_long.txt 2>&1; tail -2 $O/pytest_gpu_long.txt
echo "long sweep wall $(( $(date +%s) - T0 )) s"
timeout 900 python scripts/gpu_sweep_random.py 200 > $O/random_sweep.txt 2>&1; tail -3 $O/random_sweep.txt
