#!/bin/bash
# round 6, GPU pass I: the LONG sweep of the GPU suite (every seed of rounds 3-5, the largest shapes) on the final kernels
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06i; mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s); XRFT_GPU_SWEEP=long timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu_long.txt 2>&1; echo "wall $(( $(date +%s) - T0 )) s" >> $O/pytest_gpu_long.txt; tail -6 $O/pytest_gpu_long.txt
