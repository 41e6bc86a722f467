#!/bin/bash
# tests for the specialised path + the per-config throughput table
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider -k "${K:-fastp2 or config or slope}" 2>&1 | tail -8
timeout 600 python scripts/bench_configs.py 2>&1 | tee gpurun_out/bench_configs.log | tail -20
