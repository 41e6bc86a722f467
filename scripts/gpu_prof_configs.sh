#!/bin/bash
# rocprofv3 kernel trace of the non-headline configurations + PMC counters of the generic kernels on the C5 shape
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_configs" -o cfg -- python "$GRAFT_REPO_ROOT/scripts/bench_configs.py" > "$GRAFT_REPO_ROOT/gpurun_out/prof_configs.log" 2>&1; echo "rocprof rc=$?")
f=$(find gpurun_out/prof_configs -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" | cut -c1-200
find gpurun_out/prof_configs -name "*kernel_trace.csv" -size +5M -delete
bash scripts/gpu_pmc_generic.sh generic2 > gpurun_out/pmc_generic2.log 2>&1; tail -3 gpurun_out/pmc_generic2.log
