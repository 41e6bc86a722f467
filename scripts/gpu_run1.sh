#!/bin/bash
# First GPU pass: parity tests, smoke, bench, rocprofv3 kernel trace.  Everything is logged under gpurun_out/.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo" ; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
echo "== rocprof kernel trace"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r01" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --cpu-slabs 0 > "$GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log" 2>&1; echo "rocprof rc=$?")
find gpurun_out/prof_r01 -name "*stats*" | head; 
f=$(find gpurun_out/prof_r01 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-220
# keep the merged-back payload small
find gpurun_out/prof_r01 -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out
echo "== other configs"; timeout 600 python scripts/bench_configs.py 2>&1 | tee gpurun_out/bench_configs.log | tail -20
