#!/bin/bash
# round-2 evidence for profiles/: (1) rocprofv3 --kernel-trace --stats of the driver's bench command, (2) PMC passes of the
# bench workload (--nt 8), (3) FETCH_SIZE / WRITE_SIZE of the memory-pattern skeletons (known byte counts: calibration)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_bench" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --cpu-slabs 0 > "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_bench.err"; echo "rocprof bench rc=$?")
f=$(find gpurun_out/r02/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r02/bench_kernel_stats.csv && head -8 "$f" | cut -c1-220
find gpurun_out/r02/prof_bench -name "*kernel_trace.csv" -size +8M -delete
# the driver's command as it is, and the other BASELINE.json configurations through the same bench (one JSON line each)
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02/bench_driver_cmd.json 2> gpurun_out/r02/bench_driver_cmd.err; echo "bench rc=$?"
for w in c2 c4 c5; do
  timeout 300 python3 bench.py --gpus 1 --steps 10 --warmup 3 --workload $w --cpu-slabs 0 > gpurun_out/r02/bench_$w.json 2> gpurun_out/r02/bench_$w.err; echo "bench $w rc=$?"
done
timeout 600 python3 scripts/bench_configs.py > gpurun_out/r02/bench_configs.txt 2>&1; echo "configs rc=$?"
bash scripts/gpu_pmc_yf.sh r02 sq1 sq2 tcc2 fetch write grbm > gpurun_out/r02/pmc.log 2>&1; tail -2 gpurun_out/r02/pmc.log
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r02/pmc_ubench_$c" -o u -- "$GRAFT_REPO_ROOT/scripts/ubench/yfirst" > /dev/null 2>&1; echo "ubench $c rc=$?")
done
python3 - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r02/pmc_ubench_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == c: agg[row["Kernel_Name"][:70]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        print(c, k, "n=%d" % len(v), "first=%.0f KB" % v[0], "median=%.0f KB" % sorted(v)[len(v)//2])
PY
