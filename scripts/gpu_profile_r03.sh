#!/bin/bash
# round-3 evidence for profiles/: (1) rocprofv3 --kernel-trace --stats of the driver's bench command, (2) the driver's command as
# it is + the other BASELINE.json configurations, (3) bench_configs, (4) PMC passes of the bench workload at its own batch (--nt 64):
# FETCH_SIZE / WRITE_SIZE (-> profiles/r03_traffic.json, stamped with the SHA-1 of xrft_amd/csrc) and the SQ counters
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
# the counters first: bench.py quotes roofline.traffic from profiles/r03_traffic.json only when it carries the SHA-1 of the csrc/ that runs
if [ "${1:-}" != "nopmc" ]; then
  bash scripts/gpu_pmc_yf.sh r03 fetch write sq1 sq2 > $O/pmc.log 2>&1; tail -2 $O/pmc.log
  python3 scripts/make_traffic_json.py gpurun_out/pmc_r03 ${PMC_NT:-64} profiles/r03_ubench_fused.txt > $O/traffic.json 2> $O/traffic.err; echo "traffic rc=$?"
  [ -s $O/traffic.json ] && cp $O/traffic.json profiles/r03_traffic.json
fi
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_bench" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --cpu-slabs 0 > "$GRAFT_REPO_ROOT/$O/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$O/prof_bench.err"; echo "rocprof bench rc=$?")
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv && head -6 "$f" | cut -c1-200
find $O/prof_bench -name "*kernel_trace.csv" -size +8M -delete
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?"
for w in c2 c4 c5; do
  timeout 300 python3 bench.py --gpus 1 --steps 10 --warmup 3 --workload $w --cpu-slabs 0 > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
timeout 600 python3 scripts/bench_configs.py > $O/bench_configs.txt 2>&1; echo "configs rc=$?"
