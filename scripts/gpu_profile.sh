#!/bin/bash
# the evidence pass for profiles/ (round tag: $ROUND, default r06): (1) PMC passes of the bench workload at its own batch (FETCH_SIZE / WRITE_SIZE -> profiles/${R}_traffic.json,
# stamped with the SHA-1 of xrft_amd/csrc) and of the C2 workload (the one-pass row kernel -> profiles/${R}_traffic_c2.json), (2) rocprofv3
# --kernel-trace --stats of the driver's bench command, (3) the driver's command as it is + the other BASELINE.json configurations,
# (4) bench_configs, (5) host time per call
cd "$GRAFT_REPO_ROOT" || exit 1
R=${ROUND:-r06}
O=gpurun_out/$R
mkdir -p $O
export TMPDIR=/tmp
if [ "${1:-}" != "nopmc" ]; then
  bash scripts/gpu_pmc_yf.sh $R fetch write sq1 sq2 > $O/pmc.log 2>&1; tail -2 $O/pmc.log
  python3 scripts/make_traffic_json.py gpurun_out/pmc_$R ${PMC_NT:-64} > $O/traffic.json 2> $O/traffic.err; echo "traffic rc=$?"
  [ -s $O/traffic.json ] && cp $O/traffic.json profiles/${R}_traffic.json
  mkdir -p gpurun_out/pmc_${R}c2
  for cnt in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_${R}c2/$cnt" -o p -- python3 "$GRAFT_REPO_ROOT/bench.py" --workload c2 --steps 1 --warmup 1 --cpu-slabs 0 --no-profile --no-extra --no-floor > "$GRAFT_REPO_ROOT/gpurun_out/pmc_${R}c2/$cnt.log" 2>&1; echo "c2 pmc $cnt rc=$?")
  done
  TRAFFIC_WORKLOAD=c2 python3 scripts/make_traffic_json.py gpurun_out/pmc_${R}c2 1024 > $O/traffic_c2.json 2> $O/traffic_c2.err; echo "traffic c2 rc=$?"
  [ -s $O/traffic_c2.json ] && cp $O/traffic_c2.json profiles/${R}_traffic_c2.json
fi
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_bench" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --cpu-slabs 0 --no-extra --no-floor > "$GRAFT_REPO_ROOT/$O/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$O/prof_bench.err"; echo "rocprof bench rc=$?")
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv && head -6 "$f" | cut -c1-200
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_c2" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --workload c2 --steps 20 --warmup 5 --cpu-slabs 0 --no-extra > "$GRAFT_REPO_ROOT/$O/prof_c2.json" 2> "$GRAFT_REPO_ROOT/$O/prof_c2.err"; echo "rocprof c2 rc=$?")
f=$(find $O/prof_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/c2_kernel_stats.csv && head -4 "$f" | cut -c1-200
find $O -name "*kernel_trace.csv" -size +8M -delete
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?"
# (C2's step is one 0.18-ms kernel: 50 steps, or the barrier + synchronize bracket of the timed region -- 0.4 ms -- is a fifth of it)
timeout 300 python3 bench.py --gpus 1 --steps 50 --warmup 5 --workload c2 --cpu-slabs 0 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?"
for w in c4 c5; do
  timeout 300 python3 bench.py --gpus 1 --steps 10 --warmup 3 --workload $w --cpu-slabs 0 > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
timeout 1500 python3 scripts/bench_configs.py > $O/bench_configs.txt 2>&1; echo "configs rc=$?"
timeout 300 python3 scripts/host_overhead.py > $O/host_overhead.txt 2>&1; echo "host rc=$?"
