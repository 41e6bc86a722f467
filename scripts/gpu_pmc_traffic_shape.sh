#!/bin/bash
# memory-side counters of one shape (scripts/${RUN_SCRIPT:-run_shape.py}; NT / NY / NX / DT in the environment): FETCH_SIZE and WRITE_SIZE in separate passes with --kernel-trace only
# (MI355X_MICROARCH.md: FETCH_SIZE counts 64-byte units of 128-byte fabric reads on gfx950 -> x2; both in KB):  scripts/gpu_pmc_traffic_shape.sh <tag> [env assignments]
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-shape}; shift
mkdir -p gpurun_out/pmc_$TAG
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/$c" -o p -- python "$GRAFT_REPO_ROOT/scripts/${RUN_SCRIPT:-run_shape.py}" > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/$c.log" 2>&1
  echo "pass $c rc=$?"
done
cd "$GRAFT_REPO_ROOT"
python3 - "$TAG" <<'PY'
import csv, sys, glob, collections, os
tag = sys.argv[1]
nt, ny, nx = (int(os.environ.get(k, d)) for k, d in (("NT", "32"), ("NY", "1000"), ("NX", "1000")))
esz = 8 if os.environ.get("DT", "f32") == "f64" else 4
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_{tag}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "xrft::" not in k: continue
            tot[k][c] += float(r["Counter_Value"]);
            if c == "FETCH_SIZE": cnt[k] += 1
print(f"shape ({nt}, {ny}, {nx}) {'f64' if esz == 8 else 'f32'}: algorithmic bytes per slab = {2 * ny * nx * esz} (input read + power spectrum written)")
for k, v in tot.items():
    n = max(cnt[k], 1)
    rd, wr = v["FETCH_SIZE"] * 2 * 1024 / n / nt, v["WRITE_SIZE"] * 1024 / n / nt
    print(f"{k[:90]:90s} launches {n:3d}  read {rd / 1e6:8.2f} MB/slab  written {wr / 1e6:8.2f} MB/slab")
PY
find gpurun_out/pmc_$TAG -name "*.csv" -size +1M -delete
