#!/bin/bash
# instruction counters of the run-time-radix kernels (fastn.h) against the table kernels (fastm.h) on one shape:  scripts/gpu_pmc_fastn.sh <tag> [env assignments]
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-fastn}; shift
mkdir -p gpurun_out/pmc_$TAG
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
cd /tmp
run() {
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/$name" -o p -- python "$GRAFT_REPO_ROOT/scripts/run_shape.py" > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/$name.log" 2>&1
  echo "pass $name rc=$?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU
cd "$GRAFT_REPO_ROOT"
python3 scripts/pmc_summary.py gpurun_out/pmc_$TAG | tee gpurun_out/pmc_$TAG/summary.txt
find gpurun_out/pmc_$TAG -name "*.csv" -size +1M -delete
