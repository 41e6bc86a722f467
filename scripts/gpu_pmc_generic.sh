#!/bin/bash
# PMC counter passes for the generic tile kernels on the C5-shaped workload
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-generic}
mkdir -p gpurun_out/pmc_$TAG
export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/$name" -o p -- python "$GRAFT_REPO_ROOT/scripts/run_c5.py" > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/$name.log" 2>&1
  echo "pass $name rc=$?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run fetch FETCH_SIZE
run write WRITE_SIZE
cd "$GRAFT_REPO_ROOT"
python3 scripts/pmc_summary.py gpurun_out/pmc_$TAG | tee gpurun_out/pmc_$TAG/summary.txt
find gpurun_out/pmc_$TAG -name "*.csv" -size +5M -delete
