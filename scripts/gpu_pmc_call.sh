#!/bin/bash
# instruction counters of the kernels of ONE call:  scripts/gpu_pmc_call.sh <tag> <prof.py call arguments...>
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; shift
mkdir -p gpurun_out/pmc_$TAG
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
run() {
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$R/gpurun_out/pmc_$TAG/$name" -o p -- python "$R/scripts/prof.py" call "$@" --reps 2 > "$R/gpurun_out/pmc_$TAG/$name.log" 2>&1
  echo "pass $name rc=$?"
}
PMC="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" run sq1 "$@"
PMC="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU" run sq2 "$@"
cd "$R"
python3 scripts/pmc_summary.py gpurun_out/pmc_$TAG | tee gpurun_out/pmc_$TAG/summary.txt
find gpurun_out/pmc_$TAG -name "*.csv" -size +1M -delete
