#!/bin/bash
# PMC counter passes for the bench workload (--nt ${PMC_NT:-64}: the bench batch). Usage: gpu_pmc_yf.sh <tag> [passes...]
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r02}; shift
PASSES=${@:-sq1 sq2 fetch write grbm}
mkdir -p gpurun_out/pmc_$TAG
export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/$name" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --nt ${PMC_NT:-64} --cpu-slabs 0 --no-profile --no-extra --no-floor > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/$name.log" 2>&1
  echo "pass $name rc=$?"
}
for p in $PASSES; do
  case $p in
    sq1) run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS ;;
    sq2) run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM ;;
    sq3) run sq3 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT ;;
    tcc1) run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum ;;
    tcc2) run tcc2 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum ;;
    fetch) run fetch FETCH_SIZE ;;
    write) run write WRITE_SIZE ;;
    grbm) run grbm GRBM_GUI_ACTIVE GRBM_COUNT ;;
  esac
done
cd "$GRAFT_REPO_ROOT"
python3 scripts/pmc_summary.py gpurun_out/pmc_$TAG | tee gpurun_out/pmc_$TAG/summary.txt
