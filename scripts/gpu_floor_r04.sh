#!/bin/bash
# Round 4, the headline's floor: (a) the rendezvous of the workgroups that share an input line (tuning build, XRFTHIP_YTUNE bit 21): times
# and FETCH_SIZE with and without; (b) memory-side counters of the two passes (L2 <-> fabric requests and stalls, L2 hit rate, L1 / TA
# busy) beside the skeleton's: which unit is saturated at the two passes' ~5.6-5.9 TB/s.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_floor
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1
python3 scripts/tune_rdv.py > $O/tune_rdv.txt 2>&1; echo "tune_rdv rc=$?"
cd /tmp
pmc() {  # name tune counters...
  name=$1; tune=$2; shift 2
  XRFTHIP_YTUNE=$tune REPS=2 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/$O/$name" -o p -- python3 "$GRAFT_REPO_ROOT/scripts/run_ps_tune.py" > "$GRAFT_REPO_ROOT/$O/$name.log" 2>&1
  echo "pmc $name rc=$?"
}
for t in 0 2097152; do
  pmc fetch_$t $t FETCH_SIZE
  pmc write_$t $t WRITE_SIZE
done
pmc tcc_hit 0 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pmc tcc_ea1 0 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pmc tcc_ea2 0 TCC_EA0_WRREQ_STALL_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum
pmc tcc_ea3 0 TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum
pmc tcc_ea4 0 TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum
pmc tcp1 0 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pmc tcp2 0 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_NC_READ_REQ_sum
pmc ta 0 TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pmc tcc_lvl 0 TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_CYCLE_sum
pmc tcc_fifo 0 TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum TCC_IB_STALL_sum TCC_NORMAL_EVICT_sum
pmc grbm 0 GRBM_GUI_ACTIVE GRBM_COUNT
cd "$GRAFT_REPO_ROOT"
python3 scripts/floor_counters_summary.py $O > $O/summary.txt 2>&1; cat $O/summary.txt | head -80
find $O -name "*kernel_trace.csv" -size +4M -delete
