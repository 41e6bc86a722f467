#!/bin/bash
# rocprofv3 --kernel-trace --stats of this round's kernels: the run-time-radix pipeline with the Rader columns (csrc/fastn.h), the Rader forms of the one-axis kernel
# (csrc/fastg.h), the fused passes for the inner / mid layouts, the half-output form (real_dim along an axis)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
for s in rader2 rader rows_primes mid inner_r05 realdim_axis; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_$s" -o p -- python3 "$GRAFT_REPO_ROOT/scripts/prof_$s.py" > "$GRAFT_REPO_ROOT/$O/prof_$s.txt" 2>&1; echo "rocprof $s rc=$?")
  f=$(find $O/prof_$s -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${s}_kernel_stats.csv && grep -i "xrft::\|Name" "$f" | head -6 | cut -c1-150
done
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
