#!/bin/bash
# rocprofv3 --kernel-trace --stats of the lengths-as-data kernels (csrc/fastg.h): small slabs, one transform axis on any smooth length, inverse transforms
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
for s in time_axis rows_generic inverse; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_$s" -o p -- python3 "$GRAFT_REPO_ROOT/scripts/prof_$s.py" > "$GRAFT_REPO_ROOT/$O/prof_$s.txt" 2>&1; echo "rocprof $s rc=$?")
  f=$(find $O/prof_$s -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${s}_kernel_stats.csv && grep -i "fastg\|Name" "$f" | head -8 | cut -c1-160
done
find $O -name "*kernel_trace.csv" -size +4M -delete
