#!/bin/bash
# round 6, GPU pass B: the radial-sum row kernel after the gather rewrite; fastr stagger, fine sweep
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06b; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python scripts/prof.py iso > $O/tune_iso.txt 2>&1; grep -v "Warn\|amdgpu" $O/tune_iso.txt | cut -c1-330
for sg in 0 513 514 515 516 517 770 771 1026 1027; do echo "XRFTHIP_FASTR_STAGGER=$sg" >> $O/c2_stagger.txt; XRFTHIP_FASTR_STAGGER=$sg timeout 300 python scripts/prof.py c2 --reps 50 2>&1 | grep -v amdgpu | head -6 >> $O/c2_stagger.txt; done; cat $O/c2_stagger.txt
