#!/bin/bash
# round 6, GPU pass D: radial-sum row kernel with the late prefetch and the pipelined gather: A/B, ablations; C2 with the default stagger
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06d; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python scripts/prof.py iso > $O/tune_iso.txt 2>&1; grep -v "Warn\|amdgpu" $O/tune_iso.txt | cut -c1-330
for t in 16777216 134217728; do
  echo "XRFTHIP_YTUNE=$t" >> $O/iso_tune.txt
  XRFTHIP_YTUNE=$t timeout 300 python scripts/prof.py isoq >> $O/iso_tune.txt 2>&1
done
grep -v "Warn\|amdgpu" $O/iso_tune.txt | cut -c1-330
timeout 300 python scripts/prof.py c2 --reps 50 > $O/c2.txt 2>&1; grep -v amdgpu $O/c2.txt
