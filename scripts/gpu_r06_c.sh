#!/bin/bash
# round 6, GPU pass C: tuning bits of the radial-sum row kernel (start stagger of a CU's residents, prefetch placement, ablations)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06c; mkdir -p $O
export TMPDIR=/tmp
for t in 0 512 1024 1536 2048 2560 3072 4096 16777216 33554432 67108864 134217728 201326592; do
  echo "XRFTHIP_YTUNE=$t" >> $O/iso_tune.txt
  XRFTHIP_YTUNE=$t timeout 300 python scripts/prof.py isoq >> $O/iso_tune.txt 2>&1
done
grep -v "Warn\|amdgpu" $O/iso_tune.txt | cut -c1-330
