#!/bin/bash
# phase ablation of the fastm kernels on C5 (64, 1440, 720) float64 PS linear+hann: us per slab with a phase compiled out
# (build_dbg/libxrft_hip_m<bits>.so from scripts/build_ablate_m.sh: 4 = no transforms, 8 = no stores, 16 = no loads)
cd "$GRAFT_REPO_ROOT" || exit 1
echo "product: $(ONLY_LINEAR=1 python scripts/prof_c5m.py 2>&1 | grep 'PS f64 linear')"
for f in build_dbg/libxrft_hip_m*.so; do
  echo "$(basename $f .so): $(XRFT_LIB=$PWD/$f ONLY_LINEAR=1 python scripts/prof_c5m.py 2>&1 | grep 'PS f64 linear')"
done
