#!/bin/bash
# phase ablation of the y-first kernels: us per slab of each kernel with a phase compiled out (build_dbg/libxrft_hip_dbg<bits>.so)
cd "$GRAFT_REPO_ROOT" || exit 1
echo "product: $(ONLY=linear,hann python scripts/prof_yf.py 2>&1 | grep 'PS linear hann')"
for f in build_dbg/libxrft_hip_dbg*.so; do
  echo "$(basename $f .so): $(XRFT_LIB=$PWD/$f ONLY=linear,hann python scripts/prof_yf.py 2>&1 | grep 'PS linear hann')"
done
