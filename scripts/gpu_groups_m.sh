#!/bin/bash
# C5 (64, 1440, 720) float64 PS linear+hann on the fastm kernels: slabs per group x store policy of the intermediate / result
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in "" build_dbg/libxrft_hip_m1.so build_dbg/libxrft_hip_m3.so; do
  for g in 2 4 8 16 32 64; do
    echo "lib=${lib:-product} group=$g: $(XRFT_LIB=${lib:+$PWD/$lib} XRFTHIP_FAST_GROUP=$g ONLY_LINEAR=1 python scripts/prof_c5m.py 2>&1 | grep 'PS f64 linear')"
  done
done
