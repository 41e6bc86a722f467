#!/bin/bash
# the tuning build of the library: the y-first float32 kernels read their cache policies / start stagger from FastY::tune
# (XRFTHIP_YTUNE at plan creation) instead of the compiled-in default.  Used by scripts/tune_yf.py only.
cd "$(dirname "$0")/.." || exit 1
mkdir -p build_dbg
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -Wno-unused-result -DXRFT_YTUNE_RT \
  -Ixrft_amd/csrc xrft_amd/csrc/xrft_hip.cpp -o build_dbg/libxrft_hip_ytune.so
