#!/bin/bash
# ablation builds of the library: one .so per XRFT_YDBG value (fasty.h) under build_dbg/ (git-ignored, shipped by gpurun)
cd "$(dirname "$0")/.." || exit 1
mkdir -p build_dbg
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -Wno-unused-result -DXRFT_YDBG=$d -Ixrft_amd/csrc xrft_amd/csrc/xrft_hip.cpp -o build_dbg/libxrft_hip_dbg$d.so &
done
wait
ls -la build_dbg
