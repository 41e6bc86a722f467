#!/bin/bash
# ablation builds of the library: one .so per XRFT_MDBG value (fastm.h) under build_dbg/ (git-ignored, shipped by gpurun);
# suffix "r": half the rows per workgroup in pass 2 (one field);
# a value with suffix "b" also builds with 110 KB of LDS per workgroup (twice the sequences: 64-byte instead of 32-byte row segments)
cd "$(dirname "$0")/.." || exit 1
mkdir -p build_dbg
for d in "$@"; do
  extra=""; n=$d
  case $d in *b) n=${d%b}; extra="-DXRFT_M_LDSCAP=112640 -DXRFT_M_BIGLDS=1";; *r) n=${d%r}; extra="-DXRFT_M_ROWS_SHIFT=1";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -Wno-unused-result -DXRFT_MDBG=$n $extra -Ixrft_amd/csrc xrft_amd/csrc/xrft_hip.cpp -o build_dbg/libxrft_hip_m$d.so &
done
wait
ls -la build_dbg
