#!/bin/bash
# rebuild ONE translation unit of the library and relink (kernel-body iterations: the other units' objects in build/obj stay valid as long as
# no shared struct or launch interface changed).  usage: scripts/rebuild_unit.sh inst_g1 [more units...]
cd "$(dirname "$0")/.." || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-result -Ixrft_amd/csrc"
for u in "$@"; do
  extra=""; case "$u" in xrft_hip|host_*|ops) extra="-DXRFT_SPLIT_TUS";; esac
  /opt/rocm/bin/hipcc $F $extra -c xrft_amd/csrc/$u.cpp -o build/obj/$u.o 2>&1 | grep -E "error|Error" ; 
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/obj/*.o -Wl,-z,defs -o xrft_amd/libxrft_hip.so && ls -la xrft_amd/libxrft_hip.so
