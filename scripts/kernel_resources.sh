#!/bin/bash
# VGPRs / spills / occupancy of every kernel of one translation unit, as hipcc reports them:  scripts/kernel_resources.sh xrft_amd/csrc/inst_g6.cpp [extra flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-result -I"$(dirname "$0")/../xrft_amd/csrc" "$@" -c "$src" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage 2>&1 | awk '
    /Function Name|remark:.* Name:/ { name=$NF; sub(/\[.*/,"",name); n=$0; sub(/.*Name: /,"",n); sub(/ \[.*/,"",n); name=n }
    / VGPRs:/ { v=$0; sub(/.* VGPRs: /,"",v); sub(/ .*/,"",v) }
    /ScratchSize/ { s=$0; sub(/.*: /,"",s); sub(/ .*/,"",s) }
    /Occupancy/ { o=$0; sub(/.*: /,"",o); sub(/ .*/,"",o) }
    /SGPRs Spill/ { g=$0; sub(/.*: /,"",g); sub(/ .*/,"",g) }
    /VGPRs Spill/ { p=$0; sub(/.*: /,"",p); sub(/ .*/,"",p); printf "%-70s vgpr %3s occ %s scratch %4s B  vspill %3s sspill %3s\n", name, v, o, s, p, g }' | c++filt 2>/dev/null
