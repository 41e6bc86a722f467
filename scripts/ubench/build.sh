#!/bin/bash
# build the stand-alone micro-benchmarks for gfx950 (run the binaries on the GPU box: gpurun -- ./scripts/ubench/<name>)
cd "$(dirname "$0")" || exit 1
for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$f" -o "${f%.hip}" || exit 1; done
