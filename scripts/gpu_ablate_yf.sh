#!/bin/bash
# phase ablation of the y-first kernels (XRFTHIP_YDBG bits, fasty.h): us per slab of each kernel with a phase switched off
cd "$GRAFT_REPO_ROOT" || exit 1
for d in 0 1 2 4 8 12 16 32 64 96 112; do
  echo "YDBG=$d: $(XRFTHIP_YDBG=$d ONLY=linear,hann python scripts/prof_yf.py 2>&1 | grep 'PS linear hann')"
done
