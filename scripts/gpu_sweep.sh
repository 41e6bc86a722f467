#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for st in 0 1 2 3 4; do echo -n "STAGGER=$st "; XRFTHIP_FAST_STAGGER=$st bash scripts/gpu_quick.sh | tail -1; done
for gr in 256 384 512; do echo -n "COLS_GRID=$gr "; XRFTHIP_FAST_COLS_GRID=$gr bash scripts/gpu_quick.sh | tail -1; done
