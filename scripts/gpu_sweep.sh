#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for g in 8 4 16 32; do echo -n "FAST_GROUP=$g "; XRFTHIP_FAST_GROUP=$g bash scripts/gpu_quick.sh | tr '\n' ' '; echo; done
for gr in 256 512; do echo -n "COLS_GRID=$gr "; XRFTHIP_FAST_COLS_GRID=$gr bash scripts/gpu_quick.sh | tail -1; done
