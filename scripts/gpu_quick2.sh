#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for mr in 16 4 8 32 64; do echo -n "MOMENT_ROWS=$mr "; XRFTHIP_MOMENT_ROWS=$mr bash scripts/gpu_quick.sh | tail -1; done
bash scripts/gpu_quick.sh
