#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "0 1" "2 1" "2 2" "4 1" "4 2" "3 1" "8 1"; do set -- $cfg; echo -n "STAGGER=$1 SLEEPS=$2 "; XRFTHIP_FAST_STAGGER=$1 XRFTHIP_FAST_STAGGER_SLEEPS=$2 bash scripts/gpu_quick.sh | tail -1; done
