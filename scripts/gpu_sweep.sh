#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for ug in 8 4 16 32; do echo -n "UNTILE_GROUP=$ug "; XRFTHIP_FAST_UNTILE_GROUP=$ug bash scripts/gpu_quick.sh | tr '\n' ' '; echo; done
