#!/bin/bash
# tile / table knobs of the generic tile kernels on the C5 shape (64,1440,720) float64 (read at plan creation)
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo "$*: $(env "$@" python scripts/prof_c25.py 2>&1 | grep -E '^C5 PS f64 linear' | sed 's/.*||//')"; }
run A=0
run XRFTHIP_LDS_SOFT=81920
run XRFTHIP_LDS_SOFT=65536
run XRFTHIP_LDS_SOFT=49152
run XRFTHIP_TW_LDS=0
run XRFTHIP_REV_LDS=0
run XRFTHIP_TW_LDS=0 XRFTHIP_REV_LDS=0
run XRFTHIP_TW_LDS=0 XRFTHIP_LDS_SOFT=81920
run XRFTHIP_COMPOSITE=0
run XRFTHIP_RADIX16=0
run XRFTHIP_Y_MIN_T=2 XRFTHIP_LDS_SOFT=49152
