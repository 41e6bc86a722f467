#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for dbg in 0 1 2 4 3 5 6 7; do
  echo -n "dbg=$dbg  "
  XRFTHIP_FAST_DBG=$dbg python bench.py --steps 2 --warmup 1 --cpu-slabs 0 --nt 32 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print({k: round(v*1000/32,1) for k,v in r['kernels_ms_per_step'].items()})"
done
