#!/bin/bash
# which phase of the generic tile kernel costs what: XRFTHIP_DBG 1 = no passes, 2 = no store, 4 = no load
cd "$GRAFT_REPO_ROOT" || exit 1
for dbg in 0 1 2 3 4 5 6 7; do
  echo "== XRFTHIP_DBG=$dbg"; XRFTHIP_DBG=$dbg python scripts/prof_generic3.py 2>&1 | grep -E "^C5|^   same|^C2|^PS"
done
