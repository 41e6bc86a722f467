#!/bin/bash
# slabs per group vs time per slab for the y-first path (the intermediate of a small group stays in the Infinity Cache)
cd "$GRAFT_REPO_ROOT" || exit 1
for g in ${GROUPS_LIST:-1 2 3 4 6 8 16 32}; do
  echo "group=$g: $(XRFTHIP_FAST_GROUP=$g ONLY=linear,hann NT=${NT:-48} python scripts/prof_yf.py 2>&1 | grep 'PS linear hann')"
done
