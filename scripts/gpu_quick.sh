#!/bin/bash
# quick perf check: bench only (short), optional env passthrough
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python bench.py --steps 3 --warmup 1 --cpu-slabs 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('GFFT/s', d['value'], 'ms/step', d['ms_per_step'], 'parity', d['parity_max_rel_err_vs_oracle'])
print('per-slab us:', {k: round(v*1000/64,1) for k,v in r['kernels_ms_per_step'].items()}, 'sum', round(r['path']['sum_kernels_ms_per_step']*1000/64,1))
"
