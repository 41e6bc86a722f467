#!/bin/bash
# round 6, GPU pass A: the new radial-sum row kernel (parity, A/B, phase clocks), the wide column pass, the fastr start stagger, the new bench line
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "iso or config4 or radial or cross" > $O/pytest_iso.txt 2>&1; echo "pytest iso: $?" >> $O/pytest_iso.txt; tail -3 $O/pytest_iso.txt
timeout 600 python scripts/prof.py iso > $O/tune_iso.txt 2>&1; tail -50 $O/tune_iso.txt
for lib in "" "--lib build_dbg/libxrft_hip_wide.so"; do timeout 300 python scripts/prof.py headline $lib >> $O/cols_wide.txt 2>&1; done; cat $O/cols_wide.txt
for sg in 0 515 518 520 522 524 526 772 774 776 1028 1030; do echo "XRFTHIP_FASTR_STAGGER=$sg" >> $O/c2_stagger.txt; XRFTHIP_FASTR_STAGGER=$sg timeout 300 python scripts/prof.py c2 2>&1 | head -4 >> $O/c2_stagger.txt; done; cat $O/c2_stagger.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 6000 $O/bench.json; tail -5 $O/bench.err
timeout 600 python bench.py --workload c4 --no-extra > $O/bench_c4.json 2>> $O/bench.err; tail -c 1500 $O/bench_c4.json
