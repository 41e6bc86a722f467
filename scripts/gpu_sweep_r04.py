import sys, os, warnings, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
warnings.simplefilter("ignore")
"""Round-4 random sweep on the GPU: the register-resident one-pass kernels (csrc/fasts.h, csrc/fastr.h: small slabs, rows of 4096 ... 65536
samples; many slabs per call) and, as a regression of the host-side changes (remembered analysis, read-only coordinates, Bluestein in
float64), the round-3 generators on the two-pass / one-axis / generic kernels."""
from test_random_differential import run_random_one_pass, run_random_fastm, run_random_fast, run_random, run_random_one_axis, run_random_small_slab
bad = 0
served = collections.Counter()
N1 = int(os.environ.get("SWEEP_ONE_PASS", "500"))
for seed in range(2000, 2000 + N1):
    try:
        served[run_random_one_pass(seed, big=(seed % 4 == 0))] += 1
    except Exception as e:
        bad += 1
        print("FAIL one-pass", seed, repr(e)[:400], flush=True)
print("one-pass cases:", N1, "served by", dict(served), flush=True)
served = collections.Counter()
N2 = int(os.environ.get("SWEEP_SMALL_SLAB", "400"))
for seed in range(3000, 3000 + N2):
    try:
        served[run_random_small_slab(seed)] += 1
    except Exception as e:
        bad += 1
        print("FAIL small slab", seed, repr(e)[:400], flush=True)
print("small slabs of any smooth shape:", N2, "served by", dict(served), flush=True)
for seed in range(400, 400 + int(os.environ.get("SWEEP_FASTM", "120"))):
    for dt in ("float64", "float32"):
        try:
            run_random_fastm(seed, dtype=dt)
        except Exception as e:
            bad += 1
            print("FAIL fastm", seed, dt, repr(e)[:300], flush=True)
for seed in range(300, 300 + int(os.environ.get("SWEEP_FAST", "120"))):
    try:
        run_random_fast(seed)
    except Exception as e:
        bad += 1
        print("FAIL fast", seed, repr(e)[:300], flush=True)
for seed in range(1500, 1500 + int(os.environ.get("SWEEP_ONE_AXIS", "200"))):
    try:
        run_random_one_axis(seed)
    except Exception as e:
        bad += 1
        print("FAIL one-axis", seed, repr(e)[:300], flush=True)
for seed in range(500, 500 + int(os.environ.get("SWEEP_GENERIC", "150"))):
    try:
        run_random(seed)
    except Exception as e:
        bad += 1
        print("FAIL generic", seed, repr(e)[:300], flush=True)
print("done, failures:", bad)
