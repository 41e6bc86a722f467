import sys, os, warnings, traceback
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
warnings.simplefilter("ignore")
from test_random_differential import run_random_fastm, run_random_fast, run_random, run_random_one_axis
bad = 0
for seed in range(100, 400):
    for dt in ("float64", "float32"):
        try:
            run_random_fastm(seed, dtype=dt)
        except Exception as e:
            bad += 1
            print("FAIL fastm", seed, dt, repr(e)[:300], flush=True)
for seed in range(100, 300):
    try:
        run_random_fast(seed)
    except Exception as e:
        bad += 1
        print("FAIL fast", seed, repr(e)[:300], flush=True)
nfast = 0
for seed in range(1000, 1500):
    try:
        nfast += bool(run_random_one_axis(seed))
    except Exception as e:
        bad += 1
        print("FAIL one-axis", seed, repr(e)[:300], flush=True)
print("one-axis cases on the one-pass kernels:", nfast, "of 500")
print("done, failures:", bad)
