"""Round-3 GPU sweep (one-off): seeded random differential cases against the oracle on the kernels this round touched -- the
y-first float32 pipeline (detrend scheme, radial gather), the mixed-radix table with its new lengths, the one-axis kernels, and the
generic / Bluestein paths.  Prints every failure and a count.  python scripts/gpu_sweep_r03.py > gpurun_out/sweep_r03.txt"""
import os, sys, warnings
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
warnings.simplefilter("ignore")
from test_random_differential import run_random_fastm, run_random_fast, run_random, run_random_one_axis
bad = 0
ONLY = os.environ.get("SWEEP_ONLY", "")  # e.g. "fast": that family alone
NEW = (900, 1500, 1800, 2000, 360, 720, 1000)
GRID = (320, 540, 640, 1080, 1280, 2160, 1800, 2000)  # (the Gaussian / 1/3 ... 1/12-degree lengths; float32 also 2560, 2880, 4320)
SEED0 = int(os.environ.get("SWEEP_SEED0", "0"))  # shifts every seed range: another draw of the same families
for seed in (range(5000 + SEED0, 5100 + SEED0) if ONLY in ("", "fastm", "grid") else ()):
    for dt in ("float64", "float32"):
        try:
            run_random_fastm(seed, lengths=GRID + ((2560, 2880, 4320) if dt == "float32" else ()), dtype=dt)
        except Exception as e:
            bad += 1
            print("FAIL fastm-grid", seed, dt, repr(e)[:300], flush=True)
for seed in (range(3000 + SEED0, 3120 + SEED0) if ONLY in ("", "fastm") else ()):
    for dt in ("float64", "float32"):
        try:
            run_random_fastm(seed, lengths=NEW + ((3000, 3600) if dt == "float32" else ()), dtype=dt)
        except Exception as e:
            bad += 1
            print("FAIL fastm-new", seed, dt, repr(e)[:300], flush=True)
for seed in (range(100 + SEED0, 250 + SEED0) if ONLY in ("", "fastm") else ()):
    for dt in ("float64", "float32"):
        try:
            run_random_fastm(seed, dtype=dt)
        except Exception as e:
            bad += 1
            print("FAIL fastm", seed, dt, repr(e)[:300], flush=True)
for seed in (range(100 + SEED0, 400 + SEED0) if ONLY in ("", "fast") else ()):
    try:
        run_random_fast(seed)
    except Exception as e:
        bad += 1
        print("FAIL fast", seed, repr(e)[:300], flush=True)
for seed in (range(1000 + SEED0, 1300 + SEED0) if ONLY in ("", "one-axis") else ()):
    try:
        run_random_one_axis(seed)
    except Exception as e:
        bad += 1
        print("FAIL one-axis", seed, repr(e)[:300], flush=True)
for seed in (range(500 + SEED0, 650 + SEED0) if ONLY in ("", "generic") else ()):
    try:
        run_random(seed)
    except Exception as e:
        bad += 1
        print("FAIL generic", seed, repr(e)[:300], flush=True)
print("done, failures:", bad)
