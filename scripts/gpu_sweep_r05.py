"""Round-5 random sweep on the GPU: csrc/fastn.h -- N random (ny, nx) <= HI whose lengths are products of the butterflies 2 ... 20 (7, 11, 13 included; now and then
a column length with a large prime factor: the chirp convolution), both precisions, power / complex / cross / cross-phase / isotropic / real_dim, against the oracle.
python scripts/gpu_sweep_r05.py [N] [HI] [LO] [P(awkward column length)]"""
import sys, os, warnings, collections, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
warnings.simplefilter("ignore")
from test_random_differential import run_random_fastn
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
HI = int(sys.argv[2]) if len(sys.argv) > 2 else 4800
LO = int(sys.argv[3]) if len(sys.argv) > 3 else 256
BP = float(sys.argv[4]) if len(sys.argv) > 4 else 0.15
bad = 0
served = collections.Counter()
t0 = time.time()
for seed in range(5000, 5000 + N):
    dt = "float64" if seed % 2 == 0 else "float32"
    try:
        served[run_random_fastn(seed, lo=LO, hi=HI, dtype=dt, blue_p=BP)] += 1
    except Exception as e:
        bad += 1
        print("FAIL fastn", seed, dt, repr(e)[:400], flush=True)
print(f"fastn sweep: {N} cases (lengths {LO} .. {HI}, awkward column length with p = {BP}), served by {dict(served)}, failures: {bad}, {time.time() - t0:.0f} s", flush=True)
