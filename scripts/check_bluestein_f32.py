"""float32 data on Bluestein lengths INSIDE the tile kernels, float32 arithmetic (bluestein_in_float64(False)): the finer error norms of
tests/cases.py (worst per-bin relative error above the floor, L1) against the oracle on float64 copies of the samples."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
warnings.simplefilter("ignore")
import xrft_amd as xa
from xrft_amd import api
import cases
from cases import fine_errors, o, pair

xa.bluestein_in_float64(False)
worst = [0.0, 0.0]


def chk(got, ref, tol, bin_rel=1e-3):
    gv, rv = np.asarray(got.values), np.asarray(ref.values)
    b, l1 = fine_errors(gv, rv)
    worst[0], worst[1] = max(worst[0], b), max(worst[1], l1)
    d = next(reversed(api._plan_cache.values())).describe().split("\n")
    print(f"shape {gv.shape} {gv.dtype} binrel {b:.2e} l1 {l1:.2e} | {d[0][14:70]} | {d[1][:60]}", flush=True)
    return 0.0


cases.check = chk
cases.run_bluestein_cases("float32")
print(f"in-tile Bluestein, float32 arithmetic: worst per-bin rel err {worst[0]:.2e}, worst L1 {worst[1]:.2e}")
worst[:] = [0.0, 0.0]
rng = np.random.default_rng(5)
for shape in ((2, 721, 1440), (2, 103, 206), (1, 1440, 721)):
    v = cases._cube(rng, shape, "float32")
    da, od = pair(v, cases.D3, cases._coords3(shape))
    chk(xa.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"), o.power_spectrum(od, dim=["y", "x"], detrend="linear", window="hann"), 3e-4)
    chk(xa.power_spectrum(da, dim=["y", "x"]), o.power_spectrum(od, dim=["y", "x"]), 3e-4)
    chk(xa.fft(da, dim=["y", "x"]), o.fft(od, dim=["y", "x"]), 3e-4)
print(f"ERA5-like slabs: worst per-bin rel err {worst[0]:.2e}, worst L1 {worst[1]:.2e}")
