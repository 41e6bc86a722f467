#!/usr/bin/env python3
"""Per-kernel HIP-event timings (us per slab) of the y-first path at 4096^2 for detrend / window combinations.  GPU box."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api, _lib
if os.environ.get("XRFT_LIB"): _lib.load(os.environ["XRFT_LIB"])  # an ablation build (scripts/build_ablate_yf.sh)
warnings.simplefilter("ignore")
nt, n = int(os.environ.get("NT", 32)), int(os.environ.get("N", 4096))
a = torch.randn((nt, n, n), dtype=torch.float32, device="cuda"); c = {"y": np.arange(float(n)), "x": np.arange(float(n))}
d1 = xrft.DataArray(a, ("t", "y", "x"), c)
def prof(name, fn):
    fn(); fn(); torch.cuda.synchronize()
    plan = [p for p in api._plan_cache.values()][-1]
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3
    plan.set_profiling(True)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    p = plan.read_profile(); plan.set_profiling(False)
    tot = sum(ms for c, ms in p.values()) / 3 * 1e3 / nt
    print(f"{name:28s}", " | ".join(f"{k} {ms/3*1e3/nt:.1f}" for k, (c, ms) in p.items()), f"|| kernels {tot:.1f} us/slab, wall {wall*1e6/nt:.1f} us/slab", flush=True)
only = os.environ.get("ONLY")
for det in (None, "constant", "linear"):
    for win in (None, "hann"):
        if only and only != f"{det},{win}".lower(): continue
        prof(f"PS {det} {win}", lambda: xrft.power_spectrum(d1, dim=["y", "x"], detrend=det, window=win))
