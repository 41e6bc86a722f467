#!/usr/bin/env python3
"""A/B of two builds of the library on ONE box: per-kernel HIP-event times of the headline power spectrum (64, 4096, 4096) float32 through
the library given as argv[1] (default: the product).  python scripts/ab_lib.py build_dbg/libxrft_hip_old.so"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xrft_amd import _lib
if len(sys.argv) > 1: _lib.load(os.path.abspath(sys.argv[1]))
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
nt = 64
x = torch.randn((nt, 4096, 4096), dtype=torch.float32, device="cuda")
x += (0.01 * torch.arange(4096, device="cuda"))[None, :, None]
da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(4096.), "x": np.arange(4096.)})
f = lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")
r = f(); r = f(); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10): r = f()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 10
    plan = next(reversed(api._plan_cache.values()))
    plan.set_profiling(True)
    for _ in range(5): r = f()
    torch.cuda.synchronize()
    prof = plan.read_profile(); plan.set_profiling(False)
    print(f"{sys.argv[1] if len(sys.argv) > 1 else 'product':40s} wall {wall / nt * 1e6:6.2f} us/slab = {nt * 4096 * 4096 / wall / 1e9:6.1f} GFFT/s | " + " ".join(f"{k} {v[1] / 5 / nt * 1e3:5.2f}" for k, v in prof.items()), flush=True)
