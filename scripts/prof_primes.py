#!/usr/bin/env python3
"""Lengths with a prime factor above 5: the O(r^2) butterfly (XRFTHIP_GENERIC_MAX=128: always) vs Bluestein (default: above 16)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
for shp in ((64, 721, 1440), (64, 1440, 721), (64, 1001, 1001), (64, 1088, 992), (64, 824, 1024), (64, 1024, 1054)):
    x = torch.randn(shp, dtype=torch.float32, device="cuda")
    da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(shp[1]) * 1.0, "x": np.arange(shp[2]) * 1.0})
    fn = lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")
    fn(); fn(); torch.cuda.synchronize()
    plan = [p for p in api._plan_cache.values()][-1]
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3
    d = [l.strip()[:100] for l in plan.describe().strip().split("\n")[1:3]]
    print(f"PS f32 {shp}: {wall*1e3:.3f} ms = {x.numel()/wall/1e9:.1f} GFFT/s   {d}", flush=True)
    del x, da
