#!/usr/bin/env python3
"""Per-kernel HIP-event times of spectra over (y, x) of a (y, x, t) array: two adjacent transform axes, the batch innermost (xrfthip_desc.inner)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
for shape in ((1024, 1024, 64), (2048, 2048, 16), (720, 1440, 32)):
    for dt in (torch.float32, torch.float64):
        x = torch.randn(shape, dtype=dt, device="cuda")
        da = xrft.DataArray(x, ("y", "x", "t"), {"y": np.arange(float(shape[0])), "x": np.arange(float(shape[1]))})
        for name, fn in (("PS linear+hann", lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")), ("PS", lambda: xrft.power_spectrum(da, dim=["y", "x"])), ("fft", lambda: xrft.fft(da, dim=["y", "x"]))):
            fn(); fn(); torch.cuda.synchronize()
            plan = next(reversed(api._plan_cache.values()))
            t0 = time.perf_counter()
            for _ in range(5): fn()
            torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5
            plan.set_profiling(True)
            for _ in range(5): fn()
            torch.cuda.synchronize()
            p = plan.read_profile(); plan.set_profiling(False)
            print(f"{str(shape):18s} {str(dt)[6:]:8s} {name:15s} {x.numel() / wall / 1e9:6.1f} GFFT/s  wall {wall * 1e3:.3f} ms | " + " | ".join(f"{k} {ms / 5:.3f}" for k, (c, ms) in p.items()), flush=True)
        del x, da
