#!/usr/bin/env python3
"""1-D transforms of 8192 ... 32768 samples (between the one-pass kernels of csrc/fastm.h, <= 4096, and the four-step form, >= 65536)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
def timeit(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
for nb, n in ((16384, 8192), (8192, 16384), (4096, 32768), (8192, 10000), (4096, 20000)):
    for dt in (torch.float32, torch.float64):
        x = torch.randn((nb, n), dtype=dt, device="cuda"); da = xrft.DataArray(x, ("t", "x"), {"x": np.arange(float(n))})
        for name, fn in (("PS linear+hann", lambda: xrft.power_spectrum(da, dim=["x"], detrend="linear", window="hann")), ("fft", lambda: xrft.fft(da, dim=["x"]))):
            t = timeit(fn)
            path = next(reversed(api._plan_cache.values())).describe().strip().split("\n")[1].strip()[:90]
            print(f"({nb},{n}) {str(dt)[6:]:8s} {name:16s} {x.numel() / t / 1e9:7.1f} GFFT/s  {path}", flush=True)
        del x, da
