#!/usr/bin/env python3
"""Power spectra on the grid lengths that joined the mixed-radix table late in round 3 (Gaussian and 1/3 ... 1/12-degree grids)."""
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
for shape, dt in (((64, 320, 640), "f32"), ((64, 640, 1280), "f32"), ((64, 640, 1280), "f64"), ((64, 1280, 2560), "f32"), ((32, 1080, 2160), "f32"), ((32, 1080, 2160), "f64"), ((32, 540, 1080), "f64"),
                  ((32, 1440, 2880), "f32"), ((16, 2160, 4320), "f32"), ((16, 2160, 2160), "f64")):
    x = torch.randn(shape, dtype=torch.float32 if dt == "f32" else torch.float64, device="cuda")
    da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(float(shape[1])), "x": np.arange(float(shape[2]))})
    for name, fn in (("PS linear+hann", lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")), ("isotropic PS", lambda: xrft.isotropic_power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"))):
        t = timeit(fn)
        path = next(reversed(api._plan_cache.values())).describe().strip().split("\n")[1].strip().split("]")[0] + "]"
        print(f"{str(shape):20s} {dt} {name:16s} {x.numel() / t / 1e9:7.1f} GFFT/s  {path}", flush=True)
    del x, da
