"""Inverse transforms (xrft.ifft / idft, reference xrft/xrft.py:479-646): rate per shape and the plan that serves them.  python scripts/prof_inverse.py"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
def t(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
def line(name, F, dims, n):
    api._plan_cache.clear()
    w = t(lambda: xrft.ifft(F, dim=dims))
    d = [p.describe().split("\n") for p in api._plan_cache.values()]
    tag = " + ".join((x[1] if len(x) > 1 else x[0]).strip()[:48] for x in d)
    print(f"{name}: ifft {n/w/1e9:6.1f} GFFT/s ({w*1e3:.3f} ms) | {tag}", flush=True)
for shape, dt in (((64, 1024, 1024), "float32"), ((16, 4096, 4096), "float32"), ((64, 1440, 720), "float64"), ((4096, 100, 100), "float32")):
    x = torch.randn(shape, dtype=getattr(torch, dt), device="cuda")
    da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(float(shape[1])), "x": np.arange(float(shape[2]))})
    F = xrft.fft(da, dim=["y", "x"])
    line(f"2-D complex spectrum {shape} {dt}", F, ["freq_y", "freq_x"], x.numel())
    Fr = xrft.fft(da, dim=["y"], real_dim="x")
    api._plan_cache.clear()
    w = t(lambda: xrft.ifft(Fr, dim=["freq_y"], real_dim="freq_x"))
    print(f"   half spectrum back to real (real_dim): {x.numel()/w/1e9:6.1f} GFFT/s | " + " + ".join(p.describe().split("\n")[1].strip()[:48] for p in api._plan_cache.values()), flush=True)
    del x, da, F, Fr
for shape, dt in (((1024, 65536), "float32"), ((131072, 1024), "float32"), ((131072, 250), "float32"), ((16384, 4096), "float32"), ((8192, 6000), "float32"), ((8192, 3000), "float64")):
    x = torch.randn(shape, dtype=getattr(torch, dt), device="cuda")
    da = xrft.DataArray(x, ("t", "x"), {"x": np.arange(float(shape[1]))})
    F = xrft.fft(da, dim=["x"])
    line(f"1-D complex spectrum {shape} {dt}", F, ["freq_x"], x.numel())
    del x, da, F
for shape, dt in (((360, 512, 512), "float32"), ((250, 512, 512), "float32")):
    x = torch.randn(shape, dtype=getattr(torch, dt), device="cuda")
    da = xrft.DataArray(x, ("time", "y", "x"), {"time": np.arange(float(shape[0]))})
    F = xrft.fft(da, dim=["time"])
    line(f"along time {shape} {dt}", F, ["freq_time"], x.numel())
    del x, da, F
