#!/usr/bin/env python3
"""Throughput of common calls that take the generic tile kernels (what is NOT on a specialised path): wall GFFT/s, per-kernel ms."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
def prof(name, fn, pts, bpp):
    fn(); fn(); torch.cuda.synchronize()
    plan = [p for p in api._plan_cache.values()][-1]
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5
    plan.set_profiling(True)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    p = plan.read_profile(); plan.set_profiling(False)
    path = plan.describe().strip().split("\n")[1].strip()[:60]
    print(f"{name:58s}", " | ".join(f"{k} {ms/5:.3f}" for k, (c, ms) in p.items()), f"|| {wall*1e3:.3f} ms = {pts/wall/1e9:.1f} GFFT/s = {bpp*pts/wall/1e12:.2f} TB/s alg.  {path}", flush=True)
for shp, dt in (((131072, 1024), torch.float32), ((131072, 1000), torch.float32), ((65536, 1024), torch.float64), ((32768, 4096), torch.float32), ((4096, 32768), torch.float32)):
    x = torch.randn(shp, dtype=dt, device="cuda"); es = x.element_size()
    da = xrft.DataArray(x, ("t", "x"), {"x": np.arange(shp[1]) * 1.0})
    prof(f"1-D PS last axis linear+hann {shp} {str(dt)[6:]}", lambda: xrft.power_spectrum(da, dim=["x"], detrend="linear", window="hann"), x.numel(), 2 * es)
    prof(f"1-D fft last axis {shp}", lambda: xrft.fft(da, dim=["x"]), x.numel(), 3 * es)
    del x, da
for shp, dt in (((64, 1000, 1000), torch.float32), ((64, 721, 1440), torch.float32), ((64, 2000, 2000), torch.float32), ((16, 3000, 3000), torch.float64)):
    x = torch.randn(shp, dtype=dt, device="cuda"); es = x.element_size()
    da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(shp[1]) * 1.0, "x": np.arange(shp[2]) * 1.0})
    prof(f"2-D PS linear+hann {shp} {str(dt)[6:]}", lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"), x.numel(), 2 * es)
    del x, da
