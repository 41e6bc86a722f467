#!/usr/bin/env python3
"""fft / power_spectrum along a MIDDLE axis (time-like: not the contiguous one) in place (XRFTHIP_AXIS_Y): per-kernel times."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api, _lib
if os.environ.get("XRFT_LIB"): _lib.load(os.environ["XRFT_LIB"])
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
    print(f"{name:52s}", " | ".join(f"{k} {ms/5:.3f} ms" for k, (c, ms) in p.items()), f"|| wall {wall*1e3:.3f} ms = {pts/wall/1e9:.1f} GFFT/s = {bpp*pts/wall/1e12:.2f} TB/s algorithmic", flush=True)
    print("    ", " / ".join(l.strip()[:150] for l in plan.describe().strip().split("\n")[1:3]))
for shp, dt in (((64, 1024, 2048), torch.float32), ((16, 4096, 2048), torch.float32), ((64, 1000, 2048), torch.float32), ((64, 1024, 2048), torch.float64)):
    x = torch.randn(shp, dtype=dt, device="cuda")
    da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(shp[1]) * 1.0, "x": np.arange(shp[2]) * 1.0})
    es = x.element_size()
    prof(f"fft dim=y {shp} {str(dt)[6:]}", lambda: xrft.fft(da, dim=["y"]), x.numel(), 3 * es)
    prof(f"fft dim=y linear+hann {shp}", lambda: xrft.fft(da, dim=["y"], detrend="linear", window="hann"), x.numel(), 3 * es)
    prof(f"power_spectrum dim=y linear+hann {shp}", lambda: xrft.power_spectrum(da, dim=["y"], detrend="linear", window="hann"), x.numel(), 2 * es)
    del x, da
