#!/usr/bin/env python3
"""C2 (1024, 65536) float32 dft / power spectrum and the 256-wide 2-D power spectra: wall GFFT/s and per-kernel us per slab."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api, _lib
if os.environ.get("XRFT_LIB"): _lib.load(os.environ["XRFT_LIB"])
warnings.simplefilter("ignore")
def prof(name, fn, units, pts):
    fn(); fn(); torch.cuda.synchronize()
    plan = [p for p in api._plan_cache.values()][-1]
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 5)
    plan.set_profiling(True)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    p = plan.read_profile(); plan.set_profiling(False)
    print(f"{name:44s}", " | ".join(f"{k} {ms/5*1e3/units:.3f}" for k, (c, ms) in p.items()), f"|| best wall {best*1e6/units:.3f} us/slab = {pts/best/1e9:.1f} GFFT/s", flush=True)
y = torch.randn((1024, 65536), dtype=torch.float32, device="cuda")
db = xrft.DataArray(y, ("t", "x"), {"x": np.arange(65536) * 0.5})
prof("C2 dft f32 (1024,65536)", lambda: xrft.dft(db, dim="x"), 1024, y.numel())
prof("C2 PS f32 (1024,65536)", lambda: xrft.power_spectrum(db, dim="x"), 1024, y.numel())
del y, db
for shp in ((4096, 256, 256), (1024, 512, 256), (1024, 256, 512)):
    x = torch.randn(shp, dtype=torch.float32, device="cuda")
    da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(shp[1]) * 1.0, "x": np.arange(shp[2]) * 1.0})
    prof(f"PS f32 linear hann {shp}", lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"), shp[0], x.numel())
    del x, da
if os.environ.get("BIG"):
    for shp in ((256, 1024, 1024), (64, 2048, 2048), (32, 4096, 4096)):
        x = torch.randn(shp, dtype=torch.float32, device="cuda")
        da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(shp[1]) * 1.0, "x": np.arange(shp[2]) * 1.0})
        prof(f"PS f32 linear hann {shp}", lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"), shp[0], x.numel())
        del x, da
