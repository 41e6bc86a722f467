#!/usr/bin/env python3
"""Per-pass timings of the generic tile kernels on the configurations the specialised path does not take."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
def prof(name, fn, nslab):
    fn(); fn(); torch.cuda.synchronize()
    plan = [p for p in api._plan_cache.values()][-1]
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3
    plan.set_profiling(True)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    p = plan.read_profile(); plan.set_profiling(False)
    tot = sum(ms for c, ms in p.values()) / 3 * 1e3 / nslab
    print(f"{name:30s}", " | ".join(f"{k} {ms/3*1e3/nslab:.2f}" for k, (c, ms) in p.items()), f"|| kernels {tot:.2f} us/slab, wall {wall*1e6/nslab:.2f} us/slab")
    print("\n".join("      " + l for l in plan.describe().split("\n")[1:] if l.strip()))
x = torch.randn((64, 1440, 720), dtype=torch.float64, device="cuda")
da = xrft.DataArray(x, ("t", "lat", "lon"), {"lat": np.arange(1440) * .25, "lon": np.arange(720) * .25})
prof("C5 PS (64,1440,720) f64", lambda: xrft.power_spectrum(da, dim=["lat", "lon"], detrend="constant", window="hann"), 64)
x32 = x.float(); da32 = xrft.DataArray(x32, ("t", "lat", "lon"), {"lat": np.arange(1440) * .25, "lon": np.arange(720) * .25})
prof("   same in f32", lambda: xrft.power_spectrum(da32, dim=["lat", "lon"], detrend="constant", window="hann"), 64)
del x, x32, da, da32
x = torch.randn((1024, 65536), dtype=torch.float32, device="cuda"); da = xrft.DataArray(x, ("t", "x"), {"x": np.arange(65536) * 0.5})
prof("C2 dft (1024,65536) f32", lambda: xrft.dft(da, dim="x"), 1024)
x = torch.randn((16, 2048, 2048), dtype=torch.float64, device="cuda"); c = {"y": np.arange(2048.), "x": np.arange(2048.)}
da = xrft.DataArray(x, ("t", "y", "x"), c)
prof("PS (16,2048,2048) f64", lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"), 16)
