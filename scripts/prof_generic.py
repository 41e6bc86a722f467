#!/usr/bin/env python3
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
def prof(name, fn, nslab):
    fn(); fn(); torch.cuda.synchronize()
    plan = next(reversed(api._plan_cache.values()))
    plan.set_profiling(True)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    p = plan.read_profile(); plan.set_profiling(False)
    print(name); print(plan.describe().strip())
    for k, (c, ms) in p.items(): print(f"    {k:18s} {ms/3*1e3/nslab:8.1f} us/slab  ({c//3} launches)")
a = torch.randn((16, 2048, 2048), dtype=torch.float32, device="cuda"); c = {"y": np.arange(2048.), "x": np.arange(2048.)}
d1 = xrft.DataArray(a, ("t", "y", "x"), c)
prof("PS 2048^2 f32 linear+hann", lambda: xrft.power_spectrum(d1, dim=["y", "x"], detrend="linear", window="hann"), 16)
x = torch.randn((64, 1440, 720), dtype=torch.float64, device="cuda")
d5 = xrft.DataArray(x, ("t", "lat", "lon"), {"lat": np.arange(1440) * .25, "lon": np.arange(720) * .25})
prof("PS 1440x720 f64 constant+hann", lambda: xrft.power_spectrum(d5, dim=["lat", "lon"], detrend="constant", window="hann"), 64)
x = torch.randn((256, 65536), dtype=torch.float32, device="cuda")
d2 = xrft.DataArray(x, ("t", "x"), {"x": np.arange(65536) * .5})
prof("dft 1-D 65536 f32", lambda: xrft.dft(d2, dim="x"), 256)
