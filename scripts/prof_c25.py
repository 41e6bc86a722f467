#!/usr/bin/env python3
"""Per-kernel HIP-event timings of the generic-path configurations: C5 (64,1440,720) f64 PS and C2 (1024,65536) f32 dft."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
def prof(name, fn, units, pts):
    fn(); fn(); torch.cuda.synchronize()
    plan = [p for p in api._plan_cache.values()][-1]
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3
    plan.set_profiling(True)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    p = plan.read_profile(); plan.set_profiling(False)
    tot = sum(ms for c, ms in p.values()) / 3 * 1e3 / units
    print(f"{name:36s}", " | ".join(f"{k} {ms/3*1e3/units:.2f}" for k, (c, ms) in p.items()), f"|| kernels {tot:.2f} us/unit, wall {wall*1e6/units:.2f} us/unit = {pts/wall/1e9:.1f} GFFT/s", flush=True)
    print("   ", plan.describe().strip().split("\n")[1:4])
x = torch.randn((64, 1440, 720), dtype=torch.float64, device="cuda")
da = xrft.DataArray(x, ("t", "lat", "lon"), {"lat": np.arange(1440) * .25, "lon": np.arange(720) * .25})
for det in (("linear",) if os.environ.get("C5_ONLY_LINEAR", "1") == "1" and len(sys.argv) > 1 else (None, "constant", "linear")):
    prof(f"C5 PS f64 {det} hann (64,1440,720)", lambda: xrft.power_spectrum(da, dim=["lat", "lon"], detrend=det, window="hann"), 64, x.numel())
x32 = x.float(); da32 = xrft.DataArray(x32, da.dims, da.coords)
prof("C5-shape PS f32 linear hann", lambda: xrft.power_spectrum(da32, dim=["lat", "lon"], detrend="linear", window="hann"), 64, x.numel())
del x, x32
y = torch.randn((1024, 65536), dtype=torch.float32, device="cuda")
db = xrft.DataArray(y, ("t", "x"), {"x": np.arange(65536) * 0.5})
prof("C2 dft f32 (1024,65536)", lambda: xrft.dft(db, dim="x"), 1024, y.numel())
prof("C2 PS f32 (1024,65536)", lambda: xrft.power_spectrum(db, dim="x"), 1024, y.numel())
