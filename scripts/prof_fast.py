#!/usr/bin/env python3
"""Per-kernel HIP-event timings (us per slab) of the specialised path for several shapes/modes.  Run on the GPU box."""
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
    print(f"{name:34s}", " | ".join(f"{k} {ms/3*1e3/nslab:.1f}" for k, (c, ms) in p.items()), f"|| kernels {tot:.1f} us/slab, wall {wall*1e6/nslab:.1f} us/slab")
for (nt, n) in ((16, 4096), (64, 2048), (128, 1024), (512, 512), (2048, 256)):
    a = torch.randn((nt, n, n), dtype=torch.float32, device="cuda"); c = {"y": np.arange(float(n)), "x": np.arange(float(n))}
    d1 = xrft.DataArray(a, ("t", "y", "x"), c)
    prof(f"PS linear+hann ({nt},{n},{n})", lambda: xrft.power_spectrum(d1, dim=["y", "x"], detrend="linear", window="hann"), nt)
    prof(f"isoPS linear+hann ({nt},{n},{n})", lambda: xrft.isotropic_power_spectrum(d1, dim=["y", "x"], detrend="linear", window="hann"), nt)
    prof(f"fft complex linear+hann ({nt},{n},{n})", lambda: xrft.fft(d1, dim=["y", "x"], detrend="linear", window="hann"), nt)
    b = torch.randn((nt, n, n), dtype=torch.float32, device="cuda"); d2 = xrft.DataArray(b, ("t", "y", "x"), c)
    prof(f"cross hann ({nt},{n},{n})", lambda: xrft.cross_spectrum(d1, d2, dim=["y", "x"], window="hann"), nt)
    prof(f"iso cross hann ({nt},{n},{n})", lambda: xrft.isotropic_cross_spectrum(d1, d2, dim=["y", "x"], window="hann"), nt)
    del a, d1, b, d2
    torch.cuda.empty_cache()
a = torch.randn((16, 4096, 4096), dtype=torch.float32, device="cuda"); c = {"y": np.arange(4096.), "x": np.arange(4096.)}
d1 = xrft.DataArray(a, ("t", "y", "x"), c)
prof("PS real_dim=x linear+hann (16,4096,4096)", lambda: xrft.power_spectrum(d1, dim=["y"], real_dim="x", detrend="linear", window="hann"), 16)
