#!/usr/bin/env python3
"""Per-kernel HIP-event timings of C5 (64, 1440, 720) float64 on the mixed-radix y-first kernels (csrc/fastm.h): power spectrum
with each detrend, fft, cross spectrum; XRFTHIP_FASTM=0 in the environment times the generic tile kernels instead."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api, _lib
if os.environ.get("XRFT_LIB"): _lib.load(os.environ["XRFT_LIB"])  # an ablation build (scripts/build_ablate_m.sh)
warnings.simplefilter("ignore")
def prof(name, fn, units, pts):
    fn(); fn(); torch.cuda.synchronize()
    plan = [p for p in api._plan_cache.values()][-1]
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5
    plan.set_profiling(True)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    p = plan.read_profile(); plan.set_profiling(False)
    tot = sum(ms for c, ms in p.values()) / 5 * 1e3 / units
    print(f"{name:40s}", " | ".join(f"{k} {ms/5*1e3/units:.2f}" for k, (c, ms) in p.items()), f"|| kernels {tot:.2f} us/slab, wall {wall*1e6/units:.2f} us/slab = {pts/wall/1e9:.1f} GFFT/s", flush=True)
    return plan
shapes = [(64, 1440, 720)] + ([(64, 720, 1440), (32, 1440, 1440), (256, 360, 360)] if len(sys.argv) > 1 else [])
for shp in shapes:
    fdt = torch.float32 if os.environ.get("C5_F32") else torch.float64
    x = torch.randn(shp, dtype=fdt, device="cuda")
    y = torch.randn(shp, dtype=fdt, device="cuda")
    c = {"lat": np.arange(shp[1]) * .25, "lon": np.arange(shp[2]) * .25}
    da = xrft.DataArray(x, ("t", "lat", "lon"), c); db = xrft.DataArray(y, ("t", "lat", "lon"), c)
    for det in ((None, "linear") if not os.environ.get("ONLY_LINEAR") else ("linear",)):
        pl = prof(f"PS {'f32' if os.environ.get('C5_F32') else 'f64'} {det} hann {shp}", lambda: xrft.power_spectrum(da, dim=["lat", "lon"], detrend=det, window="hann"), shp[0], x.numel())
    print("   ", pl.describe().strip().split("\n")[1][:300])
    if os.environ.get("ONLY_LINEAR"): continue
    prof(f"fft {'f32' if os.environ.get('C5_F32') else 'f64'} linear hann {shp}", lambda: xrft.fft(da, dim=["lat", "lon"], detrend="linear", window="hann"), shp[0], x.numel())
    prof(f"isotropic PS {'f32' if os.environ.get('C5_F32') else 'f64'} linear hann {shp}", lambda: xrft.isotropic_power_spectrum(da, dim=["lat", "lon"], detrend="linear", window="hann"), shp[0], x.numel())
    prof(f"cross {'f32' if os.environ.get('C5_F32') else 'f64'} linear hann {shp}", lambda: xrft.cross_spectrum(da, db, dim=["lat", "lon"], detrend="linear", window="hann"), shp[0], x.numel())
    del x, y, da, db
    torch.cuda.empty_cache()
