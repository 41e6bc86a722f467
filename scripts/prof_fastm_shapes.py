#!/usr/bin/env python3
"""Per-kernel HIP-event times of power_spectrum(linear, hann) on mixed-radix shapes: python scripts/prof_fastm_shapes.py nt,ny,nx,f32 ..."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
for spec in sys.argv[1:]:
    nt, ny, nx, dt = spec.split(","); nt, ny, nx = int(nt), int(ny), int(nx)
    x = torch.randn((nt, ny, nx), dtype=torch.float32 if dt == "f32" else torch.float64, device="cuda")
    da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(float(ny)), "x": np.arange(float(nx))})
    fn = lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")
    fn(); fn(); torch.cuda.synchronize()
    plan = next(reversed(api._plan_cache.values()))
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5
    plan.set_profiling(True)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    p = plan.read_profile(); plan.set_profiling(False)
    mb = ny * nx * (4 if dt == "f32" else 8) / 1e6
    print(f"{spec:22s} " + " | ".join(f"{k} {ms / 5 * 1e3 / nt:.2f}" for k, (c, ms) in p.items()) + f" || wall {wall * 1e6 / nt:.2f} us/slab = {x.numel() / wall / 1e9:.1f} GFFT/s; slab {mb:.1f} MB: each pass moves {2 * mb:.1f} MB = {2 * mb / 5.3:.2f} us at 5.3 TB/s", flush=True)
    print("    " + plan.describe().strip().split("\n")[1][:260])
    del x, da
