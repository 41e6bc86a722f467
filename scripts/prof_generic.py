"""Where the time goes on slabs that take the generic tile kernels (lengths outside the tables of fastm.h): per-kernel HIP-event times of
the plans a power_spectrum call runs.  python scripts/prof_generic.py [shape ...] on the GPU box"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")

shapes = [(16, 3000, 3000, "float64"), (16, 2500, 1250, "float32"), (32, 750, 1500, "float64"), (16, 1215, 1215, "float32")]
for nt, ny, nx, dt in shapes:
    x = torch.randn((nt, ny, nx), dtype=getattr(torch, dt), device="cuda")
    da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(float(ny)), "x": np.arange(float(nx))})
    api._plan_cache.clear()
    f = lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")
    f(); f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 5
    print(f"({nt},{ny},{nx}) {dt}: {x.numel()/w/1e9:6.1f} GFFT/s  {w*1e6/nt:8.1f} us/slab", flush=True)
    for pl in api._plan_cache.values():
        pl.set_profiling(True)
    f(); torch.cuda.synchronize()
    for pl in api._plan_cache.values():
        print("   ", pl.describe().strip().replace("\n", "\n    "))
        for k, (n, ms) in pl.read_profile().items():
            print(f"       {k:28s} x{n:3d} {ms*1e3/nt:9.2f} us/slab")
