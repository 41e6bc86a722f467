import os, sys, time, warnings
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
for shp in ((64, 1001, 1001), (64, 896, 1792), (64, 1430, 1430), (64, 700, 1400), (64, 1232, 1232)):
    x = torch.randn(shp, dtype=torch.float32, device="cuda")
    da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(shp[1]) * 1.0, "x": np.arange(shp[2]) * 1.0})
    fn = lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")
    fn(); fn(); torch.cuda.synchronize()
    plan = [p for p in api._plan_cache.values()][-1]
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3
    d = [l.strip()[17:60] for l in plan.describe().strip().split("\n")[1:3]]
    print(f"PS f32 {shp}: {wall*1e3:.3f} ms = {x.numel()/wall/1e9:.1f} GFFT/s   {d}", flush=True)
    del x, da
