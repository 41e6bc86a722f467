"""The table kernels of fastm.h on their SMALL entries against the run-time-radix kernels with workgroups sized by their points (XRFTHIP_FASTN_TABLES=0)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
shapes = (((1024, 180, 360), torch.float32), ((1024, 192, 384), torch.float32), ((512, 240, 480), torch.float32), ((256, 360, 720), torch.float32), ((256, 320, 640), torch.float32), ((128, 500, 1000), torch.float32),
          ((128, 720, 1440), torch.float32), ((64, 1000, 1000), torch.float32), ((512, 180, 360), torch.float64), ((256, 256, 512), torch.float64), ((128, 360, 720), torch.float64), ((64, 640, 1280), torch.float64), ((64, 1440, 720), torch.float64))
for tab in ("1", "0"):
    os.environ["XRFTHIP_FASTN_TABLES"] = tab
    api._plan_cache.clear()
    print(f"--- XRFTHIP_FASTN_TABLES={tab}")
    for shape, dt in shapes:
        x = torch.randn(shape, dtype=dt, device="cuda")
        da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(float(shape[1])), "x": np.arange(float(shape[2]))})
        f = lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")
        f(); f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 10
        pl = next(reversed(api._plan_cache.values()))
        pl.set_profiling(True); f(); torch.cuda.synchronize(); pr = pl.read_profile(); pl.set_profiling(False)
        d = pl.describe().splitlines()[1]
        print(f"{shape} {str(dt)[-7:]}: {x.numel()/w/1e9:6.1f} GFFT/s {w*1e3:.3f} ms | " + " ".join(f"{k}={v[1]/v[0]*1e3:.0f}us" for k, v in pr.items()) + " | " + d[:100], flush=True)
        del x, da
