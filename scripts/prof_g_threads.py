"""fastg_kernel (small slabs in one pass; groups of rows of a 1-D transform): threads per workgroup (XRFTHIP_FASTG_THREADS; 0 = the plan's own choice)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
one = (((65536, 96), torch.float32), ((131072, 250), torch.float32), ((65536, 50), torch.float64), ((65536, 182), torch.float32), ((16384, 1250), torch.float64))
two = (((14400, 50, 50), torch.float32), ((14400, 50, 50), torch.float64), ((8192, 96, 96), torch.float32), ((2048, 150, 150), torch.float32), ((14400, 45, 45), torch.float32), ((4096, 72, 144), torch.float32))
for thr in ("0", "64", "128", "256", "512"):
    os.environ["XRFTHIP_FASTG_THREADS"] = thr
    api._plan_cache.clear()
    print(f"--- XRFTHIP_FASTG_THREADS={thr}")
    for shape, dt in one + two:
        x = torch.randn(shape, dtype=dt, device="cuda")
        if len(shape) == 2:
            da = xrft.DataArray(x, ("s", "x"), {"x": np.arange(float(shape[1]))}); dim = ["x"]
        else:
            da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(float(shape[1])), "x": np.arange(float(shape[2]))}); dim = ["y", "x"]
        res = []
        for name, f in (("PS", lambda: xrft.power_spectrum(da, dim=dim, detrend="linear", window="hann")), ("fft", lambda: xrft.fft(da, dim=dim))):
            f(); f(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): f()
            torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 5
            res.append(f"{name} {x.numel()/w/1e9:6.1f}")
        d = next(reversed(api._plan_cache.values())).describe().splitlines()[1]
        print(f"{shape} {str(dt)[-7:]}: " + " | ".join(res) + " | " + d[:90], flush=True)
        del x, da
