"""One transform axis (not the contiguous one): threads per workgroup of fastgy_kernel (XRFTHIP_FASTGY_THR) on short and long axes."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
shapes = (((48, 1024, 1024), torch.float32), ((96, 512, 512), torch.float32), ((97, 1024, 1024), torch.float32), ((150, 512, 512), torch.float32), ((250, 512, 512), torch.float32), ((365, 512, 512), torch.float32),
          ((1250, 256, 256), torch.float32), ((96, 512, 512), torch.float64), ((250, 256, 512), torch.float64), ((365, 256, 512), torch.float64))
for thr in ("256", "128", "64"):
    os.environ["XRFTHIP_FASTGY_THR"] = thr
    api._plan_cache.clear()
    print(f"--- XRFTHIP_FASTGY_THR={thr}")
    for shape, dt in shapes:
        x = torch.randn(shape, dtype=dt, device="cuda")
        da = xrft.DataArray(x, ("time", "y", "x"), {"time": np.arange(float(shape[0]))})
        res = []
        for name, f in (("PS", lambda: xrft.power_spectrum(da, dim="time", detrend="linear", window="hann")), ("fft", lambda: xrft.fft(da, dim="time"))):
            f(); f(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): f()
            torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 5
            res.append(f"{name} {x.numel()/w/1e9:6.1f}")
        d = next(reversed(api._plan_cache.values())).describe().splitlines()[1]
        print(f"{shape} {str(dt)[-7:]}: " + " | ".join(res) + " | " + d[:100], flush=True)
        del x, da
