"""Two adjacent transform axes with the batch innermost ((y, x, t) arrays, xrfthip_desc.inner): rate of power_spectrum / fft, for rocprofv3."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
for shape in ((1024, 1024, 64), (720, 1440, 32)):
    x = torch.randn(shape, dtype=torch.float32, device="cuda")
    da = xrft.DataArray(x, ("y", "x", "t"), {"y": np.arange(float(shape[0])), "x": np.arange(float(shape[1]))})
    for name, f in (("PS linear+hann", lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")), ("PS plain", lambda: xrft.power_spectrum(da, dim=["y", "x"])),
                    ("fft", lambda: xrft.fft(da, dim=["y", "x"]))):
        f(); f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 5
        print(f"{shape} {name}: {x.numel()/w/1e9:6.1f} GFFT/s {w*1e3:.3f} ms", flush=True)
        print("   ", next(reversed(api._plan_cache.values())).describe().strip().replace("\n", "\n    ")[:1500])
