"""power_spectrum along a non-contiguous axis with real_dim (half output): today through a transposed copy; against the full-spectrum call that runs where the axis lies."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
for shape, dt in (((360, 512, 512), torch.float32), ((365, 512, 512), torch.float32), ((250, 512, 512), torch.float32), ((1024, 256, 512), torch.float32), ((1440, 128, 256), torch.float64)):
    x = torch.randn(shape, dtype=dt, device="cuda")
    da = xrft.DataArray(x, ("time", "y", "x"), {"time": np.arange(float(shape[0]))})
    res = []
    for name, f in (("full", lambda: xrft.power_spectrum(da, dim="time", detrend="linear", window="hann")), ("real_dim", lambda: xrft.power_spectrum(da, dim="time", real_dim="time", detrend="linear", window="hann"))):
        f(); f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 5
        d = next(reversed(api._plan_cache.values())).describe().splitlines()[1]
        res.append(f"{name} {x.numel()/w/1e9:6.1f} GFFT/s {w*1e3:.3f} ms {d.strip()[:28]}")
    print(f"{shape} {str(dt)[-7:]}: " + " | ".join(res), flush=True)
    del x, da
