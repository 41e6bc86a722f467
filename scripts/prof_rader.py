"""Round 5: one transform axis (not the contiguous one) on lengths with ONE prime factor 17 ... 127 -- a year of daily / 6-hourly samples (365 = 5 x 73, 1460 = 20 x 73),
a leap year (366 = 6 x 61) -- the prime-factor form with Rader's algorithm along the prime (fastg.h) against the chirp convolution (XRFTHIP_RADER=0) and a smooth neighbour."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
shapes = (((365, 512, 512), torch.float32), ((364, 512, 512), torch.float32), ((366, 512, 512), torch.float32), ((730, 256, 512), torch.float32), ((1460, 128, 256), torch.float64),
          ((1440, 128, 256), torch.float64), ((1460, 256, 256), torch.float32), ((365, 256, 512), torch.float64), ((97, 1024, 1024), torch.float32), ((2920, 128, 128), torch.float32))
for rader in ("1", "0"):
    os.environ["XRFTHIP_RADER"] = rader
    api._plan_cache.clear()
    print(f"--- XRFTHIP_RADER={rader}")
    for shape, dt in shapes:
        x = torch.randn(shape, dtype=dt, device="cuda")
        da = xrft.DataArray(x, ("time", "y", "x"), {"time": np.arange(float(shape[0]))})
        for name, f in (("PS linear+hann", lambda: xrft.power_spectrum(da, dim="time", detrend="linear", window="hann")), ("fft", lambda: xrft.fft(da, dim="time"))):
            f(); f(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): f()
            torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 5
            d = next(reversed(api._plan_cache.values())).describe()
            tag = "Rader" if "Rader" in d else "Bluestein" if "Bluestein" in d else "smooth"
            print(f"{shape} {'f32' if dt == torch.float32 else 'f64'} {name}: {x.numel()/w/1e9:6.1f} GFFT/s {w*1e3:.3f} ms [{tag}]", flush=True)
        del x, da
