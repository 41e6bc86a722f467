"""1-D spectra along the CONTIGUOUS axis on lengths with a prime factor that has no butterfly (365, 730, 1460; 73 x 144 boxes)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
for shape, dt in (((131072, 365), torch.float32), ((131072, 364), torch.float32), ((65536, 730), torch.float32), ((32768, 1460), torch.float64), ((32768, 1440), torch.float64), ((262144, 73), torch.float32)):
    x = torch.randn(shape, dtype=dt, device="cuda")
    da = xrft.DataArray(x, ("s", "time"), {"time": np.arange(float(shape[1]))})
    for name, f in (("PS linear+hann", lambda: xrft.power_spectrum(da, dim="time", detrend="linear", window="hann")), ("fft", lambda: xrft.fft(da, dim="time"))):
        f(); f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 5
        d = next(reversed(api._plan_cache.values())).describe().splitlines()
        print(f"{shape} {'f32' if dt == torch.float32 else 'f64'} {name}: {x.numel()/w/1e9:6.1f} GFFT/s {w*1e3:.3f} ms | {d[1][:150] if len(d) > 1 else d}", flush=True)
    del x, da
for shape, dt in (((4096, 73, 144), torch.float32), ((4096, 72, 144), torch.float32), ((1024, 181, 360), torch.float32), ((1024, 241, 480), torch.float32), ((2048, 94, 192), torch.float32)):
    x = torch.randn(shape, dtype=dt, device="cuda")
    da = xrft.DataArray(x, ("t", "lat", "lon"), {"lat": np.arange(float(shape[1])), "lon": np.arange(float(shape[2]))})
    f = lambda: xrft.power_spectrum(da, dim=["lat", "lon"], detrend="linear", window="hann")
    f(); f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 5
    d = next(reversed(api._plan_cache.values())).describe().splitlines()
    print(f"{shape} f32 PS 2-D: {x.numel()/w/1e9:6.1f} GFFT/s {w*1e3:.3f} ms | {d[1][:170]}", flush=True)
    del x, da
