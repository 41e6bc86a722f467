"""Spectra along the FIRST axis ("time") of (time, y, x) arrays on lengths in and outside the tables: rate per shape.  python scripts/prof_time_axis.py"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
def t(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for nt, ny, nx, dt in ((96, 512, 512, "float32"), (120, 512, 512, "float32"), (250, 512, 512, "float32"), (360, 512, 512, "float32"), (500, 256, 512, "float64"), (250, 256, 512, "float64"),
                       (730, 256, 256, "float32"), (1250, 256, 256, "float32"), (1460, 128, 256, "float64"), (3000, 128, 128, "float32"), (48, 1024, 1024, "float32"), (150, 512, 512, "float64"), (365, 512, 512, "float32"), (365, 256, 512, "float64"), (8760, 64, 64, "float32")):
    x = torch.randn((nt, ny, nx), dtype=getattr(torch, dt), device="cuda")
    da = xrft.DataArray(x, ("time", "y", "x"), {"time": np.arange(float(nt))})
    api._plan_cache.clear()
    w = t(lambda: xrft.power_spectrum(da, dim="time", detrend="linear", window="hann"))
    d = next(reversed(api._plan_cache.values())).describe().split("\n")[1][:80]
    w2 = t(lambda: xrft.fft(da, dim="time"))
    w3 = t(lambda: xrft.power_spectrum(da, dim="time"))
    bpp = 8 if dt == "float32" else 16
    print(f"({nt},{ny},{nx}) {dt}: PS linear+hann {x.numel()/w/1e9:6.1f} GFFT/s ({bpp*x.numel()/w/1e12:4.2f} TB/s) | fft {x.numel()/w2/1e9:6.1f} | PS plain {x.numel()/w3/1e9:6.1f} | {d}", flush=True)
