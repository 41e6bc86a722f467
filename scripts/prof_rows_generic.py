"""1-D spectra along the last axis on lengths outside the tables (generic row tiles): rate per shape.  python scripts/prof_rows_generic.py"""
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
for nt, nx, dt in ((262144, 100, "float32"), (262144, 100, "float64"), (131072, 250, "float32"), (65536, 360, "float64"), (65536, 750, "float32"), (32768, 1000, "float32"), (16384, 1250, "float64"),
                   (16384, 3000, "float32"), (8192, 3000, "float64"), (8192, 6000, "float32"), (131072, 125, "float32"), (65536, 243, "float32"), (32768, 729, "float64"), (65536, 96, "float32"), (65536, 50, "float64")):
    x = torch.randn((nt, nx), dtype=getattr(torch, dt), device="cuda")
    da = xrft.DataArray(x, ("t", "x"), {"x": np.arange(float(nx))})
    api._plan_cache.clear()
    w = t(lambda: xrft.power_spectrum(da, dim="x", detrend="linear", window="hann"))
    d = next(reversed(api._plan_cache.values())).describe().split("\n")[1][:70]
    w2 = t(lambda: xrft.fft(da, dim="x"))
    w3 = t(lambda: xrft.power_spectrum(da, dim="x", real_dim="x"))
    bpp = 8 if dt == "float32" else 16
    print(f"({nt},{nx}) {dt}: PS linear+hann {x.numel()/w/1e9:6.1f} GFFT/s ({bpp*x.numel()/w/1e12:4.2f} TB/s) | fft {x.numel()/w2/1e9:6.1f} | PS real_dim {x.numel()/w3/1e9:6.1f} | {d}", flush=True)
