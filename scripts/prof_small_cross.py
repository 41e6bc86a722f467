"""Cross spectra of float64 small slabs whose two tiles do not fit the LDS of one workgroup (verdict r4, weak #6): now on the two-pass lengths-as-data pipeline (csrc/fastn.h)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
def rate(f, n, reps=5):
    f(); f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return n / ((time.perf_counter() - t0) / reps) / 1e9
for shape, dt in (((4096, 100, 100), "float64"), ((4096, 81, 81), "float64"), ((4096, 100, 100), "float32"), ((2048, 150, 150), "float64"), ((1024, 180, 180), "float64"), ((2048, 125, 125), "float32")):
    a = torch.randn(shape, dtype=getattr(torch, dt), device="cuda"); b = torch.randn(shape, dtype=getattr(torch, dt), device="cuda")
    c = {"y": np.arange(float(shape[1])), "x": np.arange(float(shape[2]))}
    d1, d2 = xrft.DataArray(a, ("t", "y", "x"), c), xrft.DataArray(b, ("t", "y", "x"), c)
    for name, f in (("cross_spectrum", lambda: xrft.cross_spectrum(d1, d2, dim=["y", "x"], detrend="linear", window="hann")),
                    ("isotropic_cross_spectrum", lambda: xrft.isotropic_cross_spectrum(d1, d2, dim=["y", "x"], window="hann")),
                    ("power_spectrum", lambda: xrft.power_spectrum(d1, dim=["y", "x"], detrend="linear", window="hann"))):
        api._plan_cache.clear()
        r = rate(f, a.numel())
        tag = " + ".join(p.describe().split("\n")[1].strip()[:70] for p in api._plan_cache.values())
        print(f"{shape} {dt} {name}: {r:6.1f} GFFT/s | {tag}", flush=True)
