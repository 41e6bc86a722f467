"""Lengths with prime factors 7 / 11 / 13 on the one-pass kernels (csrc/fastg.h; round 5: butterflies instead of Bluestein) and on the two-pass pipeline (csrc/fastn.h).
XRFTHIP_PRIME_BUTTERFLIES=0 is not a knob of the library: the 'before' numbers are those of profiles/r04_time_axis.txt / r04_small_slabs.txt for the same shapes where they exist."""
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
for shape, dims, dim, dt in [((364, 512, 512), ("time", "y", "x"), ["time"], "float32"), ((728, 256, 512), ("time", "y", "x"), ["time"], "float32"), ((91, 512, 512), ("time", "y", "x"), ["time"], "float32"),
                             ((1001, 128, 256), ("time", "y", "x"), ["time"], "float64"), ((364, 256, 512), ("time", "y", "x"), ["time"], "float64"), ((365, 512, 512), ("time", "y", "x"), ["time"], "float32"),
                             ((65536, 182), ("s", "x"), ["x"], "float32"), ((16384, 1001), ("s", "x"), ["x"], "float32"), ((8192, 77, 77), ("t", "y", "x"), ["y", "x"], "float32"),
                             ((4096, 98, 98), ("t", "y", "x"), ["y", "x"], "float32"), ((2048, 154, 154), ("t", "y", "x"), ["y", "x"], "float64")]:
    x = torch.randn(shape, dtype=getattr(torch, dt), device="cuda")
    da = xrft.DataArray(x, dims, {d: np.arange(float(n)) for d, n in zip(dims, shape)})
    api._plan_cache.clear()
    r = rate(lambda: xrft.power_spectrum(da, dim=dim, detrend="linear", window="hann"), x.numel())
    tag = " + ".join(p.describe().split("\n")[1].strip()[:90] for p in api._plan_cache.values())
    print(f"{shape} {dt} dim={dim}: {r:6.1f} GFFT/s | {tag}", flush=True)
    del x, da
