#!/usr/bin/env python3
"""Where the time of a detrended spectrum over (y, x) of a (y, x, t) array goes: the stand-alone detrend, and the spectrum with / without detrend and window."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
warnings.simplefilter("ignore")
def timeit(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
for shape in ((1024, 1024, 64), (2048, 2048, 16)):
    x = torch.randn(shape, dtype=torch.float32, device="cuda")
    da = xrft.DataArray(x, ("y", "x", "t"), {"y": np.arange(float(shape[0])), "x": np.arange(float(shape[1]))})
    mb = x.numel() * 4 / 1e6
    for name, fn in (("detrend linear alone", lambda: xrft.detrend(da, ["y", "x"], "linear")), ("detrend constant alone", lambda: xrft.detrend(da, ["y", "x"], "constant")),
                     ("PS", lambda: xrft.power_spectrum(da, dim=["y", "x"])), ("PS hann", lambda: xrft.power_spectrum(da, dim=["y", "x"], window="hann")),
                     ("PS linear", lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear")), ("PS constant", lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="constant"))):
        t = timeit(fn)
        print(f"{str(shape):18s} {name:24s} {t * 1e3:.3f} ms  ({mb:.0f} MB array: one read + one write at 5.3 TB/s = {2 * mb / 5.3e3:.3f} ms)", flush=True)
