#!/usr/bin/env python3
"""The one-pass row kernels (csrc/fastr.h) on rows of 8192 ... 65536 samples, 64M samples per cube, against what ran before (XRFTHIP_FASTR=0:
the generic passes below 65536)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")


def prof(name, fn, pts, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    plan = [p for p in api._plan_cache.values()][-1]
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps
    plan.set_profiling(True)
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    p = plan.read_profile(); plan.set_profiling(False)
    ks = " | ".join(f"{k} {ms / c * 1e3:.1f} us" for k, (c, ms) in p.items())
    print(f"{name:46s} {ks} || wall {wall * 1e6:.1f} us = {pts / wall / 1e9:.1f} GFFT/s", flush=True)


for n in (4096, 8192, 16384, 32768, 65536):
    nt = (1 << 26) // n
    y = torch.randn((nt, n), dtype=torch.float32, device="cuda") + 3.0
    db = xrft.DataArray(y, ("t", "x"), {"x": np.arange(n) * 0.5 + 7.0})
    pts = y.numel()
    print(f"=== ({nt}, {n}) float32", flush=True)
    for label, env in (("one pass (fastr)", {"XRFTHIP_FASTR": "1"}), ("before", {"XRFTHIP_FASTR": "0"})):
        os.environ.update(env); api._plan_cache.clear()
        print("---", label, flush=True)
        prof("dft", lambda: xrft.dft(db, dim="x"), pts)
        prof("power_spectrum", lambda: xrft.power_spectrum(db, dim="x"), pts)
        prof("power_spectrum, linear detrend + hann", lambda: xrft.power_spectrum(db, dim="x", detrend="linear", window="hann"), pts)
    del y, db
