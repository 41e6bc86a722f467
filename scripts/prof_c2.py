#!/usr/bin/env python3
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
x = torch.randn((1024, 65536), dtype=torch.float32, device="cuda"); da = xrft.DataArray(x, ("t", "x"), {"x": np.arange(65536) * 0.5})
def prof(name, fn):
    fn(); fn(); torch.cuda.synchronize()
    plan = [p for p in api._plan_cache.values()][-1]
    plan.set_profiling(True)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    p = plan.read_profile(); plan.set_profiling(False)
    print(name, " | ".join(f"{k} {ms/3*1e3/1024:.3f}" for k, (c, ms) in p.items()), "||", " ; ".join(l.strip() for l in plan.describe().split("\n")[1:] if l.strip()))
prof("dft", lambda: xrft.dft(da, dim="x"))
prof("ps hann linear", lambda: xrft.power_spectrum(da, dim="x", detrend="linear", window="hann"))
