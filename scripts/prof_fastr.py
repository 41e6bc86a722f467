#!/usr/bin/env python3
"""The one-pass 65536-sample row kernel (csrc/fastr.h) on BASELINE.json configs[1], (1024, 65536) float32: per-kernel HIP-event time and
wall time per call for dft / fft with true phase / power spectra, one workgroup per row and resident sets of workgroups
(XRFTHIP_FASTR_GRID), against the two four-step passes (XRFTHIP_FASTR=0).  Plans read the environment when they are created, so the
plan cache is cleared between settings."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
NT = int(os.environ.get("NT", "1024"))


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
    ktot = sum(ms / c for c, ms in p.values()) * 1e-3
    print(f"{name:58s} {ks} || wall {wall * 1e6:.1f} us = {pts / wall / 1e9:.1f} GFFT/s (kernels alone {pts / ktot / 1e9:.1f})", flush=True)


y = torch.randn((NT, 65536), dtype=torch.float32, device="cuda") + 3.0
db = xrft.DataArray(y, ("t", "x"), {"x": np.arange(65536) * 0.5 + 7.0})
pts = y.numel()
settings = [("fastr, one workgroup per row", {"XRFTHIP_FASTR": "1", "XRFTHIP_FASTR_GRID": "0"}),
            ("fastr, 256 resident workgroups", {"XRFTHIP_FASTR": "1", "XRFTHIP_FASTR_GRID": "256"}),
            ("fastr, 512 workgroups", {"XRFTHIP_FASTR": "1", "XRFTHIP_FASTR_GRID": "512"}),
            ("four-step (two passes)", {"XRFTHIP_FASTR": "0"})]
for label, env in settings:
    os.environ.update(env)
    api._plan_cache.clear()
    print(f"--- {label}", flush=True)
    prof("dft (1024, 65536) f32", lambda: xrft.dft(db, dim="x"), pts)
    prof("fft, true phase (table multiply on the way out)", lambda: xrft.fft(db, dim="x"), pts)
    prof("power_spectrum", lambda: xrft.power_spectrum(db, dim="x"), pts)
    prof("power_spectrum, linear detrend + hann", lambda: xrft.power_spectrum(db, dim="x", detrend="linear", window="hann"), pts)
    if env.get("XRFTHIP_FASTR") == "1" and env.get("XRFTHIP_FASTR_GRID") == "0":
        prof("power_spectrum, real_dim (half output)", lambda: xrft.power_spectrum(db, dim=["x"], real_dim="x"), pts)
        prof("dft, linear detrend + hann", lambda: xrft.dft(db, dim="x", detrend="linear", window="hann"), pts)
# parity of the resident form against the one-workgroup-per-row form, bit for bit
os.environ.update({"XRFTHIP_FASTR": "1", "XRFTHIP_FASTR_GRID": "0"}); api._plan_cache.clear()
a = xrft.dft(db, dim="x").data.clone()
os.environ["XRFTHIP_FASTR_GRID"] = "256"; api._plan_cache.clear()
b = xrft.dft(db, dim="x").data
print("resident == per-row launch, bit for bit:", bool(torch.equal(a, b)))
os.environ["XRFTHIP_FASTR"] = "0"; api._plan_cache.clear()
c = xrft.dft(db, dim="x").data
print("max |fastr - four-step| / max|.|:", float((a - c).abs().max() / c.abs().max()))
