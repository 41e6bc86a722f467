#!/usr/bin/env python3
"""Where the time of the fused radial sums goes (csrc/fasty.h, ISO): per-kernel HIP-event times per slab of isotropic_power_spectrum
(4096^2) and isotropic_cross_spectrum (2048^2) with parts of the radial-sum code switched off (XRFTHIP_YTUNE bits 16-18,
tuning build: scripts/build_tune_yf.sh).  Run on the GPU box: python scripts/tune_iso.py > gpurun_out/tune_iso.txt"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xrft_amd import _lib
_lib.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_dbg", "libxrft_hip_ytune.so"))
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")


def run(fn, nt, tune, reps=5):
    os.environ["XRFTHIP_YTUNE"] = str(tune)
    api.clear_plan_cache()
    r = fn(); r = fn(); torch.cuda.synchronize()
    plan = next(reversed(api._plan_cache.values()))
    plan.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    prof = plan.read_profile()
    plan.set_profiling(False)
    return wall / nt * 1e6, {k: v[1] / reps / nt * 1e3 for k, v in prof.items()}


for name, n, nt, two in (("isotropic_power_spectrum 4096^2", 4096, 32, False), ("isotropic_cross_spectrum 2048^2", 2048, 64, True), ("isotropic_power_spectrum 2048^2", 2048, 64, False)):
    a = torch.randn((nt, n, n), dtype=torch.float32, device="cuda")
    b = torch.randn((nt, n, n), dtype=torch.float32, device="cuda") if two else None
    c = {"y": np.arange(float(n)), "x": np.arange(float(n))}
    d1 = xrft.DataArray(a, ("t", "y", "x"), c)
    d2 = xrft.DataArray(b, ("t", "y", "x"), c) if two else None
    fn = (lambda: xrft.isotropic_cross_spectrum(d1, d2, dim=["y", "x"], window="hann")) if two else (lambda: xrft.isotropic_power_spectrum(d1, dim=["y", "x"], window="hann"))
    full = (lambda: xrft.cross_spectrum(d1, d2, dim=["y", "x"], window="hann")) if two else (lambda: xrft.power_spectrum(d1, dim=["y", "x"], window="hann"))
    print(f"== {name}, {nt} slabs; us per slab")
    for label, tune in (("product", 0), ("no bin-code loads", 1 << 16), ("no sweeps", 1 << 17), ("no partial-table writes", 1 << 18),
                        ("no code loads, no sweeps, no table writes", 7 << 16)):
        wall, k = run(fn, nt, tune)
        print(f"  {label:44s} wall {wall:6.2f} | " + " ".join(f"{kk.replace('fasty_', '')} {v:5.2f}" for kk, v in k.items()), flush=True)
    wall, k = run(full, nt, 0)
    print(f"  {'the full spectrum (stored), for comparison':44s} wall {wall:6.2f} | " + " ".join(f"{kk.replace('fasty_', '')} {v:5.2f}" for kk, v in k.items()), flush=True)
    del a, b, d1, d2
