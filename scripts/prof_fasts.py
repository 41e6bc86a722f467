#!/usr/bin/env python3
"""The one-pass 256 x 256 slab kernel (csrc/fasts.h) against the two-pass pipeline (XRFTHIP_FASTS=0) on (4096, 256, 256) float32 power
spectra: per-kernel HIP-event time and wall time per call; resident workgroups vs one per slab."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
NT = int(os.environ.get("NT", "4096"))


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
    print(f"{name:50s} {ks} || wall {wall * 1e6:.1f} us = {pts / wall / 1e9:.1f} GFFT/s", flush=True)


for ny, nx in ((256, 256), (128, 128), (64, 64), (128, 256), (256, 64)):
    nt = (NT * 256 * 256) // (ny * nx)
    x = torch.randn((nt, ny, nx), dtype=torch.float32, device="cuda") + 2.0
    da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(float(ny)), "x": np.arange(float(nx))})
    pts = x.numel()
    res = {}
    print(f"=== ({nt}, {ny}, {nx}) float32", flush=True)
    for label, env in (("fasts, resident workgroups", {"XRFTHIP_FASTS": "1", "XRFTHIP_FASTS_GRID": "-1"}),
                       ("fasts, one workgroup per slab", {"XRFTHIP_FASTS": "1", "XRFTHIP_FASTS_GRID": "0"}),
                       ("without (two passes: fasty at 256 x 256, the generic tile kernels below)", {"XRFTHIP_FASTS": "0"})):
        os.environ.update(env); api._plan_cache.clear()
        print("---", label, flush=True)
        prof("power_spectrum linear + hann", lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"), pts)
        prof("power_spectrum, no detrend, no window", lambda: xrft.power_spectrum(da, dim=["y", "x"]), pts)
        if "per slab" not in label:
            prof("fft (complex result) linear + hann", lambda: xrft.fft(da, dim=["y", "x"], detrend="linear", window="hann"), pts)
            prof("isotropic_power_spectrum linear + hann", lambda: xrft.isotropic_power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"), pts)
        res[label] = xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann").data[:64].clone()
    ks = list(res)
    print("resident == per-slab launch, bit for bit:", bool(torch.equal(res[ks[0]], res[ks[1]])),
          "| max |fasts - the other path| / max:", float((res[ks[0]] - res[ks[2]]).abs().max() / res[ks[2]].abs().max()), flush=True)
    del x, da, res
