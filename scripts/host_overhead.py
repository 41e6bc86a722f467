#!/usr/bin/env python3
"""Host time of one call (round 4: the analysis of a labelled array is remembered per label set, csrc untouched): per shape the time the
Python side takes to enqueue a call (perf_counter around the calls, the device still busy) and the time per call with the device
drained, plus a perf_counter breakdown of one cached-plan power_spectrum call on the (4, 256, 256) cube.
Run on the GPU box: python scripts/host_overhead.py > gpurun_out/r04/host_overhead.txt"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api, engine
warnings.simplefilter("ignore")


def run(name, f, pts, n=200):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:64s} host {(t1 - t0) / n * 1e6:7.1f} us per call, with the device drained {(t2 - t0) / n * 1e6:8.1f} us per call = {pts / ((t2 - t0) / n) / 1e9:7.2f} GFFT/s", flush=True)


for shp, dt in (((4, 256, 256), torch.float64), ((64, 1440, 720), torch.float64), ((64, 4096, 4096), torch.float32)):
    x = torch.randn(shp, dtype=dt, device="cuda")
    da = xrft.DataArray(x, ("t", "y", "x"), {"t": np.arange(shp[0]), "y": np.arange(shp[1]) * 1.0, "x": np.arange(shp[2]) * 1.0})
    run(f"power_spectrum linear + hann {shp} {str(dt)[6:]}", lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"),
        x.numel(), 200 if shp[0] == 4 else 20)
    if shp[0] == 4:
        run(f"isotropic_power_spectrum {shp}", lambda: xrft.isotropic_power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"), x.numel())
        # the same call on a NEW labelled array each time (same coordinate objects): what a caller that builds its arrays per call pays
        cs = da.coords
        run("... a new DataArray per call (shared coordinate objects)",
            lambda: xrft.power_spectrum(xrft.DataArray(x, ("t", "y", "x"), cs), dim=["y", "x"], detrend="linear", window="hann"), x.numel())
        # breakdown of the cached call
        f = lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")
        plan = next(reversed(api._plan_cache.values()))
        t = api._to_device(da.data).contiguous()
        n = 500
        t0 = time.perf_counter()
        for _ in range(n):
            plan.execute(t)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"    of which engine.SpectralPlan.execute (output allocation, stream lock, workspace, the C call's launches): {(t1 - t0) / n * 1e6:.1f} us", flush=True)
    del x, da
y = torch.randn((1024, 65536), dtype=torch.float32, device="cuda")
db = xrft.DataArray(y, ("t", "x"), {"x": np.arange(65536) * 0.5 + 3.0})
run("dft (1024, 65536) f32", lambda: xrft.dft(db, dim="x"), y.numel(), 50)
run("fft with true phase (1024, 65536) f32 (a 65536-entry phase table)", lambda: xrft.fft(db, dim="x"), y.numel(), 50)
z = torch.randn((16, 2160, 4320), dtype=torch.float32, device="cuda")
dz = xrft.DataArray(z, ("t", "y", "x"), {"y": np.arange(2160.), "x": np.arange(4320.)})
run("power_spectrum (16, 2160, 4320) f32", lambda: xrft.power_spectrum(dz, dim=["y", "x"], detrend="linear", window="hann"), z.numel(), 20)
run("isotropic_power_spectrum (16, 2160, 4320) f32", lambda: xrft.isotropic_power_spectrum(dz, dim=["y", "x"], detrend="linear", window="hann"), z.numel(), 20)
