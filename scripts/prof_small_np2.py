"""Small real slabs that are not 64 | 128 | 256 points per axis: the one-pass LDS kernel (csrc/fastg.h) against the two-pass generic kernels
(XRFTHIP_FASTG=0), and the workgroup size (XRFTHIP_FASTG_THREADS).  Run on the GPU box: python scripts/prof_small_np2.py"""
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


shapes = ((14400, 50, 50, torch.float32), (14400, 50, 50, torch.float64), (8192, 96, 96, torch.float32), (4096, 100, 100, torch.float32), (4096, 100, 100, torch.float64),
          (2048, 150, 150, torch.float32), (1024, 192, 192, torch.float32), (2048, 120, 240, torch.float32), (4096, 128, 128, torch.float64), (8192, 64, 64, torch.float64),
          (4096, 90, 180, torch.float32), (8192, 60, 60, torch.float32), (8192, 80, 80, torch.float64), (1024, 180, 180, torch.float32), (2048, 144, 96, torch.float64),
          (14400, 45, 45, torch.float32), (4096, 75, 75, torch.float32), (4096, 81, 81, torch.float64), (2048, 125, 125, torch.float32))
for nt, ny, nx, dt in shapes:
    x = torch.randn((nt, ny, nx), dtype=dt, device="cuda")
    da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(float(ny)), "x": np.arange(float(nx))})
    line = f"({nt},{ny},{nx}) {str(dt)[6:]}:"
    for label, env in (("two-pass", {"XRFTHIP_FASTG": "0"}), ("fastg", {}), ("128thr", {"XRFTHIP_FASTG_THREADS": "128"}),
                       ("256thr", {"XRFTHIP_FASTG_THREADS": "256"}), ("512thr", {"XRFTHIP_FASTG_THREADS": "512"}), ("1024thr", {"XRFTHIP_FASTG_THREADS": "1024"})):
        for k in ("XRFTHIP_FASTG", "XRFTHIP_FASTG_THREADS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        api._plan_cache.clear()
        w = t(lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"))
        on = "[fastg]" in next(reversed(api._plan_cache.values())).describe()
        line += f" {label}{'*' if on else ''} {x.numel()/w/1e9:6.1f}"
        if label == "fastg" and not on:
            break
    for k in ("XRFTHIP_FASTG", "XRFTHIP_FASTG_THREADS"):
        os.environ.pop(k, None)
    api._plan_cache.clear()
    w2 = t(lambda: xrft.fft(da, dim=["y", "x"], detrend="linear", window="hann"))
    w3 = t(lambda: xrft.isotropic_power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"))
    db = xrft.DataArray(torch.roll(x, 3, dims=2) * 0.5, ("t", "y", "x"), {"y": np.arange(float(ny)), "x": np.arange(float(nx))})
    w5 = t(lambda: xrft.cross_spectrum(da, db, dim=["y", "x"], detrend="linear", window="hann"))
    on5 = any("[fastg cross" in p.describe() for p in api._plan_cache.values())
    w7 = t(lambda: xrft.isotropic_cross_spectrum(da, db, dim=["y", "x"], detrend="linear", window="hann"))
    os.environ["XRFTHIP_FASTG"] = "0"
    api._plan_cache.clear()
    w4 = t(lambda: xrft.isotropic_power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"))
    w6 = t(lambda: xrft.cross_spectrum(da, db, dim=["y", "x"], detrend="linear", window="hann"))
    w8 = t(lambda: xrft.isotropic_cross_spectrum(da, db, dim=["y", "x"], detrend="linear", window="hann"))
    os.environ.pop("XRFTHIP_FASTG", None)
    api._plan_cache.clear()
    print(line + f" | fft {x.numel()/w2/1e9:6.1f} | isotropic PS {x.numel()/w3/1e9:6.1f} (two-pass {x.numel()/w4/1e9:6.1f}) | cross spectrum{'*' if on5 else ''} {x.numel()/w5/1e9:6.1f} (two-pass {x.numel()/w6/1e9:6.1f}) | isotropic cross {x.numel()/w7/1e9:6.1f} (two-pass {x.numel()/w8/1e9:6.1f}) GFFT/s", flush=True)
