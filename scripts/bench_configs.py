#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations (parity-test cases, not the bench line) at reduced batch sizes,
to document where the generic kernels stand.  Run on the GPU box: python scripts/bench_configs.py"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
warnings.simplefilter("ignore")
dev = "cuda"

def timeit(fn, reps=3):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

def cube(shape, dtype):
    return torch.randn(shape, dtype=dtype, device=dev)

rows = []
# C1: PS (4,256,256) f64
x = cube((4, 256, 256), torch.float64); c = {"t": np.arange(4), "y": np.arange(256.), "x": np.arange(256.)}
da = xrft.DataArray(x, ("t", "y", "x"), c)
t = timeit(lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")); rows.append(("C1 PS (4,256,256) f64", x.numel() / t / 1e9, t))
# C2: dft 1-D (1024, 65536) f32
x = cube((1024, 65536), torch.float32); da = xrft.DataArray(x, ("t", "x"), {"x": np.arange(65536) * 0.5})
t = timeit(lambda: xrft.dft(da, dim="x")); rows.append(("C2 dft 1-D (1024,65536) f32", x.numel() / t / 1e9, t))
# C3 variants on the generic path
x = cube((8, 4096, 4096), torch.float32); c = {"y": np.arange(4096.), "x": np.arange(4096.)}
da = xrft.DataArray(x, ("t", "y", "x"), c)
t = timeit(lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")); rows.append(("C3 PS (8,4096,4096) f32 [fastp2]", x.numel() / t / 1e9, t))
t = timeit(lambda: xrft.fft(da, dim=["y", "x"], detrend="linear", window="hann")); rows.append(("   fft complex out (8,4096,4096) f32 [fastp2]", x.numel() / t / 1e9, t))
t = timeit(lambda: xrft.isotropic_power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")); rows.append(("   isotropic PS (8,4096,4096) f32 [fastp2]", x.numel() / t / 1e9, t))
# C4: cross + isotropic on two (16,2048,2048) f32
a = cube((16, 2048, 2048), torch.float32); b = cube((16, 2048, 2048), torch.float32); c = {"y": np.arange(2048.), "x": np.arange(2048.)}
d1 = xrft.DataArray(a, ("t", "y", "x"), c); d2 = xrft.DataArray(b, ("t", "y", "x"), c)
t = timeit(lambda: xrft.cross_spectrum(d1, d2, dim=["y", "x"], window="hann")); rows.append(("C4 cross_spectrum 2x(16,2048,2048) f32", a.numel() / t / 1e9, t))
t = timeit(lambda: xrft.isotropic_cross_spectrum(d1, d2, dim=["y", "x"], window="hann")); rows.append(("C4 isotropic_cross_spectrum", a.numel() / t / 1e9, t))
t = timeit(lambda: xrft.isotropic_power_spectrum(d1, dim=["y", "x"], window="hann")); rows.append(("C4 isotropic_power_spectrum", a.numel() / t / 1e9, t))
t = timeit(lambda: xrft.power_spectrum(d1, dim=["y", "x"], detrend="linear", window="hann")); rows.append(("   PS (16,2048,2048) f32 [fastp2]", a.numel() / t / 1e9, t))
x = cube((64, 1024, 1024), torch.float32); c = {"y": np.arange(1024.), "x": np.arange(1024.)}
da = xrft.DataArray(x, ("t", "y", "x"), c)
t = timeit(lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")); rows.append(("   PS (64,1024,1024) f32 [fastp2]", x.numel() / t / 1e9, t))
# C5: PS (64,1440,720) f64
x = cube((64, 1440, 720), torch.float64); da = xrft.DataArray(x, ("t", "lat", "lon"), {"lat": np.arange(1440) * .25, "lon": np.arange(720) * .25})
t = timeit(lambda: xrft.power_spectrum(da, dim=["lat", "lon"], detrend="constant", window="hann")); rows.append(("C5 PS (64,1440,720) f64", x.numel() / t / 1e9, t))
for name, g, t in rows:
    print(f"{name:50s} {g:8.2f} GFFT/s   {t*1e3:8.2f} ms")
