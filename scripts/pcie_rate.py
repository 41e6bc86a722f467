#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline workload: slabs start in (pinned) host memory and the spectrum returns to it."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
warnings.simplefilter("ignore")
nt, n = 16, 4096
h_in = torch.randn((nt, n, n), dtype=torch.float32).pin_memory()
h_out = torch.empty((nt, n, n), dtype=torch.float32).pin_memory()
c = {"t": np.arange(nt), "y": np.arange(float(n)), "x": np.arange(float(n))}
def step():
    d = h_in.cuda(non_blocking=True)
    ps = xrft.power_spectrum(xrft.DataArray(d, ("t", "y", "x"), c), dim=["y", "x"], detrend="linear", window="hann")
    h_out.copy_(ps.data, non_blocking=True)
    torch.cuda.synchronize()
step(); step()
t0 = time.perf_counter()
for _ in range(3): step()
dt = (time.perf_counter() - t0) / 3
t0 = time.perf_counter(); d = h_in.cuda(non_blocking=True); torch.cuda.synchronize(); th = time.perf_counter() - t0
t0 = time.perf_counter(); h_out.copy_(d, non_blocking=True); torch.cuda.synchronize(); td = time.perf_counter() - t0
gb = nt * n * n * 4 / 1e9
print(f"host->device {gb/th:.1f} GB/s, device->host {gb/td:.1f} GB/s, PCIe-inclusive power_spectrum {nt*n*n/dt/1e9:.2f} GFFT/s ({dt*1e3:.1f} ms for {nt} slabs)")
