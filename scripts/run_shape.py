#!/usr/bin/env python3
"""One workload for counter collection: power_spectrum (linear detrend + Hann) of a (NT, NY, NX) DT cube, 3 calls; the environment picks the kernels
(XRFTHIP_FASTN_TABLES=0: run-time radices on table shapes)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
warnings.simplefilter("ignore")
nt, ny, nx = (int(os.environ.get(k, d)) for k, d in (("NT", "32"), ("NY", "1000"), ("NX", "1000")))
dt = torch.float64 if os.environ.get("DT", "f32") == "f64" else torch.float32
x = torch.randn((nt, ny, nx), dtype=dt, device="cuda")
da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(float(ny)), "x": np.arange(float(nx))})
for _ in range(3):
    xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")
torch.cuda.synchronize()
