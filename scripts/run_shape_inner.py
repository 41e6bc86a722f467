#!/usr/bin/env python3
"""One workload for counter collection: power_spectrum (linear detrend + Hann) over (y, x) of a (NY, NX, NT) float32 array -- the inner layout, 3 calls
(XRFTHIP_FUSED_INNER=0: the composite of one-axis plans)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
warnings.simplefilter("ignore")
nt, ny, nx = (int(os.environ.get(k, d)) for k, d in (("NT", "64"), ("NY", "1024"), ("NX", "1024")))
x = torch.randn((ny, nx, nt), dtype=torch.float32, device="cuda")
da = xrft.DataArray(x, ("y", "x", "t"), {"y": np.arange(float(ny)), "x": np.arange(float(nx))})
for _ in range(3):
    xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")
torch.cuda.synchronize()
