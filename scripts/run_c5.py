#!/usr/bin/env python3
"""One workload for counter collection: C5-shaped PS (C5_NT = 16, 1440, 720) f64, C5_DET = constant detrend + hann; 3 calls.
XRFTHIP_FASTM=0 in the environment selects the generic tile kernels instead of csrc/fastm.h."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
warnings.simplefilter("ignore")
dt = torch.float32 if os.environ.get("C5_F32") else torch.float64
x = torch.randn((int(os.environ.get("C5_NT", "16")), 1440, 720), dtype=dt, device="cuda")
da = xrft.DataArray(x, ("t", "lat", "lon"), {"lat": np.arange(1440) * .25, "lon": np.arange(720) * .25})
for _ in range(3):
    xrft.power_spectrum(da, dim=["lat", "lon"], detrend=os.environ.get("C5_DET", "constant"), window="hann")
torch.cuda.synchronize()
