#!/usr/bin/env python3
"""One warm-up + a few power spectra of the headline cube through the TUNING build (build_dbg/libxrft_hip_ytune.so), XRFTHIP_YTUNE from the
environment: the command rocprofv3 --pmc wraps for the round-4 floor experiments (scripts/gpu_floor_r04.sh)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xrft_amd import _lib
if os.environ.get("USE_TUNE_LIB", "1") == "1":
    _lib.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_dbg", "libxrft_hip_ytune.so"))
import xrft_amd as xrft
warnings.simplefilter("ignore")
nt = int(os.environ.get("NT", "64"))
x = torch.randn((nt, 4096, 4096), dtype=torch.float32, device="cuda")
x += (0.01 * torch.arange(4096, device="cuda"))[None, :, None]
da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(4096.), "x": np.arange(4096.)})
for _ in range(int(os.environ.get("REPS", "3"))):
    r = xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")
torch.cuda.synchronize()
