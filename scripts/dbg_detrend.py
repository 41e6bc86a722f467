import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xa
for (nt, ny, nx) in ((8, 4096, 4096), (8, 1024, 1024), (16, 512, 4096), (3, 4096, 4096)):
    x = torch.randn((nt, ny, nx), dtype=torch.float32, device="cuda")
    x += (0.01 * torch.arange(ny, device="cuda"))[None, :, None] + (-0.02 * torch.arange(nx, device="cuda"))[None, None, :]
    da = xa.DataArray(x, ("time", "y", "x"))
    det = xa.detrend(da, ["y", "x"], "linear")
    print((nt, ny, nx), "rms per slab", [round(float(v), 3) for v in (det.data.double() ** 2).mean(dim=(1, 2)).sqrt()])
