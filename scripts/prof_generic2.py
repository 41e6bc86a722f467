#!/usr/bin/env python3
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
def prof(name, fn, nslab):
    fn(); fn(); torch.cuda.synchronize()
    plan = next(reversed(api._plan_cache.values()))
    plan.set_profiling(True)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    p = plan.read_profile(); plan.set_profiling(False)
    print(name, " | ".join(f"{k} {ms/3*1e3/nslab:.1f}" for k, (c, ms) in p.items()), "||", plan.describe().split("\n")[1].strip()[:90])
a = torch.randn((16, 2048, 2048), dtype=torch.float32, device="cuda"); c = {"y": np.arange(2048.), "x": np.arange(2048.)}
d1 = xrft.DataArray(a, ("t", "y", "x"), c)
prof("PS plain          ", lambda: xrft.power_spectrum(d1, dim=["y", "x"]), 16)
prof("PS hann           ", lambda: xrft.power_spectrum(d1, dim=["y", "x"], window="hann"), 16)
prof("PS linear         ", lambda: xrft.power_spectrum(d1, dim=["y", "x"], detrend="linear"), 16)
prof("PS noshift        ", lambda: xrft.power_spectrum(d1, dim=["y", "x"], shift=False), 16)
prof("PS real_dim       ", lambda: xrft.power_spectrum(d1, dim=["y"], real_dim="x"), 16)
prof("fft complex out   ", lambda: xrft.fft(d1, dim=["y", "x"], true_phase=False), 16)
ac = torch.randn((16, 2048, 2048), dtype=torch.complex64, device="cuda")
d2 = xrft.DataArray(ac, ("t", "y", "x"), c)
prof("fft complex in    ", lambda: xrft.fft(d2, dim=["y", "x"], true_phase=False), 16)
for env, val in (("XRFTHIP_LDS_SOFT", "32768"), ("XRFTHIP_LDS_SOFT", "150000")):
    os.environ[env] = val; api._plan_cache.clear()
    prof(f"PS plain {env}={val}", lambda: xrft.power_spectrum(d1, dim=["y", "x"]), 16)
