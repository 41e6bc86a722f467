"""Short smooth columns in the two-pass pipeline (slabs too long for the one-pass kernel): threads per column workgroup."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
def run(env):
    for k in ("XRFTHIP_FASTN_GC", "XRFTHIP_FASTN_TC", "XRFTHIP_FASTN_RPU", "XRFTHIP_FASTN_TR"): os.environ.pop(k, None)
    os.environ.update(env); api._plan_cache.clear()
    print("---", env)
    for shape, dt in (((512, 100, 2000), torch.float32), ((512, 150, 1500), torch.float32), ((256, 250, 3000), torch.float32), ((1024, 98, 1000), torch.float32), ((256, 330, 2200), torch.float32),
                      ((256, 100, 2000), torch.float64), ((256, 150, 1500), torch.float64), ((128, 250, 3000), torch.float64), ((128, 500, 1500), torch.float64)):
        x = torch.randn(shape, dtype=dt, device="cuda")
        da = xrft.DataArray(x, ("t", "lat", "lon"), {"lat": np.arange(float(shape[1])), "lon": np.arange(float(shape[2]))})
        f = lambda: xrft.power_spectrum(da, dim=["lat", "lon"], detrend="linear", window="hann")
        f(); f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 10
        pl = next(reversed(api._plan_cache.values()))
        pl.set_profiling(True); f(); torch.cuda.synchronize(); pr = pl.read_profile(); pl.set_profiling(False)
        d = pl.describe().splitlines()[1]
        print(f"{shape} {str(dt)[-7:]}: {x.numel()/w/1e9:6.1f} GFFT/s {w*1e3:.3f} ms | " + " ".join(f"{k}={v[1]/v[0]*1e3:.0f}us" for k, v in pr.items()) + " | " + d[:120], flush=True)
        del x, da
run({})
if os.environ.get("SWEEP"):
    for tc in (64, 128, 192, 256):
        run({"XRFTHIP_FASTN_TC": str(tc)})
