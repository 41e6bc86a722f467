"""Small slabs whose column length holds an awkward prime (73 x 144: the 2.5-degree NCEP grid; 37 x 72; 145 x 192): the two-pass pipeline with the Rader columns, per kernel."""
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
    for shape in ((4096, 73, 144), (4096, 72, 144), (16384, 37, 72), (2048, 145, 192), (1460, 73, 144)):
        x = torch.randn(shape, dtype=torch.float32, device="cuda")
        da = xrft.DataArray(x, ("t", "lat", "lon"), {"lat": np.arange(float(shape[1])), "lon": np.arange(float(shape[2]))})
        f = lambda: xrft.power_spectrum(da, dim=["lat", "lon"], detrend="linear", window="hann")
        f(); f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 10
        pl = next(reversed(api._plan_cache.values()))
        pl.set_profiling(True); f(); torch.cuda.synchronize(); pr = pl.read_profile(); pl.set_profiling(False)
        d = pl.describe().splitlines()[1]
        print(f"{shape} f32: {x.numel()/w/1e9:6.1f} GFFT/s {w*1e3:.3f} ms | " + " ".join(f"{k}={v[1]/v[0]*1e3:.0f}us" for k, v in pr.items()) + " | " + d[:230], flush=True)
        del x, da
run({})
