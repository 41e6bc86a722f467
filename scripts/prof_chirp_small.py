"""Small / medium grids whose column length needs the chirp convolution (181 x 360, 241 x 480, 94 x 192, 361 x 720): geometry of the Bluestein column kernel."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
def run(env):
    for k in ("XRFTHIP_FASTN_GC", "XRFTHIP_FASTN_TC"): os.environ.pop(k, None)
    os.environ.update(env); api._plan_cache.clear()
    print("---", env)
    for shape in ((1024, 181, 360), (1024, 241, 480), (2048, 94, 192), (128, 361, 720), (64, 1013, 1024)):
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
        print(f"{shape} f32: {x.numel()/w/1e9:6.1f} GFFT/s {w*1e3:.3f} ms | " + " ".join(f"{k}={v[1]/v[0]*1e3:.0f}us" for k, v in pr.items()) + " | " + d[:110], flush=True)
        del x, da
run({})
if os.environ.get("SWEEP"):
    for gc, tc in ((2, 64), (2, 128), (4, 64), (4, 128), (4, 256), (8, 128), (8, 256), (1, 64)):
        run({"XRFTHIP_FASTN_GC": str(gc), "XRFTHIP_FASTN_TC": str(tc)})
