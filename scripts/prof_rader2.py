"""Round 5: the two-pass pipeline's columns in the prime-factor / Rader form (fastn_cols_kernel<T, 2, 16>) against the chirp convolution (XRFTHIP_FASTN_RADER=0):
the ERA5 grid (721 = 7 x 103 latitudes), 365 x 720, 1460 x 600."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
shapes = (((64, 721, 1440), torch.float32), ((64, 720, 1440), torch.float32), ((64, 721, 1440), torch.float64), ((64, 365, 720), torch.float32), ((64, 365, 720), torch.float64), ((32, 1460, 1440), torch.float32),
          ((128, 361, 720), torch.float32))
def run(env):
    for k in ("XRFTHIP_FASTN_RADER", "XRFTHIP_FASTN_GC", "XRFTHIP_FASTN_TC", "XRFTHIP_FASTN_DBG"): os.environ.pop(k, None)
    os.environ.update(env)
    api._plan_cache.clear()
    print(f"--- {env}")
    for shape, dt in shapes:
        x = torch.randn(shape, dtype=dt, device="cuda")
        da = xrft.DataArray(x, ("time", "lat", "lon"), {"lat": np.arange(float(shape[1])) * 0.25, "lon": np.arange(float(shape[2])) * 0.25})
        f = lambda: xrft.power_spectrum(da, dim=["lat", "lon"], detrend="linear", window="hann")
        f(); f(); torch.cuda.synchronize()
        pl = [p for p in api._plan_cache.values()][-1]
        t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 10
        pl.set_profiling(True)
        for _ in range(5): f()
        torch.cuda.synchronize()
        pr = pl.read_profile(); pl.set_profiling(False)
        d = pl.describe()
        tag = "Rader" if "Rader" in d else "chirp" if "chirp" in d else "smooth"
        print(f"{shape} {'f32' if dt == torch.float32 else 'f64'}: {x.numel()/w/1e9:6.1f} GFFT/s {w*1e3:.3f} ms [{tag}] " + " ".join(f"{k}={v[1]/v[0]*1e3:.0f}us" for k, v in pr.items()), flush=True)
        del x, da
run({})
run({"XRFTHIP_FASTN_RADER": "0"})
if os.environ.get("ABLATE"):
    shapes = shapes[:1] + shapes[3:4]
    for dbg in (1, 2, 4, 3, 5, 6, 7):
        run({"XRFTHIP_FASTN_DBG": str(dbg)})
    for gc, tc in ((4, 128), (8, 256), (2, 128), (4, 192), (4, 384), (8, 384)):
        run({"XRFTHIP_FASTN_GC": str(gc), "XRFTHIP_FASTN_TC": str(tc)})
