"""The fused inner-layout passes on SMALL grids ((lat, lon, time) arrays): threads / elements per row workgroup (XRFTHIP_FI_TR, _GE, _TC, _GC)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
KN = ("XRFTHIP_FI_GC", "XRFTHIP_FI_GE", "XRFTHIP_FI_TC", "XRFTHIP_FI_TR")
def run(env):
    for k in KN: os.environ.pop(k, None)
    os.environ.update(env); api._plan_cache.clear()
    print("---", env)
    for shape, dt in (((73, 144, 1460), torch.float32), ((180, 360, 365), torch.float32), ((96, 192, 1024), torch.float32), ((72, 144, 1440), torch.float64), ((256, 256, 512), torch.float32)):
        x = torch.randn(shape, dtype=dt, device="cuda")
        da = xrft.DataArray(x, ("lat", "lon", "time"), {"lat": np.arange(float(shape[0])), "lon": np.arange(float(shape[1]))})
        f = lambda: xrft.power_spectrum(da, dim=["lat", "lon"], detrend="linear", window="hann")
        f(); f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 10
        pl = next(reversed(api._plan_cache.values()))
        pl.set_profiling(True); f(); torch.cuda.synchronize(); pr = pl.read_profile(); pl.set_profiling(False)
        d = pl.describe().splitlines()[1]
        import re
        m = re.search(r"cols: .*?(\d+) thr, (\d+) packed.*rows: (\d+) thr, (\d+) indep", d)
        print(f"{shape} {str(dt)[-7:]}: {x.numel()/w/1e9:6.1f} GFFT/s {w*1e3:.3f} ms | " + " ".join(f"{k.replace('fastn_','')}={v[1]/v[0]*1e3:.0f}us" for k, v in pr.items()) + (f" | cols {m.group(1)} thr G={m.group(2)}, rows {m.group(3)} thr GE={m.group(4)}" if m else " | " + d[:80]), flush=True)
        del x, da
run({})
for tr in (64, 128, 256):
    run({"XRFTHIP_FI_TR": str(tr)})
for ge, tr in ((16, 256), (4, 64), (4, 128)):
    run({"XRFTHIP_FI_GE": str(ge), "XRFTHIP_FI_TR": str(tr)})
