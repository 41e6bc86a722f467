"""Round 5: geometry knobs of the fused inner-layout passes (XRFTHIP_FI_GC column pairs, _GE elements per row workgroup, _TC/_TR threads)."""
import os, sys, time, warnings, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
def run(shape, dt, env):
    for k in ("XRFTHIP_FI_GC", "XRFTHIP_FI_GE", "XRFTHIP_FI_TC", "XRFTHIP_FI_TR", "XRFTHIP_FI_VEC", "XRFTHIP_FI_DBG"): os.environ.pop(k, None)
    os.environ.update(env)
    api._plan_cache.clear()
    x = torch.randn(shape, dtype=dt, device="cuda")
    da = xrft.DataArray(x, ("y", "x", "t"), {"y": np.arange(float(shape[0])), "x": np.arange(float(shape[1]))})
    out = []
    for name, f in (("PS", lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")), ("fft", lambda: xrft.fft(da, dim=["y", "x"]))):
        try:
            f(); f(); torch.cuda.synchronize()
        except Exception as e:
            out.append(f"{name} failed {str(e)[:60]}"); continue
        pl = next(reversed(api._plan_cache.values()))
        pl.set_profiling(True)
        for _ in range(5): f()
        torch.cuda.synchronize()
        pr = pl.read_profile(); pl.set_profiling(False)
        out.append(f"{name} " + " ".join(f"{k.replace('fastn_', '')}={v[1]/v[0]*1e3:.0f}us" for k, v in pr.items() if "fit" not in k) + f" -> {x.numel()/sum(v[1]/v[0] for v in pr.values())/1e6:.0f}")
    print(shape, "f32" if dt == torch.float32 else "f64", env, " | ".join(out), flush=True)
KN = ("XRFTHIP_FI_GC", "XRFTHIP_FI_GE", "XRFTHIP_FI_TC", "XRFTHIP_FI_TR", "XRFTHIP_FI_VEC", "XRFTHIP_FI_DBG")
for shape, dt in (((1024, 1024, 64), torch.float32), ((720, 1440, 32), torch.float32), ((1024, 1024, 32), torch.float64)):
    run(shape, dt, {})
    run(shape, dt, {"XRFTHIP_FI_VEC": "0"})
    for ge in (2, 4, 8):
        for tr in (256, 512):
            for vec in (0, 1):
                run(shape, dt, {"XRFTHIP_FI_GE": str(ge), "XRFTHIP_FI_TR": str(tr), "XRFTHIP_FI_VEC": str(vec)})
