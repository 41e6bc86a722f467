"""Round 5: the inner layout ((y, x, t) arrays, dim = ["y", "x"]) as two fused passes (csrc/fastn.h: fastn_cols_kernel on the [ny][nx t] view, fastn_irows_kernel)
against the composite of one-axis plans (XRFTHIP_FUSED_INNER=0), per kernel."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
shapes = (((1024, 1024, 64), torch.float32), ((720, 1440, 32), torch.float32), ((1024, 1024, 32), torch.float64), ((1000, 1000, 24), torch.float32), ((512, 512, 365), torch.float32), ((360, 720, 100), torch.float64))
for fused in ("1", "0"):
    os.environ["XRFTHIP_FUSED_INNER"] = fused
    api._plan_cache.clear()
    print(f"--- XRFTHIP_FUSED_INNER={fused}")
    for shape, dt in shapes:
        x = torch.randn(shape, dtype=dt, device="cuda")
        da = xrft.DataArray(x, ("y", "x", "t"), {"y": np.arange(float(shape[0])), "x": np.arange(float(shape[1]))})
        for name, f in (("PS linear+hann", lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")), ("PS plain", lambda: xrft.power_spectrum(da, dim=["y", "x"])),
                        ("fft", lambda: xrft.fft(da, dim=["y", "x"]))):
            f(); f(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): f()
            torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 5
            pl = next(reversed(api._plan_cache.values()))
            print(f"{shape} {'f32' if dt == torch.float32 else 'f64'} {name}: {x.numel()/w/1e9:6.1f} GFFT/s {w*1e3:.3f} ms", flush=True)
            if name != "PS plain":
                pl.set_profiling(True); f(); torch.cuda.synchronize()
                print("     ", {k: round(v[1], 3) for k, v in pl.read_profile().items()})
                pl.set_profiling(False)
            if name == "fft" and fused == "1": print("   ", pl.describe().strip().replace("\n", "\n    ")[:900])
        del x, da
