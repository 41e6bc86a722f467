"""Two transform axes that are NOT adjacent: dim = ["time", "lon"] of a (time, lat, lon) array (wavenumber-frequency spectra), xrfthip_desc.mid."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
for fused in ("1", "0"):
    os.environ["XRFTHIP_FUSED_INNER"] = fused
    api._plan_cache.clear()
    print(f"--- XRFTHIP_FUSED_INNER={fused}")
    for shape, dt in (((1440, 73, 144), torch.float32), ((1024, 64, 512), torch.float32), ((360, 180, 360), torch.float32), ((720, 91, 360), torch.float64), ((2048, 32, 1024), torch.float32), ((1460, 73, 144), torch.float32)):
        x = torch.randn(shape, dtype=dt, device="cuda")
        da = xrft.DataArray(x, ("time", "lat", "lon"), {"time": np.arange(float(shape[0])), "lon": np.arange(float(shape[2])) * 2.5})
        for name, f in (("PS linear+hann", lambda: xrft.power_spectrum(da, dim=["time", "lon"], detrend="linear", window="hann")), ("fft", lambda: xrft.fft(da, dim=["time", "lon"]))):
            f(); f(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): f()
            torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 10
            pl = next(reversed(api._plan_cache.values()))
            pl.set_profiling(True); f(); torch.cuda.synchronize(); pr = pl.read_profile(); pl.set_profiling(False)
            print(f"{shape} {'f32' if dt == torch.float32 else 'f64'} {name}: {x.numel()/w/1e9:6.1f} GFFT/s {w*1e3:.3f} ms | " + " ".join(f"{k}={v[1]/v[0]*1e3:.0f}us" for k, v in pr.items()), flush=True)
        del x, da
