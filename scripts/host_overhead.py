import sys, time, warnings
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import xrft_amd as xrft
warnings.simplefilter("ignore")
for shp, dt in (((64,1440,720), torch.float64), ((4,256,256), torch.float64), ((64,4096,4096), torch.float32)):
    x = torch.randn(shp, dtype=dt, device="cuda")
    da = xrft.DataArray(x, ("t","y","x"), {"t": np.arange(shp[0]), "y": np.arange(shp[1])*1.0, "x": np.arange(shp[2])*1.0})
    f = lambda: xrft.power_spectrum(da, dim=["y","x"], detrend="linear", window="hann")
    for _ in range(3): f()
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n): r = f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(shp, "host per call %.1f us, total per call %.1f us" % ((t1-t0)/n*1e6, (t2-t0)/n*1e6))
    import cProfile, pstats
    if shp[1] == 1440:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(50): f()
        pr.disable(); torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
