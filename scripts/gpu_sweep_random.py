#!/usr/bin/env python3
"""One-off GPU sweep: random larger shapes (four-step, Bluestein, composite radices, float64/32, real/complex) against the oracle."""
import os, sys, warnings, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
warnings.simplefilter("ignore")
import numpy as np, torch
import xrft_amd as xa
from xrft_amd import api
from oracle import xrft_oracle as o
import cases
N1 = [1000, 1024, 1440, 2160, 3000, 4099, 6000, 8192, 10000, 12289, 16384, 30030, 65536, 100000, 131072, 250000]
N2 = [96, 100, 128, 131, 180, 243, 250, 256, 360, 384, 500, 512, 720, 1000, 1024, 1440, 1801, 2048]
bad = []
t0 = time.time()
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 120):
    rng = np.random.default_rng(9000 + seed)
    dtype = str(rng.choice(["float64", "float32", "complex128", "complex64"], p=[0.35, 0.35, 0.15, 0.15]))
    two = rng.random() < 0.5
    if two:
        ny, nx = int(rng.choice(N2)), int(rng.choice(N2)); shape = (2, ny, nx); dims = ("t", "y", "x"); td = ["y", "x"]
    else:
        nx = int(rng.choice(N1)); shape = (3, nx); dims = ("t", "x"); td = ["x"]
    v = rng.standard_normal(shape)
    if dtype.startswith("complex"): v = v + 1j * rng.standard_normal(shape)
    v = (v + 0.001 * np.arange(shape[-1])).astype(dtype)
    c = {"t": np.arange(shape[0]), "x": np.arange(shape[-1]) * 0.5}
    if two: c["y"] = np.arange(shape[1]) * 2.0
    kw = dict(detrend=rng.choice([None, "constant", "linear"]), window=rng.choice([None, "hann"]))
    kind = str(rng.choice(["fft", "ps", "roundtrip"]))
    real_dim = td[-1] if (not dtype.startswith("complex") and rng.random() < 0.3 and kind != "roundtrip") else None
    da = xa.DataArray(torch.from_numpy(v).cuda(), dims, c); od = o.OArr(v, dims, c)
    tol = cases.TOL[dtype]
    try:
        if kind == "fft":
            got, ref = xa.fft(da, dim=td, real_dim=real_dim, **kw), o.fft(od, dim=td, real_dim=real_dim, **kw)
        elif kind == "ps":
            got, ref = xa.power_spectrum(da, dim=td, real_dim=real_dim, **kw), o.power_spectrum(od, dim=td, real_dim=real_dim, **kw)
        else:
            got = xa.ifft(xa.fft(da, dim=td), dim=["freq_" + d for d in td]); ref = o.ifft(o.fft(od, dim=td), dim=["freq_" + d for d in td])
        cases.check(got, ref, tol * (3 if shape[-1] > 50000 else 1))
    except Exception as e:
        bad.append((seed, dtype, shape, kind, real_dim, dict(kw), repr(e)[:160]))
    api._plan_cache.clear()
print(f"{seed + 1} cases in {time.time() - t0:.0f} s, failures: {len(bad)}")
for b in bad[:10]: print(b)
