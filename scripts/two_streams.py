#!/usr/bin/env python3
"""Experiment: does running two halves of the batch on two HIP streams concurrently help (kernel-level overlap)?"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import _lib, engine
warnings.simplefilter("ignore")
dev = torch.device("cuda", 0)
nt, n = 64, 4096
x = torch.randn((nt, n, n), dtype=torch.float32, device=dev)
import scipy.signal as sps
w = sps.windows.hann(n, sym=False)
def mkplan(b):
    return engine.SpectralPlan(2, b, n, n, torch.float32, out_mode=_lib.OUT_POWER, detrend=_lib.DETREND_LINEAR,
                               flags=_lib.SHIFT_X | _lib.SHIFT_Y, scale=1.0 / (n * n), window_y=w, window_x=w)
out = torch.empty_like(x)
def run_single(p):
    p.execute(x, out=out)
def run_two(pa, pb, sa, sb):
    h = nt // 2
    with torch.cuda.stream(sa):
        pa.execute(x[:h], out=out[:h])
    with torch.cuda.stream(sb):
        pb.execute(x[h:], out=out[h:])
def timeit(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
p = mkplan(nt)
t1 = timeit(lambda: run_single(p))
print(f"single stream: {t1*1e3:.3f} ms  {nt*n*n/t1/1e9:.1f} GFFT/s")
for g in (8, 16, 32):
    os.environ["XRFTHIP_FAST_GROUP"] = str(g)
    pa, pb = mkplan(nt // 2), mkplan(nt // 2)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    t2 = timeit(lambda: run_two(pa, pb, sa, sb))
    print(f"two streams, group {g}: {t2*1e3:.3f} ms  {nt*n*n/t2/1e9:.1f} GFFT/s")
