#!/usr/bin/env python3
"""Round 4, headline floor experiment (a): the column blocks that share the input's 128-byte lines meet on a counter before they load
(XRFTHIP_YTUNE bit 21, csrc/fasty.h), against the default, through the tuning build (scripts/build_tune_yf.sh); per-kernel HIP-event
times per 4096^2 slab, three rounds each (box noise)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xrft_amd import _lib
_lib.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_dbg", "libxrft_hip_ytune.so"))
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
nt = int(os.environ.get("NT", "64"))
x = torch.randn((nt, 4096, 4096), dtype=torch.float32, device="cuda")
x += (0.01 * torch.arange(4096, device="cuda"))[None, :, None]
da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(4096.), "x": np.arange(4096.)})
ref = None


def run(tune, reps=8):
    global ref
    os.environ["XRFTHIP_YTUNE"] = str(tune)
    api.clear_plan_cache()
    f = lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")
    r = f(); r = f(); torch.cuda.synchronize()
    if ref is None:
        ref = r.data[:2].clone()
    same = bool(torch.equal(ref, r.data[:2]))
    plan = next(reversed(api._plan_cache.values()))
    plan.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    prof = plan.read_profile()
    plan.set_profiling(False)
    del r
    k = {n: v[1] / reps / nt * 1e3 for n, v in prof.items()}
    return wall, k, same


for rnd in range(3):
    for label, tune in (("default", 0), ("sharers of a line meet before they load", 1 << 21), ("... + W2 stores write-through", (1 << 21) + 2)):
        wall, k, same = run(tune)
        print(f"{label:46s} tune={tune:8d} wall {wall / nt * 1e6:6.2f} us/slab = {nt * 4096 * 4096 / wall / 1e9:6.1f} GFFT/s | "
              + " ".join(f"{n.replace('fasty_', '')} {v:5.2f}" for n, v in k.items()) + f" | bits equal: {same}", flush=True)
