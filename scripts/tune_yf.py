#!/usr/bin/env python3
"""Cache policies and start stagger of the y-first float32 kernels on the headline shape (XRFTHIP_YTUNE, csrc/fasty.h): per-kernel
HIP-event times per 4096^2 slab for every setting, through the tuning build (scripts/build_tune_yf.sh).
Run on the GPU box: python scripts/tune_yf.py > gpurun_out/tune_yf.txt"""
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


def run(tune, reps=6):
    os.environ["XRFTHIP_YTUNE"] = str(tune)
    api.clear_plan_cache()
    f = lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")
    r = f(); r = f(); torch.cuda.synchronize()
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
    return wall, {k: v[1] / reps / nt * 1e3 for k, v in prof.items()}


def show(label, tune):
    wall, k = run(tune)
    print(f"{label:58s} tune={tune:6d}  wall {wall / nt * 1e6:6.2f} us/slab = {nt * 4096 * 4096 / wall / 1e9:6.1f} GFFT/s | "
          + " ".join(f"{n.replace('fasty_', '')} {v:5.2f}" for n, v in k.items()), flush=True)


show("default: nt W2 stores, plain loads, nt result stores", 0)
if os.environ.get("ONLY_MAP"):
    for _ in range(2):
        show("sharers of a line 16 workgroups apart (two per CU in the first round)", 1 << 19)
        show("sharers of a line: pairs 32 apart", 2 << 19)
        show("default again", 0)
    sys.exit(0)
show("W2 stores plain", 1)
show("W2 stores write-through (sc1)", 2)
show("input loads nt", 4)
show("W2 loads nt", 8)
show("result stores plain", 16)
show("W2 stores sc1 + W2 loads nt", 2 + 8)
show("W2 stores sc1 + input loads nt", 2 + 4)
for n in (1, 2, 3, 4, 5, 6, 8):
    show(f"stagger {n} x 3.4 us", n << 8)
show("stagger 4 + W2 stores sc1", (4 << 8) + 2)
show("sharers of a line 16 workgroups apart (two per CU in the first round)", 1 << 19)
show("sharers of a line: pairs 32 apart", 2 << 19)
show("default again", 0)
