"""csrc/fastn.h on the GPU: the verdict's target shapes (large real slabs off the tables) -- rate, per-kernel times, parity of one slab against the oracle,
and the geometry knobs of the run-time-radix kernels (XRFTHIP_FASTN_GC: sequences per column workgroup, XRFTHIP_FASTN_RPU: rows per row workgroup,
XRFTHIP_FASTN_TABLES=0: run-time radices even where fastm.h's table has the length, XRFTHIP_FASTN=0: the generic passes as before).
python scripts/prof_fastn.py [quick] on the GPU box"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
from oracle import xrft_oracle as oracle
warnings.simplefilter("ignore")
quick = "quick" in sys.argv

def rate(da, nt, fn, reps=5, **kw):
    f = lambda: fn(da, dim=["y", "x"], **kw)
    f(); f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

def one(nt, ny, nx, dt, env=None, profile=False, parity=False, fn=xrft.power_spectrum, **kw):
    env = env or {}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        g = torch.Generator(device="cuda"); g.manual_seed(5)
        x = torch.randn((nt, ny, nx), dtype=getattr(torch, dt), device="cuda", generator=g)
        x += (0.01 * torch.arange(ny, device="cuda", dtype=x.dtype))[None, :, None] + 3.0
        c = {"y": np.arange(float(ny)), "x": np.arange(float(nx))}
        da = xrft.DataArray(x, ("t", "y", "x"), c)
        api._plan_cache.clear()
        kw = kw or dict(detrend="linear", window="hann")
        w = rate(da, nt, fn, **kw)
        tag = " | ".join(l.strip()[:110] for pl in api._plan_cache.values() for l in pl.describe().split("\n")[1:2])
        line = f"({nt},{ny},{nx}) {dt} {env}: {x.numel()/w/1e9:6.1f} GFFT/s {w*1e6/nt:8.1f} us/slab  {tag}"
        if profile:
            for pl in api._plan_cache.values(): pl.set_profiling(True)
            fn(da, dim=["y", "x"], **kw); torch.cuda.synchronize()
            for pl in api._plan_cache.values():
                line += "\n      " + "  ".join(f"{k} {ms*1e3/nt:.2f}us" for k, (n, ms) in pl.read_profile().items())
                pl.set_profiling(False)
        if parity:
            got = fn(da, dim=["y", "x"], **kw).values[:1]
            ref = getattr(oracle, fn.__name__)(oracle.OArr(x[:1].cpu().numpy().astype(np.float64), ("t", "y", "x"), {"t": np.arange(1), **c}), dim=["y", "x"], **kw).values
            err = np.abs(np.asarray(got) - ref).max() / np.abs(ref).max()
            big = np.abs(ref) > 1e-6 * np.abs(ref).max()
            binrel = (np.abs(np.asarray(got) - ref)[big] / np.abs(ref)[big]).max()
            line += f"\n      parity vs oracle (float64-fed): max-norm {err:.2e}, worst bin above 1e-6 of the peak {binrel:.2e}"
        print(line, flush=True)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v

def main():
    targets = [(64, 721, 1440, "float32"), (16, 3000, 3000, "float64"), (16, 2200, 2200, "float32"), (16, 1215, 1215, "float32"), (32, 750, 1500, "float64")]
    print("== targets, default geometry, per-kernel times, parity")
    for t in targets:
        one(*t, profile=True, parity=True)
    print("== the generic passes on the same shapes (XRFTHIP_FASTN=0)")
    for t in targets:
        one(*t, env={"XRFTHIP_FASTN": 0})
    print("== geometry knobs")
    for t in targets:
        for gc in (1, 2, 4, 8):
            one(*t, env={"XRFTHIP_FASTN_GC": gc})
        for rpu in (1, 2, 4):
            one(*t, env={"XRFTHIP_FASTN_RPU": rpu})
    if not quick:
        print("== run-time radices against the table kernels on table shapes")
        for t in [(64, 1440, 720, "float64"), (64, 1440, 720, "float32"), (32, 1000, 1000, "float32"), (64, 2000, 2000, "float32"), (16, 3000, 3000, "float32"), (32, 2000, 2000, "float64"),
                  (64, 360, 720, "float64"), (16, 2160, 4320, "float32"), (64, 1024, 1024, "float64")]:
            one(*t)
            one(*t, env={"XRFTHIP_FASTN_TABLES": 0}, profile=True)
        print("== other shapes off the tables")
        for t in [(16, 2500, 1250, "float32"), (16, 4800, 4800, "float32"), (16, 2187, 2187, "float32"), (16, 2401, 2401, "float32"), (32, 1001, 1001, "float32"), (32, 1331, 1331, "float64"),
                  (16, 1536, 3072, "float64"), (64, 700, 1400, "float32"), (64, 343, 686, "float64"), (8, 4096, 4096, "float64"), (32, 1999, 1000, "float32"), (32, 1013, 1024, "float64")]:
            one(*t, profile=True, parity=True)
        print("== modes on (16, 2200, 2200) float32 and (32, 750, 1500) float64")
        for t in [(16, 2200, 2200, "float32"), (32, 750, 1500, "float64")]:
            one(*t, fn=xrft.fft, parity=True, detrend="linear", window="hann")
            one(*t, fn=xrft.isotropic_power_spectrum, parity=True, detrend="linear", window="hann")
            one(*t, fn=xrft.power_spectrum, parity=True, real_dim="x", detrend="constant")


if __name__ == "__main__":
    main()
