#!/usr/bin/env python3
"""scripts/prof.py -- ONE parametrised per-kernel profiler for the GPU box (replaces the prof_*.py family of rounds 2-5).

    python scripts/prof.py call  <fn> <nt,ny,nx[,dtype]> [key=value ...] [--lib path.so] [--env K=V ...] [--reps N]
    python scripts/prof.py iso          old vs new radial-sum row kernel (XRFTHIP_ISOROWS = 0 | 1), and its phase clocks (= 2)
    python scripts/prof.py headline [--lib path.so]   the headline call, wall + per-kernel us per slab (A/B of two builds on one box)
    python scripts/prof.py c2           dft / power_spectrum of (1024, 65536) float32 rows, per kernel

`call` times any public function of xrft_amd on a synthetic cube: fn in {power_spectrum, fft, dft, ifft, cross_spectrum,
isotropic_power_spectrum, isotropic_cross_spectrum, cross_phase}; key=value are passed on (dim=y,x detrend=linear window=hann
real_dim=x shift=0 ...).  Every line printed: wall us per slab (or per row), GFFT/s, and the library's HIP-event time of each kernel.
"""
import argparse
import os
import subprocess
import sys
import time
import warnings

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def setup(lib=None, env=()):
    for kv in env:
        k, v = kv.split("=", 1)
        os.environ[k] = v
    from xrft_amd import _lib

    if lib:
        _lib.load(os.path.abspath(lib))
    warnings.simplefilter("ignore")


def timed(fn, units, reps=5, label=""):
    """(wall us per unit, {kernel: us per unit}) of fn(): two untimed calls, `reps` timed ones, then `reps` with the HIP events on."""
    import torch

    from xrft_amd import api

    r = fn(); r = fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    plan = next(reversed(api._plan_cache.values()))
    plan.set_profiling(True)
    r = fn(); torch.cuda.synchronize()
    plan.set_profiling(True)
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    prof = plan.read_profile()
    plan.set_profiling(False)
    del r
    return wall / units * 1e6, {k: v[1] / reps / units * 1e3 for k, v in prof.items()}, plan


def line(label, wall_us, kern, pts_per_unit):
    ks = " ".join(f"{k.replace('fasty_', '').replace('fastm_', 'm.')} {v:6.4g}" for k, v in kern.items())
    tot = sum(kern.values())  # (the kernels alone, HIP events: what a call costs when the host keeps the queue full)
    kg = f" = {pts_per_unit / tot / 1e3:6.1f} GFFT/s by the kernels" if tot > 0 else ""
    print(f"  {label:58s} wall {wall_us:7.4g} us = {pts_per_unit / wall_us / 1e3:6.1f} GFFT/s | {ks}{kg}", flush=True)


def make(shape, dtype="float32", two=False, trend=True, freq=False):
    import numpy as np
    import torch

    import xrft_amd as xrft

    dt = getattr(torch, dtype)
    x = torch.randn(shape, dtype=dt, device="cuda")
    if trend and len(shape) == 3 and not dt.is_complex:
        x += (0.01 * torch.arange(shape[1], device="cuda", dtype=dt))[None, :, None]
    dims = ("t", "y", "x")[-len(shape):]
    c = {d: np.arange(float(n)) for d, n in zip(dims, shape) if d != "t"}
    if freq:  # the input of an inverse transform: fftshifted frequency coordinates, centred on zero
        dims = tuple(d if d == "t" else "freq_" + d for d in dims)
        c = {"freq_" + d: np.fft.fftshift(np.fft.fftfreq(n, 1.0)) for d, n in zip(("t", "y", "x")[-len(shape):], shape) if d != "t"}
    d1 = xrft.DataArray(x, dims, c)
    d2 = xrft.DataArray(torch.randn(shape, dtype=dt, device="cuda"), dims, c) if two else None
    return d1, d2


def parse_kw(items):
    kw = {}
    for it in items:
        k, v = it.split("=", 1)
        if k in ("dim",):
            v = v.split(",")
            v = v if len(v) > 1 else v[0]
        elif v in ("None", "none"):
            v = None
        elif v in ("0", "1", "True", "False"):
            v = v in ("1", "True")
        kw[k] = v
    return kw


def cmd_call(a):
    setup(a.lib, a.env)
    import xrft_amd as xrft
    from xrft_amd import api

    parts = a.shape.split(",")
    dtype = parts[-1] if not parts[-1].isdigit() else "float32"
    shape = tuple(int(p) for p in parts if p.isdigit())
    two = a.fn in ("cross_spectrum", "isotropic_cross_spectrum", "cross_phase")
    inverse = a.fn in ("ifft", "idft")
    d1, d2 = make(shape, dtype, two, trend=not inverse, freq=inverse)
    kw = parse_kw(a.kw)
    if inverse and "dim" in kw:
        kw["dim"] = ["freq_" + d for d in kw["dim"]] if isinstance(kw["dim"], list) else "freq_" + kw["dim"]
    if inverse and kw.get("real_dim"):  # the stored half of the real axis: n/2 + 1 samples at rfftfreq
        import numpy as np
        import xrft_amd as xrft_
        rd = "freq_" + kw["real_dim"]
        kw["real_dim"] = rd
        co = {k: v.values for k, v in d1.coords.items()}
        nh = d1.sizes[rd]
        co[rd] = np.fft.rfftfreq(2 * (nh - 1), 1.0)
        d1 = xrft_.DataArray(d1.data, d1.dims, co)
    f = getattr(xrft, a.fn)
    fn = (lambda: f(d1, d2, **kw)) if two else (lambda: f(d1, **kw))
    units = shape[0]
    pts = 1
    for n in shape[1:]:
        pts *= n
    wall, kern, plan = timed(fn, units, a.reps)
    line(f"{a.fn} {shape} {dtype} {kw}", wall, kern, pts)
    print("   " + plan.describe().strip().replace("\n", "\n   "))


def cmd_iso(a):
    """profiles/r06_tune_iso.txt: fasty_rows_kernel<.., ISO> (XRFTHIP_ISOROWS=0) against fasty_isorows_kernel (1), then the phase clocks (2)."""
    setup(a.lib, a.env)
    import numpy as np
    import torch

    import xrft_amd as xrft
    from xrft_amd import api

    for name, n, nt, two, det in (("isotropic_power_spectrum 4096^2", 4096, 32, False, None), ("isotropic_power_spectrum 4096^2 linear", 4096, 32, False, "linear"),
                                  ("isotropic_power_spectrum 2048^2", 2048, 64, False, None), ("isotropic_cross_spectrum 2048^2", 2048, 64, True, None),
                                  ("isotropic_cross_spectrum 2048^2 linear", 2048, 64, True, "linear"), ("isotropic_power_spectrum 1024^2", 1024, 256, False, None)):
        d1, d2 = make((nt, n, n), "float32", two)
        fn = (lambda: xrft.isotropic_cross_spectrum(d1, d2, dim=["y", "x"], window="hann", detrend=det)) if two else \
             (lambda: xrft.isotropic_power_spectrum(d1, dim=["y", "x"], window="hann", detrend=det))
        print(f"== {name}, {nt} slabs; us per slab", flush=True)
        res = {}
        for mode, label in (("0", "fasty_rows_kernel<ISO> (round 5)"), ("1", "fasty_isorows_kernel (persistent, prefetch)")):
            os.environ["XRFTHIP_ISOROWS"] = mode
            api.clear_plan_cache()
            wall, kern, _plan = timed(fn, nt, a.reps)
            line(label, wall, kern, n * n)
            res[mode] = np.asarray(fn().values)
        print(f"  results bit-identical between the two kernels: {np.array_equal(res['0'], res['1'])}; repeats of the new one bit-identical: "
              f"{np.array_equal(res['1'], np.asarray(fn().values))}", flush=True)
        os.environ["XRFTHIP_ISOROWS"] = "2"  # the profiling build of the kernel prints its phase clocks on stderr after every launch
        api.clear_plan_cache()
        sys.stderr.flush()
        fn(); torch.cuda.synchronize()
        fn(); torch.cuda.synchronize()
        os.environ["XRFTHIP_ISOROWS"] = "1"
        api.clear_plan_cache()
        del d1, d2
        torch.cuda.empty_cache()


def cmd_isoq(a):
    """the new row kernel only (XRFTHIP_ISOROWS=1), then its phase clocks (=2), under whatever XRFTHIP_YTUNE the caller set: tuning sweeps"""
    setup(a.lib, a.env)
    import torch

    import xrft_amd as xrft
    from xrft_amd import api

    for name, n, nt, two in (("iso PS 4096^2", 4096, 32, False), ("iso PS 2048^2", 2048, 64, False), ("iso CS 2048^2", 2048, 64, True)):
        d1, d2 = make((nt, n, n), "float32", two)
        fn = (lambda: xrft.isotropic_cross_spectrum(d1, d2, dim=["y", "x"], window="hann")) if two else (lambda: xrft.isotropic_power_spectrum(d1, dim=["y", "x"], window="hann"))
        os.environ["XRFTHIP_ISOROWS"] = "1"
        api.clear_plan_cache()
        wall, kern, _plan = timed(fn, nt, a.reps)
        line(name, wall, kern, n * n)
        os.environ["XRFTHIP_ISOROWS"] = "2"
        api.clear_plan_cache()
        fn(); torch.cuda.synchronize()
        os.environ["XRFTHIP_ISOROWS"] = "1"
        api.clear_plan_cache()
        del d1, d2
        torch.cuda.empty_cache()


def cmd_headline(a):
    setup(a.lib, a.env)
    import numpy as np
    import torch

    import xrft_amd as xrft

    nt = a.nt
    d1, _ = make((nt, 4096, 4096))
    fn = lambda: xrft.power_spectrum(d1, dim=["y", "x"], detrend="linear", window="hann")
    for _rep in range(3):
        wall, kern, _plan = timed(fn, nt, a.reps)
        line(f"{a.lib or 'product'} PS ({nt},4096,4096) linear+hann", wall, kern, 4096 * 4096)


def cmd_c2(a):
    setup(a.lib, a.env)
    import numpy as np
    import torch

    import xrft_amd as xrft

    x = torch.randn((1024, 65536), dtype=torch.float32, device="cuda")
    da = xrft.DataArray(x, ("t", "x"), {"x": np.arange(65536) * 0.5})
    for label, fn in (("dft (1024,65536) f32", lambda: xrft.dft(da, dim="x")), ("power_spectrum", lambda: xrft.power_spectrum(da, dim="x")),
                      ("power_spectrum linear+hann", lambda: xrft.power_spectrum(da, dim="x", detrend="linear", window="hann")),
                      ("dft real_dim=x (half output)", lambda: xrft.dft(da, dim="x", real_dim="x"))):
        for _rep in range(2):
            wall, kern, _plan = timed(fn, 1024, max(a.reps, 20))
            line(label, wall, kern, 65536)


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name in ("call", "iso", "isoq", "headline", "c2"):
        sp = sub.add_parser(name)
        sp.add_argument("--lib", default=None)
        sp.add_argument("--env", action="append", default=[])
        sp.add_argument("--reps", type=int, default=5)
        if name == "call":
            sp.add_argument("fn")
            sp.add_argument("shape")
            sp.add_argument("kw", nargs="*")
        if name == "headline":
            sp.add_argument("--nt", type=int, default=64)
    a = ap.parse_args()
    {"call": cmd_call, "iso": cmd_iso, "isoq": cmd_isoq, "headline": cmd_headline, "c2": cmd_c2}[a.cmd](a)


if __name__ == "__main__":
    main()
