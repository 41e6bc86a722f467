"""Seeded random differential test: product API (kernels on the CPU emulator) against the oracle over random shapes, dtypes
and option combinations -- every prologue / epilogue switch of the generic tile kernel in combinations the hand-written
cases do not enumerate."""
import os
import sys
import warnings

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emu"))
import build_emu  # noqa: E402

from xrft_amd import _lib, api  # noqa: E402
from oracle import xrft_oracle as o  # noqa: E402

import cases  # noqa: E402

warnings.simplefilter("ignore")

SIZES = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 15, 16, 18, 20, 21, 24, 25, 27, 30, 32, 33, 36, 40, 45, 48, 49, 60, 64, 77, 90, 96, 131]
WINDOWS = [None, "hann", "hamming", "blackman", "boxcar", "tukey"]


@pytest.fixture(scope="module", autouse=True)
def emulated_library():
    api._plan_cache.clear()
    _lib._load_for_testing(build_emu.build())
    yield
    api._plan_cache.clear()
    _lib._state.update(dll=None, path=None, device="cuda")


def _draw(rng):
    two_d = rng.random() < 0.6
    dtype = rng.choice(["float64", "float32", "complex128", "complex64"], p=[0.35, 0.35, 0.15, 0.15])
    nb = int(rng.integers(1, 4))
    ny = int(rng.choice(SIZES[2:])) if two_d else None
    nx = int(rng.choice(SIZES[2:]))
    shape = (nb, ny, nx) if two_d else (nb, nx)
    v = rng.standard_normal(shape)
    if dtype.startswith("complex"):
        v = v + 1j * rng.standard_normal(shape)
    v = v + 0.05 * np.arange(nx)[None, ...] if not two_d else v + 0.05 * np.arange(nx)[None, None, :] - 0.03 * np.arange(ny)[None, :, None]
    v = v.astype(dtype)
    dims = ("t", "y", "x") if two_d else ("t", "x")
    coords = {"t": np.arange(nb), "x": np.arange(nx) * float(rng.choice([0.25, 1.0, 3.0])) + float(rng.choice([0.0, -4.0, 2.5]))}
    if two_d:
        coords["y"] = np.arange(ny) * float(rng.choice([0.5, 1.0, 2.0])) + float(rng.choice([0.0, 1.0]))
    if rng.random() < 0.15:  # descending coordinate (flip under true_phase)
        coords["x"] = coords["x"][::-1].copy()
    kw = {}
    tdims = ["y", "x"] if two_d else ["x"]
    if two_d and rng.random() < 0.25:
        tdims = [str(rng.choice(["y", "x"]))]
    kw["detrend"] = rng.choice([None, "constant", "linear"])
    kw["window"] = rng.choice(WINDOWS)
    kw["shift"] = bool(rng.random() < 0.7)
    real_ok = not dtype.startswith("complex")
    real_dim = None
    if real_ok and rng.random() < 0.3:
        real_dim = tdims[-1]
    return v, dims, coords, tdims, real_dim, kw, dtype


@pytest.mark.parametrize("seed", range(int(os.environ.get("XRFT_RANDOM_CASES", "70"))))
def test_random_case(seed):
    run_random(seed)


def run_random(seed):
    """One random case (also run on the GPU by tests/test_gpu_parity.py)."""
    import xrft_amd as xa

    rng = np.random.default_rng(1000 + seed)
    v, dims, coords, tdims, real_dim, kw, dtype = _draw(rng)
    tol = cases.TOL[dtype]
    da, od = cases.pair(v, dims, coords)
    kind = rng.choice(["fft", "ps", "cs", "roundtrip"])
    if kw["detrend"] == "linear" and len(tdims) == 1 and kw["window"] is None and kind == "ps":
        pass
    try:
        if kind == "fft":
            tp = bool(rng.random() < 0.5)
            ref = o.fft(od, dim=tdims, real_dim=real_dim, true_phase=tp, **kw)
            got = xa.fft(da, dim=tdims, real_dim=real_dim, true_phase=tp, **kw)
        elif kind == "ps":
            sc = str(rng.choice(["density", "spectrum"]))
            wc = bool(kw["window"] is not None and rng.random() < 0.5)
            ref = o.power_spectrum(od, dim=tdims, real_dim=real_dim, scaling=sc, window_correction=wc, **kw)
            got = xa.power_spectrum(da, dim=tdims, real_dim=real_dim, scaling=sc, window_correction=wc, **kw)
        elif kind == "cs":
            w = (np.roll(v, 1, axis=-1) * 0.5 + 0.1).astype(dtype)
            db, ob = cases.pair(w, dims, coords)
            if "x" in coords and coords["x"][0] > coords["x"][-1] and kw["window"] is not None:
                kw["window"] = None
            ref = o.cross_spectrum(od, ob, dim=tdims, real_dim=real_dim, **kw)
            got = xa.cross_spectrum(da, db, dim=tdims, real_dim=real_dim, **kw)
        else:
            F, Fo = xa.fft(da, dim=tdims), o.fft(od, dim=tdims)
            fd = ["freq_" + d for d in tdims]
            ref = o.ifft(Fo, dim=fd)
            got = xa.ifft(F, dim=fd)
    except (ValueError, NotImplementedError) as e:  # both sides must refuse alike
        with pytest.raises(type(e)):
            {"fft": lambda: xa.fft(da, dim=tdims, real_dim=real_dim, **kw),
             "ps": lambda: xa.power_spectrum(da, dim=tdims, real_dim=real_dim, **kw),
             "cs": lambda: xa.cross_spectrum(da, da, dim=tdims, real_dim=real_dim, **kw),
             "roundtrip": lambda: xa.ifft(xa.fft(da, dim=tdims), dim=["freq_" + d for d in tdims])}[kind]()
        return
    cases.check(got, ref, tol)


@pytest.mark.parametrize("seed", range(int(os.environ.get("XRFT_RANDOM_FAST_CASES", "24"))))
def test_random_fastp2_case(seed):
    run_random_fast(seed)


def run_random_fast(seed):
    """Random mode / option combinations on the shapes the specialised kernels take (256 and 512 keep the emulator quick)."""
    import xrft_amd as xa

    rng = np.random.default_rng(5000 + seed)
    ny, nx = int(rng.choice([256, 512])), int(rng.choice([256, 512]))
    nb = int(rng.integers(1, 4))
    v = rng.standard_normal((nb, ny, nx)).astype(np.float32)
    v += (0.01 * np.arange(ny, dtype=np.float32))[None, :, None] + (-0.02 * np.arange(nx, dtype=np.float32) + 3)[None, None, :]
    v *= (1 + np.arange(nb, dtype=np.float32))[:, None, None]
    c = {"t": np.arange(nb), "y": np.arange(ny) * float(rng.choice([0.5, 1.0])) + float(rng.choice([0.0, 2.0])),
         "x": np.arange(nx) * float(rng.choice([0.25, 1.0])) - float(rng.choice([0.0, 3.0]))}
    da, od = cases.pair(v, ("t", "y", "x"), c)
    w = rng.standard_normal((nb, ny, nx)).astype(np.float32)
    db, ob = cases.pair(w, ("t", "y", "x"), c)
    kw = dict(detrend=rng.choice([None, "constant", "linear"]), window=rng.choice([None, "hann", "hamming"]))
    kind = str(rng.choice(["ps", "fft", "cs", "iso", "isocs", "ps_real", "fft_real"]))
    shift = bool(rng.random() < 0.7)
    tp = bool(rng.random() < 0.5)
    api._plan_cache.clear()
    if kind == "ps":
        sc = str(rng.choice(["density", "spectrum"]))
        got, ref = xa.power_spectrum(da, dim=["y", "x"], shift=shift, scaling=sc, **kw), o.power_spectrum(od, dim=["y", "x"], shift=shift, scaling=sc, **kw)
    elif kind == "fft":
        got, ref = xa.fft(da, dim=["y", "x"], shift=shift, true_phase=tp, **kw), o.fft(od, dim=["y", "x"], shift=shift, true_phase=tp, **kw)
    elif kind == "cs":
        got, ref = xa.cross_spectrum(da, db, dim=["y", "x"], shift=shift, true_phase=tp, **kw), o.cross_spectrum(od, ob, dim=["y", "x"], shift=shift, true_phase=tp, **kw)
    elif kind == "iso":
        tr = bool(rng.random() < 0.5)
        got, ref = xa.isotropic_power_spectrum(da, dim=["y", "x"], truncate=tr, **kw), o.isotropic_power_spectrum(od, dim=["y", "x"], truncate=tr, **kw)
    elif kind == "isocs":
        tr = bool(rng.random() < 0.5)
        got, ref = xa.isotropic_cross_spectrum(da, db, dim=["y", "x"], truncate=tr, **kw), o.isotropic_cross_spectrum(od, ob, dim=["y", "x"], truncate=tr, **kw)
    elif kind == "ps_real":
        got, ref = xa.power_spectrum(da, dim=["y"], real_dim="x", **kw), o.power_spectrum(od, dim=["y"], real_dim="x", **kw)
    else:
        got, ref = xa.fft(da, dim=["y"], real_dim="x", true_phase=tp, **kw), o.fft(od, dim=["y"], real_dim="x", true_phase=tp, **kw)
    assert any("[fast" in p.describe() for p in api._plan_cache.values()), kind
    cases.check(got, ref, 3e-4)


def _smooth_lengths(lo, hi, even=False):
    out = []
    for n in range(lo, hi + 1):
        m = n
        for f in (2, 3, 5):
            while m % f == 0:
                m //= f
        if m == 1 and (not even or n % 2 == 0):
            out.append(n)
    return out


def run_random_small_slab(seed, max_points=19000):
    """Random small real slabs of any smooth shape (the half spectrum fits one workgroup's LDS), both precisions, many slabs per call: the
    one-pass kernel with run-time radices (csrc/fastg.h; 64 | 128 | 256 points per axis in float32: csrc/fasts.h) -- power / complex /
    isotropic spectra with random options against the oracle.  Returns the tag of the kernel that served the call."""
    import xrft_amd as xa

    rng = np.random.default_rng(9000 + seed)
    dtype = str(rng.choice(["float32", "float64"]))
    cap = max_points if dtype == "float32" else max_points // 2
    while True:
        ny = int(rng.choice(_smooth_lengths(2, 200)))
        nx = int(rng.choice(_smooth_lengths(4, 300, even=bool(rng.random() < 0.7))))
        cols = nx // 2 + 2 if nx % 2 == 0 else nx + 1  # (an odd nx: the whole spectrum in the tile)
        if ny * cols + 2 * (nx + ny) <= cap - 300:  # (the tile, the tables and the windows in 150 KB of LDS)
            break
    nb = int(rng.integers(1, 40))
    v = rng.standard_normal((nb, ny, nx))
    v += (0.01 * np.arange(ny))[None, :, None] + (-0.02 * np.arange(nx) + 3)[None, None, :]
    v *= (1 + (np.arange(nb) % 5))[:, None, None]
    v = v.astype(dtype)
    c = {"t": np.arange(nb), "y": np.arange(ny) * float(rng.choice([0.5, 1.0])) + float(rng.choice([0.0, 2.0])),
         "x": np.arange(nx) * float(rng.choice([0.25, 1.0])) - float(rng.choice([0.0, 3.0]))}
    da, od = cases.pair(v, ("t", "y", "x"), c)
    kw = dict(detrend=rng.choice([None, "constant", "linear"]), window=rng.choice([None, "hann", "hamming"]))
    kind = str(rng.choice(["ps", "ps", "fft", "iso", "ps_real", "fft_real", "cs", "isocs"]))
    shift = bool(rng.random() < 0.7)
    tp = bool(rng.random() < 0.5)
    api._plan_cache.clear()
    if kind in ("cs", "isocs"):
        w = (np.roll(v, 1, axis=-1) * 0.5 + 0.1 + 0.2 * rng.standard_normal(v.shape)).astype(dtype)
        c2 = dict(c, x=c["x"] + float(rng.choice([0.0, 0.0, 1.0])))
        db, ob = cases.pair(w, ("t", "y", "x"), c2)
        if kind == "cs":
            got, ref = xa.cross_spectrum(da, db, dim=["y", "x"], shift=shift, true_phase=tp, **kw), o.cross_spectrum(od, ob, dim=["y", "x"], shift=shift, true_phase=tp, **kw)
        else:
            if min(ny, nx) < 8:
                kw["nfactor"] = 1
            tr = bool(rng.random() < 0.5)
            got, ref = xa.isotropic_cross_spectrum(da, db, dim=["y", "x"], truncate=tr, **kw), o.isotropic_cross_spectrum(od, ob, dim=["y", "x"], truncate=tr, **kw)
        cases.check(got, ref, max(cases.TOL[dtype], 1e-9))
        return next((t for t in ("[fastg]", "[fasts]", "[fastm]", "[main]") if any(t in p.describe() for p in api._plan_cache.values())), None)
    if kind == "ps":
        sc = str(rng.choice(["density", "spectrum"]))
        got, ref = xa.power_spectrum(da, dim=["y", "x"], shift=shift, scaling=sc, **kw), o.power_spectrum(od, dim=["y", "x"], shift=shift, scaling=sc, **kw)
    elif kind == "fft":
        got, ref = xa.fft(da, dim=["y", "x"], shift=shift, true_phase=tp, **kw), o.fft(od, dim=["y", "x"], shift=shift, true_phase=tp, **kw)
    elif kind == "ps_real":
        got, ref = xa.power_spectrum(da, dim=["y"], real_dim="x", **kw), o.power_spectrum(od, dim=["y"], real_dim="x", **kw)
    elif kind == "fft_real":
        got, ref = xa.fft(da, dim=["y"], real_dim="x", true_phase=tp, **kw), o.fft(od, dim=["y"], real_dim="x", true_phase=tp, **kw)
    else:
        if min(ny, nx) < 8:
            kw["nfactor"] = 1
        tr = bool(rng.random() < 0.5)
        got, ref = xa.isotropic_power_spectrum(da, dim=["y", "x"], truncate=tr, **kw), o.isotropic_power_spectrum(od, dim=["y", "x"], truncate=tr, **kw)
    tags = [t for t in ("[fastg]", "[fasts]", "[fastm]") if any(t in p.describe() for p in api._plan_cache.values())]
    assert tags, (kind, ny, nx, dtype, [p.describe() for p in api._plan_cache.values()])
    cases.check(got, ref, max(cases.TOL[dtype], 1e-9))
    return tags[0]


@pytest.mark.parametrize("seed", range(int(os.environ.get("XRFT_RANDOM_SMALL_SLAB_CASES", "30"))))
def test_random_small_slab_case(seed):
    run_random_small_slab(seed)


def run_random_fastm(seed, lengths=(180, 240, 360, 480, 500, 720, 960, 1000, 1200, 1440), dtype="float64"):
    """Random mode / option combinations in float64 / float32 on the lat/lon lengths of csrc/fastm.h (BASELINE.json configs[4] is
    (64, 1440, 720)); modes the mixed-radix kernels do not take (a flipped axis) must come out
    right through the generic ones."""
    import xrft_amd as xa

    rng = np.random.default_rng(7000 + seed)
    ny, nx = int(rng.choice(lengths)), int(rng.choice(lengths))
    nb = int(rng.integers(1, 4))
    v = rng.standard_normal((nb, ny, nx)).astype(dtype)
    v += ((0.01 * np.arange(ny))[None, :, None] + (-0.02 * np.arange(nx) + 3)[None, None, :]).astype(dtype)
    v *= (1 + np.arange(nb, dtype=dtype))[:, None, None]
    desc = bool(rng.random() < 0.15)
    yc = np.arange(ny) * float(rng.choice([0.5, 1.0])) + float(rng.choice([0.0, 2.0]))
    c = {"t": np.arange(nb), "y": yc[::-1].copy() if desc else yc, "x": np.arange(nx) * float(rng.choice([0.25, 1.0])) - float(rng.choice([0.0, 3.0]))}
    da, od = cases.pair(v, ("t", "y", "x"), c)
    w = rng.standard_normal((nb, ny, nx)).astype(dtype)
    db, ob = cases.pair(w, ("t", "y", "x"), c)
    kw = dict(detrend=rng.choice([None, "constant", "linear"]), window=rng.choice([None, "hann", "hamming"]))
    kind = str(rng.choice(["ps", "ps", "fft", "cs", "iso", "isocs", "ps_real"]))
    shift = bool(rng.random() < 0.7)
    tp = bool(rng.random() < 0.5)
    tr = bool(rng.random() < 0.5)  # (truncate=False: every sample binned, a radial map: the per-bin gather; True: unbinned corners, the tables)
    api._plan_cache.clear()
    if kind == "ps":
        sc = str(rng.choice(["density", "spectrum"]))
        got, ref = xa.power_spectrum(da, dim=["y", "x"], shift=shift, scaling=sc, **kw), o.power_spectrum(od, dim=["y", "x"], shift=shift, scaling=sc, **kw)
    elif kind == "fft":
        got, ref = xa.fft(da, dim=["y", "x"], shift=shift, true_phase=tp, **kw), o.fft(od, dim=["y", "x"], shift=shift, true_phase=tp, **kw)
    elif kind == "cs":
        got, ref = xa.cross_spectrum(da, db, dim=["y", "x"], shift=shift, true_phase=tp, **kw), o.cross_spectrum(od, ob, dim=["y", "x"], shift=shift, true_phase=tp, **kw)
    elif kind == "iso":
        got, ref = xa.isotropic_power_spectrum(da, dim=["y", "x"], truncate=tr, **kw), o.isotropic_power_spectrum(od, dim=["y", "x"], truncate=tr, **kw)
    elif kind == "isocs":
        got, ref = xa.isotropic_cross_spectrum(da, db, dim=["y", "x"], truncate=tr, true_phase=tp, **kw), o.isotropic_cross_spectrum(od, ob, dim=["y", "x"], truncate=tr, true_phase=tp, **kw)
    else:
        got, ref = xa.power_spectrum(da, dim=["y"], real_dim="x", **kw), o.power_spectrum(od, dim=["y"], real_dim="x", **kw)
    on = any("[fastm]" in p.describe() for p in api._plan_cache.values())
    flipped = desc and tp and kind in ("fft", "cs")  # (the reference flips only under true_phase, xrft.py:436-441; power spectra never)
    # which (ny, nx, mode) combinations the mixed-radix kernels take depends on their per-length geometry (columns per pass-1
    # workgroup, rows per pass-2 workgroup: csrc/fastm.h); what must hold is that a flipped axis never does, and that the plain
    # power spectrum of a table length always does when the row length divides into the column blocks
    assert not (on and flipped), (kind, desc, tp, ny, nx, on)
    if kind == "ps" and nx % 8 == 0:
        assert on, (kind, desc, tp, ny, nx, on)
    cases.check(got, ref, 1e-10 if dtype == "float64" else 3e-4)


@pytest.mark.parametrize("seed", range(int(os.environ.get("XRFT_RANDOM_FASTM_CASES", "12"))))
def test_random_fastm_case(seed):
    run_random_fastm(seed, lengths=(360,), dtype="float64" if seed % 2 == 0 else "float32")


def smooth_lengths(lo, hi, primes=(2, 3, 5, 7, 11, 13)):
    """Every n in [lo, hi] whose prime factors are all in `primes` (the butterflies of csrc/fastn.h)."""
    out = []
    for n in range(lo, hi + 1):
        m = n
        for q in primes:
            while m % q == 0:
                m //= q
        if m == 1:
            out.append(n)
    return out


def run_random_fastn(seed, lo=16, hi=200, dtype="float64", blue_p=0.15):
    """Random mode / option combinations on slabs whose lengths are NOT in any table: csrc/fastn.h, the y-first two-pass pipeline with the lengths
    as data (run-time radices incl. 7 / 11 / 13, odd lengths, ragged column blocks; now and then a column length with a large prime factor:
    the chirp convolution inside the tile).  A flipped axis must come out right through the generic passes."""
    import xrft_amd as xa

    rng = np.random.default_rng(9000 + seed)
    lens = smooth_lengths(lo, hi)
    ny, nx = int(rng.choice(lens)), int(rng.choice(lens))
    blue = bool(rng.random() < blue_p)
    if blue:  # a length the butterflies do not factor (a prime factor >= 17)
        cand = [n for n in range(lo, hi + 1) if n not in set(lens)]
        ny = int(rng.choice(cand))
    nb = int(rng.integers(1, 4))
    v = rng.standard_normal((nb, ny, nx)).astype(dtype)
    v += ((0.01 * np.arange(ny))[None, :, None] + (-0.02 * np.arange(nx) + 3)[None, None, :]).astype(dtype)
    v *= (1 + np.arange(nb, dtype=dtype))[:, None, None]
    desc = bool(rng.random() < 0.1)
    yc = np.arange(ny) * float(rng.choice([0.5, 1.0])) + float(rng.choice([0.0, 2.0]))
    c = {"t": np.arange(nb), "y": yc[::-1].copy() if desc else yc, "x": np.arange(nx) * float(rng.choice([0.25, 1.0])) - float(rng.choice([0.0, 3.0]))}
    da, od = cases.pair(v, ("t", "y", "x"), c)
    w = rng.standard_normal((nb, ny, nx)).astype(dtype)
    db, ob = cases.pair(w, ("t", "y", "x"), c)
    kw = dict(detrend=rng.choice([None, "constant", "linear"]), window=rng.choice([None, "hann", "hamming"]))
    kind = str(rng.choice(["ps", "ps", "fft", "cs", "iso", "isocs", "ps_real", "phase"]))
    shift = bool(rng.random() < 0.7)
    tp = bool(rng.random() < 0.5)
    tr = bool(rng.random() < 0.3)
    api._plan_cache.clear()
    if kind == "ps":
        sc = str(rng.choice(["density", "spectrum"]))
        got, ref = xa.power_spectrum(da, dim=["y", "x"], shift=shift, scaling=sc, **kw), o.power_spectrum(od, dim=["y", "x"], shift=shift, scaling=sc, **kw)
    elif kind == "fft":
        got, ref = xa.fft(da, dim=["y", "x"], shift=shift, true_phase=tp, **kw), o.fft(od, dim=["y", "x"], shift=shift, true_phase=tp, **kw)
    elif kind == "cs":
        got, ref = xa.cross_spectrum(da, db, dim=["y", "x"], shift=shift, true_phase=tp, **kw), o.cross_spectrum(od, ob, dim=["y", "x"], shift=shift, true_phase=tp, **kw)
    elif kind == "phase":
        got, ref = xa.cross_phase(da, db, dim=["y", "x"], shift=shift, true_phase=tp, **kw), o.cross_phase(od, ob, dim=["y", "x"], shift=shift, true_phase=tp, **kw)
    elif kind == "iso":
        got, ref = xa.isotropic_power_spectrum(da, dim=["y", "x"], truncate=tr, **kw), o.isotropic_power_spectrum(od, dim=["y", "x"], truncate=tr, **kw)
    elif kind == "isocs":
        got, ref = xa.isotropic_cross_spectrum(da, db, dim=["y", "x"], truncate=tr, true_phase=tp, **kw), o.isotropic_cross_spectrum(od, ob, dim=["y", "x"], truncate=tr, true_phase=tp, **kw)
    else:
        got, ref = xa.power_spectrum(da, dim=["y"], real_dim="x", **kw), o.power_spectrum(od, dim=["y"], real_dim="x", **kw)
    tags = [p.describe() for p in api._plan_cache.values()]
    on = any("[fastn]" in t for t in tags)
    flipped = desc and tp and kind in ("fft", "cs", "phase", "isocs")
    assert not (on and flipped), (kind, desc, tp, ny, nx, on)
    if kind == "ps" and not blue and not any("[fastg" in t or "[fasts" in t or "[fastm]" in t or "[fasty" in t for t in tags):
        assert on, (kind, ny, nx, dtype, tags)  # (a plain power spectrum of a slab no other specialised kernel takes is always served here; a chirp
        #                                          convolution of 2 ny points that does not fit the LDS stays with the generic passes)
    if kind == "phase":  # (the angle of a near-zero cross spectrum amplifies rounding: compare where the product is not small)
        assert tuple(got.dims) == tuple(ref.dims)
        cs = o.cross_spectrum(od, ob, dim=["y", "x"], shift=shift, true_phase=tp, **kw).values
        big = np.abs(cs) > 1e-3 * np.abs(cs).max()
        dphi = np.angle(np.exp(1j * (np.asarray(got.values) - ref.values)))
        assert np.abs(dphi[big]).max() < (1e-7 if dtype == "float64" else 2e-2), np.abs(dphi[big]).max()
    else:
        cases.check(got, ref, 1e-10 if dtype == "float64" else 3e-4)
    return "fastn" if on else "other"


def run_random_fused_layout(seed, lo=16, hi=160):
    """Random power spectra / ffts over two transform axes that are NOT the trailing pair -- the independent elements innermost ((y, x, t) arrays) or between
    the axes ((y, t, x): dim = ["time", "lon"] of (time, lat, lon)) -- on random smooth / Rader lengths, element counts, leading batches, options, either order
    of the dims: the engine's two fused passes (csrc/fastn.h: fastn_cols_kernel, fastn_fit_inner_kernel, fastn_irows_kernel) or the composite of one-axis
    plans, against the oracle.  Returns "fused" | "other"."""
    import xrft_amd as xa

    rng = np.random.default_rng(seed)
    lens = smooth_lengths(lo, hi) + [n for n in (34, 51, 57, 58, 68, 73, 97, 146) if lo <= n <= hi]
    ny, nx = int(rng.choice(lens)), int(rng.choice(lens))
    ne = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 17, 33]))
    lay = str(rng.choice(["inner", "mid"]))
    if ne == 1 and lay == "inner":
        ne = 6
    nb = int(rng.choice([0, 0, 2, 3]))
    dtype = str(rng.choice(["float64", "float32"]))
    shape, dims = ((ny, nx, ne), ("y", "x", "t")) if lay == "inner" else ((ny, ne, nx), ("y", "t", "x"))
    if nb:
        shape, dims = (nb,) + shape, ("b",) + dims
    v = rng.standard_normal(shape)
    for d, n, sl in (("y", ny, 0.02), ("x", nx, -0.03)):
        sh = [1] * len(shape)
        sh[dims.index(d)] = n
        v = v + sl * np.arange(n).reshape(sh)
    v = (v + 1.0).astype(dtype)
    c = {d: np.arange(n) * float(rng.choice([0.5, 1.0, 2.0])) + float(rng.choice([0.0, 1.0, -3.0])) for d, n in zip(dims, shape)}
    da, od = cases.pair(v, dims, c)
    kw = dict(detrend=rng.choice([None, "constant", "linear"]), window=rng.choice([None, "hann", "hamming"]))
    dd = ["y", "x"] if rng.random() < 0.7 else ["x", "y"]
    shift = bool(rng.random() < 0.7)
    api._plan_cache.clear()
    if rng.random() < 0.6:
        sc = str(rng.choice(["density", "spectrum"]))
        got, ref = xa.power_spectrum(da, dim=dd, shift=shift, scaling=sc, **kw), o.power_spectrum(od, dim=dd, shift=shift, scaling=sc, **kw)
    else:
        tp = bool(rng.random() < 0.5)
        got, ref = xa.fft(da, dim=dd, shift=shift, true_phase=tp, **kw), o.fft(od, dim=dd, shift=shift, true_phase=tp, **kw)
    tol = (1e-10 if dtype == "float64" else 3e-4) * (10 if kw["detrend"] == "linear" else 1)  # (a plane fit on a trend of 5 x the noise: the residual's conditioning)
    cases.check(got, ref, tol)
    return "fused" if any("[fastn fused]" in p.describe() for p in api._plan_cache.values()) else "other"


@pytest.mark.parametrize("seed", range(int(os.environ.get("XRFT_RANDOM_FUSED_CASES", "24"))))
def test_random_fused_layout_case(seed):
    run_random_fused_layout(31000 + seed)


@pytest.mark.parametrize("seed", range(int(os.environ.get("XRFT_RANDOM_FASTN_CASES", "24"))))
def test_random_fastn_case(seed, monkeypatch):
    """csrc/fastn.h on the emulator: small slabs are kept off the one-pass kernel (XRFTHIP_FASTG=0) so that the two-pass pipeline with run-time
    radices serves them."""
    monkeypatch.setenv("XRFTHIP_FASTG", "0")
    run_random_fastn(seed, lo=16, hi=130, dtype="float64" if seed % 2 == 0 else "float32")


_ONE_AXIS_LENGTHS = (100, 128, 180, 200, 240, 256, 360, 400, 480, 500, 512, 600, 720, 800, 960, 1000, 1024, 1200, 1440, 2048, 4096)


def run_random_one_axis(seed):
    """Random one-axis calls on the lengths of the fastm table (csrc/fastm.h: fastm_yonly_kernel / fastm_xonly_kernel): which axis,
    batch and inner sizes (odd ones fall back to the generic kernels), precision, real / complex input, fft / power spectrum /
    cross spectrum / real_dim, detrend, window, shift, true phase -- whatever kernel serves the call, the oracle's numbers."""
    import xrft_amd as xa

    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice(_ONE_AXIS_LENGTHS))
    dtype = str(rng.choice(["float64", "float32"]))
    last = bool(rng.random() < 0.4)  # the transform axis is the contiguous one
    nb = int(rng.integers(1, 6))
    inner = int(rng.choice([8, 16, 24, 40, 7, 12])) if not last else 1
    shape = (nb, int(rng.integers(1, 5)), n) if last else (nb, n, inner)
    dims = ("t", "y", "x")
    ax = "x" if last else "y"
    v = rng.standard_normal(shape) + 1.5 + 2.0 * np.arange(n).reshape((1, 1, n) if last else (1, n, 1)) / n
    cplx = bool(rng.random() < 0.2)
    if cplx:
        v = v + 1j * rng.standard_normal(shape)
    v = v.astype(("complex128" if dtype == "float64" else "complex64") if cplx else dtype)
    c = {"t": np.arange(shape[0]), "y": np.arange(shape[1]) * 0.5 + float(rng.choice([0.0, 3.0])), "x": np.arange(shape[2]) * 0.25 - float(rng.choice([0.0, 2.0]))}
    da, od = cases.pair(v, dims, c)
    kw = dict(detrend=rng.choice([None, "constant", "linear"]), window=rng.choice([None, "hann", "hamming"]))
    kind = str(rng.choice(["fft", "ps", "cs", "ps_real"] if not cplx else ["fft", "ps"]))
    shift = bool(rng.random() < 0.7)
    tp = bool(rng.random() < 0.5)
    api._plan_cache.clear()
    if kind == "fft":
        got, ref = xa.fft(da, dim=[ax], shift=shift, true_phase=tp, **kw), o.fft(od, dim=[ax], shift=shift, true_phase=tp, **kw)
    elif kind == "ps":
        got, ref = xa.power_spectrum(da, dim=[ax], shift=shift, **kw), o.power_spectrum(od, dim=[ax], shift=shift, **kw)
    elif kind == "cs":
        w = rng.standard_normal(shape).astype(dtype)
        db, ob = cases.pair(w, dims, c)
        got, ref = xa.cross_spectrum(da, db, dim=[ax], shift=shift, true_phase=tp, **kw), o.cross_spectrum(od, ob, dim=[ax], shift=shift, true_phase=tp, **kw)
    else:
        got, ref = xa.power_spectrum(da, dim=[ax], real_dim=ax, **kw), o.power_spectrum(od, dim=[ax], real_dim=ax, **kw)
    cases.check(got, ref, 1e-10 if dtype == "float64" else 3e-4)
    return any("[fastm " in p.describe() for p in api._plan_cache.values())


@pytest.mark.parametrize("seed", range(int(os.environ.get("XRFT_RANDOM_ONE_AXIS_CASES", "16"))))
def test_random_one_axis_case(seed):
    run_random_one_axis(seed)


def run_random_one_pass(seed, big=False):
    """Random calls on what the register-resident one-pass kernels serve (round 4): real float32 slabs of 64 | 128 | 256 points per axis
    (csrc/fasts.h: power spectrum, fft, isotropic power spectrum; the other modes must come out right through the other kernels) and real
    float32 rows of 4096 ... 65536 samples (csrc/fastr.h: fft / power spectrum, full or real_dim half).  Descending coordinates, every
    detrend / window / shift / scaling / true-phase combination.  Returns the one-pass kernel that served the call (or None)."""
    import xrft_amd as xa

    rng = np.random.default_rng(11000 + seed)
    kw = dict(detrend=rng.choice([None, "constant", "linear"]), window=rng.choice([None, "hann", "hamming", "blackman"]))
    shift = bool(rng.random() < 0.7)
    tp = bool(rng.random() < 0.5)
    api._plan_cache.clear()
    if rng.random() < 0.6:
        ny, nx = int(rng.choice([64, 128, 256])), int(rng.choice([64, 128, 256]))
        nb = int(rng.integers(1, 700 if big else 4))
        v = rng.standard_normal((nb, ny, nx)).astype(np.float32)
        v += (0.01 * np.arange(ny, dtype=np.float32))[None, :, None] + (-0.02 * np.arange(nx, dtype=np.float32) + 3)[None, None, :]
        v *= (1 + (np.arange(nb) % 5).astype(np.float32))[:, None, None]
        yc = np.arange(ny) * float(rng.choice([0.5, 1.0])) + float(rng.choice([0.0, 2.0]))
        desc = bool(rng.random() < 0.15)
        c = {"t": np.arange(nb), "y": yc[::-1].copy() if desc else yc, "x": np.arange(nx) * float(rng.choice([0.25, 1.0])) - float(rng.choice([0.0, 3.0]))}
        da, od = cases.pair(v, ("t", "y", "x"), c)
        kind = str(rng.choice(["ps", "ps", "fft", "iso", "iso", "cs", "ps_real"]))
        if kind == "ps":
            sc = str(rng.choice(["density", "spectrum"]))
            wc = bool(kw["window"] is not None and rng.random() < 0.4)
            got, ref = (xa.power_spectrum(da, dim=["y", "x"], shift=shift, scaling=sc, window_correction=wc, **kw),
                        o.power_spectrum(od, dim=["y", "x"], shift=shift, scaling=sc, window_correction=wc, **kw))
        elif kind == "fft":
            got, ref = xa.fft(da, dim=["y", "x"], shift=shift, true_phase=tp, **kw), o.fft(od, dim=["y", "x"], shift=shift, true_phase=tp, **kw)
        elif kind == "iso":
            tr = bool(rng.random() < 0.5)
            nf = int(rng.choice([4, 4, 2, 8]))
            got, ref = (xa.isotropic_power_spectrum(da, dim=["y", "x"], truncate=tr, nfactor=nf, **kw),
                        o.isotropic_power_spectrum(od, dim=["y", "x"], truncate=tr, nfactor=nf, **kw))
        elif kind == "cs":
            w = rng.standard_normal((nb, ny, nx)).astype(np.float32)
            db, ob = cases.pair(w, ("t", "y", "x"), c)
            got, ref = xa.cross_spectrum(da, db, dim=["y", "x"], shift=shift, true_phase=tp, **kw), o.cross_spectrum(od, ob, dim=["y", "x"], shift=shift, true_phase=tp, **kw)
        else:
            got, ref = xa.power_spectrum(da, dim=["y"], real_dim="x", **kw), o.power_spectrum(od, dim=["y"], real_dim="x", **kw)
        served = "[fasts]"
        expect = kind in ("ps", "iso") or (kind == "fft" and not (desc and tp))  # (a flipped axis is not the one-pass kernel's; iso: any bin map, radial or not)
        if kind == "iso":
            expect = None  # (truncate=True leaves unbinned corners: not a radial map -> the other kernels; both must be right)
    else:
        n = int(rng.choice([4096, 8192, 16384, 32768, 65536]))
        nb = int(rng.integers(1, 40 if big else 3))
        v = (rng.standard_normal((nb, n)) + 2.0 + 1e-4 * np.arange(n)).astype(np.float32)
        xc = np.arange(n) * 0.25 + float(rng.choice([0.0, 5.0]))
        desc = bool(rng.random() < 0.15)
        da, od = cases.pair(v, ("t", "x"), {"t": np.arange(nb), "x": xc[::-1].copy() if desc else xc})
        od64 = o.OArr(v.astype("float64"), ("t", "x"), {"t": np.arange(nb), "x": xc[::-1].copy() if desc else xc})
        kind = str(rng.choice(["fft", "ps", "fft_real", "ps_real"]))
        if kind == "fft":
            got, ref = xa.fft(da, dim=["x"], shift=shift, true_phase=tp, **kw), o.fft(od64, dim=["x"], shift=shift, true_phase=tp, **kw)
        elif kind == "ps":
            sc = str(rng.choice(["density", "spectrum"]))
            got, ref = xa.power_spectrum(da, dim=["x"], shift=shift, scaling=sc, **kw), o.power_spectrum(od64, dim=["x"], shift=shift, scaling=sc, **kw)
        elif kind == "fft_real":
            got, ref = xa.fft(da, dim=["x"], real_dim="x", true_phase=tp, **kw), o.fft(od64, dim=["x"], real_dim="x", true_phase=tp, **kw)
        else:
            got, ref = xa.power_spectrum(da, dim=["x"], real_dim="x", **kw), o.power_spectrum(od64, dim=["x"], real_dim="x", **kw)
        served = "[fastr]"
        expect = not (desc and tp and kind in ("fft", "fft_real"))
    on = any(served in p.describe() for p in api._plan_cache.values())
    if expect is not None:
        assert on == expect, (kind, served, on, [p.describe()[:200] for p in api._plan_cache.values()])
    # (the rows are held against the float64-fed oracle: a trend of 1e-4 x 65536 beside unit noise in float32, cases.run_fourstep_1d)
    err = float(np.abs(np.asarray(got.values) - ref.values).max() / np.abs(ref.values).max())
    assert err < 3e-4, (kind, err)
    for d in ref.dims:
        if d in ref.coords:
            assert np.array_equal(np.asarray(got[d].values), np.asarray(ref.coord(d)), equal_nan=True), d
    return served if on else None


@pytest.mark.parametrize("seed", range(int(os.environ.get("XRFT_RANDOM_ONE_PASS_CASES", "14"))))
def test_random_one_pass_case(seed):
    run_random_one_pass(seed)


# (smooth, odd, with prime factors that take Bluestein -- 131, 94 = 2 x 47, 262 -- and ONE prime 17 ... 127 with a smooth p - 1: the Rader forms -- 146, 365, 366, 73, 97, 58, 68, 206)
_ANY_AXIS_LENGTHS = [6, 12, 27, 45, 48, 50, 75, 96, 120, 125, 150, 243, 250, 77, 131, 146, 365, 366, 73, 97, 58, 68, 94, 262, 206]


def run_random_any_axis(seed):
    """Random one-axis calls on lengths OUTSIDE the mixed-radix table -- smooth, odd, Bluestein -- along a first / middle / last axis, real or complex input, fft /
    power spectrum / cross spectrum / real_dim / ifft of the call's own spectrum: the lengths-as-data kernels of csrc/fastg.h (fastgy_kernel, fastg on groups of
    rows) or whatever else serves the call, against the oracle."""
    import warnings

    import xrft_amd as xa

    rng = np.random.default_rng(11000 + seed)
    n = int(rng.choice(_ANY_AXIS_LENGTHS))
    dtype = str(rng.choice(["float64", "float32"]))
    where = int(rng.integers(0, 3))  # the transform axis
    sizes = [int(rng.choice([1, 2, 3, 5])), int(rng.choice([2, 4, 6, 7])), int(rng.choice([2, 8, 10, 13]))]
    sizes[where] = n
    shape = tuple(sizes)
    dims = ("t", "y", "x")
    ax = dims[where]
    rs = [1, 1, 1]
    rs[where] = n
    v = rng.standard_normal(shape) + 1.5 + 2.0 * np.arange(n).reshape(rs) / n
    cplx = bool(rng.random() < 0.25)
    if cplx:
        v = v + 1j * rng.standard_normal(shape)
    v = v.astype(("complex128" if dtype == "float64" else "complex64") if cplx else dtype)
    c = {d: np.arange(s) * float(rng.choice([0.5, 1.0])) + float(rng.choice([0.0, 3.0])) for d, s in zip(dims, shape)}
    da, od = cases.pair(v, dims, c)
    kw = dict(detrend=rng.choice([None, "constant", "linear"]), window=rng.choice([None, "hann", "hamming"]))
    kind = str(rng.choice(["fft", "ps", "cs", "ps_real", "inverse"] if not cplx else ["fft", "ps", "inverse"]))
    shift = bool(rng.random() < 0.7)
    tp = bool(rng.random() < 0.5)
    tol = 1e-10 if dtype == "float64" else 3e-4
    api._plan_cache.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if kind == "fft":
            got, ref = xa.fft(da, dim=[ax], shift=shift, true_phase=tp, **kw), o.fft(od, dim=[ax], shift=shift, true_phase=tp, **kw)
        elif kind == "ps":
            got, ref = xa.power_spectrum(da, dim=[ax], shift=shift, **kw), o.power_spectrum(od, dim=[ax], shift=shift, **kw)
        elif kind == "cs":
            w = rng.standard_normal(shape).astype(dtype)
            db, ob = cases.pair(w, dims, c)
            got, ref = xa.cross_spectrum(da, db, dim=[ax], shift=shift, true_phase=tp, **kw), o.cross_spectrum(od, ob, dim=[ax], shift=shift, true_phase=tp, **kw)
        elif kind == "ps_real":
            got, ref = xa.power_spectrum(da, dim=[ax], real_dim=ax, **kw), o.power_spectrum(od, dim=[ax], real_dim=ax, **kw)
        else:
            F, Fo = xa.fft(da, dim=[ax], shift=shift, true_phase=tp), o.fft(od, dim=[ax], shift=shift, true_phase=tp)
            got, ref = xa.ifft(F, dim=["freq_" + ax], shift=shift, true_phase=tp), o.ifft(Fo, dim=["freq_" + ax], shift=shift, true_phase=tp)
    if dtype == "float64" or kind == "inverse" or cplx or kw["detrend"] is None:
        cases.check(got, ref, tol)
    else:  # (float32 with a detrend: the oracle on the float64 copy of the samples, as everywhere -- cases.pair)
        cases.check(got, ref, tol)
    return [t for t in ("[fastg y-only]", "[fastg rows]", "[fastm ", "[main]") if any(t in p.describe() for p in api._plan_cache.values())]


@pytest.mark.parametrize("seed", range(int(os.environ.get("XRFT_RANDOM_ANY_AXIS_CASES", "48"))))
def test_random_any_axis_case(seed):
    run_random_any_axis(seed)


def run_random_inverse(seed):
    """Random inverse transforms over two axes (xrft.ifft of the call's own xrft.fft, with and without real_dim) on small and medium slabs of any smooth shape:
    one pass, two one-pass stages, or the generic passes -- against the oracle."""
    import warnings

    import xrft_amd as xa

    rng = np.random.default_rng(13000 + seed)
    dtype = str(rng.choice(["float64", "float32"]))
    ny = int(rng.choice(_smooth_lengths(2, 150)))
    nx = int(rng.choice(_smooth_lengths(4, 200, even=bool(rng.random() < 0.7))))
    nb = int(rng.integers(1, 5))
    v = (rng.standard_normal((nb, ny, nx)) + 0.5).astype(dtype)
    c = {"t": np.arange(nb), "y": np.arange(ny) * float(rng.choice([0.5, 1.0])) + float(rng.choice([0.0, 2.0])),
         "x": np.arange(nx) * float(rng.choice([0.25, 1.0])) - float(rng.choice([0.0, 3.0]))}
    da, od = cases.pair(v, ("t", "y", "x"), c)
    shift = bool(rng.random() < 0.7)
    tp = bool(rng.random() < 0.5)
    real = bool(rng.random() < 0.4) and nx % 2 == 0
    api._plan_cache.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if real:
            F, Fo = xa.fft(da, dim=["y"], real_dim="x", shift=shift, true_phase=tp), o.fft(od, dim=["y"], real_dim="x", shift=shift, true_phase=tp)
            got, ref = xa.ifft(F, dim=["freq_y"], real_dim="freq_x", shift=shift, true_phase=tp), o.ifft(Fo, dim=["freq_y"], real_dim="freq_x", shift=shift, true_phase=tp)
        else:
            F, Fo = xa.fft(da, dim=["y", "x"], shift=shift, true_phase=tp), o.fft(od, dim=["y", "x"], shift=shift, true_phase=tp)
            got, ref = xa.ifft(F, dim=["freq_y", "freq_x"], shift=shift, true_phase=tp), o.ifft(Fo, dim=["freq_y", "freq_x"], shift=shift, true_phase=tp)
    cases.check(got, ref, 1e-10 if dtype == "float64" else 3e-4)
    return [t for t in ("[fastg] one pass", "[fastg y-only]", "[fastg rows]") if any(t in p.describe() for p in api._plan_cache.values())]


@pytest.mark.parametrize("seed", range(int(os.environ.get("XRFT_RANDOM_INVERSE_CASES", "32"))))
def test_random_inverse_case(seed):
    run_random_inverse(seed)
