"""Shared parity cases: every case builds seeded inputs, runs the product API (xrft_amd, bound to whichever
library the calling test selected) and the CPU oracle, and returns (got, ref).  Used by

  tests/test_emulated_api.py   CPU: product host code + kernels compiled for the emulator (index arithmetic)
  tests/test_gpu_parity.py     GPU: the real libxrft_hip.so through the C ABI  (-m gpu)

Tolerances follow BASELINE.json: 1e-6 relative in float64, 1e-3 in float32 (relative to max |reference|);
the tests assert much tighter float64 agreement (1e-10).
"""
import warnings

import numpy as np
import numpy.testing as npt
import pytest

import xrft_amd as xa
from oracle import xrft_oracle as o

warnings.simplefilter("ignore")

TOL = {"float64": 1e-10, "float32": 3e-4, "complex128": 1e-10, "complex64": 3e-4}


def pair(data, dims, coords=None):
    """The product's array and the oracle's, from the same samples.  float32 / complex64 samples go to the oracle as float64 /
    complex128 (the same values): the yardstick of the float32 path is what the reference's algorithm gives in exact-enough
    arithmetic, not the reference's own float32 rounding (its float32 `da - da.mean()` and scipy detrend leave up to 5e-3 of
    relative error in bins 1e-6 of the peak -- tests/cases.py fine_errors -- which would otherwise be charged to the product)."""
    a = np.asarray(data)
    ref = a.astype(np.float64) if a.dtype == np.float32 else (a.astype(np.complex128) if a.dtype == np.complex64 else a)
    return xa.DataArray(data, dims, coords), o.OArr(ref, dims, coords)


def rel_err(got, ref):
    g = np.asarray(got.values)
    r = np.asarray(ref.values)
    assert g.shape == r.shape, (g.shape, r.shape)
    den = max(float(np.abs(r).max()), 1e-300)
    return float(np.abs(g - r).max()) / den


# float32 results are also held to bounds that do not hide behind the largest bin (a power spectrum spans many decades):
#   * every bin above BIN_FLOOR x max |reference| to BIN_REL relative error.  The floor is 1e-6 of the maximum for non-negative
#     (power-like) results = 1e-3 in amplitude, and 1e-3 for signed / complex results, whose small values are differences of large
#     ones: a float32 transform carries ~1e-7 of the spectrum's rms into every output, in the reference's float32 path as here;
#   * the L1 norm of the error against the L1 norm of the reference, to the same tolerance as the maximum norm.
BIN_REL = 1e-3


def fine_errors(g, r):
    """(worst relative error over the bins above the floor, L1 error / L1 reference) of a float32 result."""
    g = np.asarray(g)
    r = np.asarray(r)
    fin = np.isfinite(r)
    a = np.abs(np.where(fin, r, 0))
    d = np.abs(np.where(fin, g - r, 0))
    mx = float(a.max()) if a.size else 0.0
    power_like = r.dtype.kind == "f" and bool((np.where(fin, r, 0) >= 0).all())
    big = a > (1e-6 if power_like else 1e-3) * mx
    binrel = float((d[big] / a[big]).max()) if big.any() else 0.0
    l1 = float(d.sum()) / max(float(a.sum()), 1e-300)
    return binrel, l1


def check(got, ref, tol, bin_rel=BIN_REL):
    assert tuple(got.dims) == tuple(ref.dims), (got.dims, ref.dims)
    for d in ref.dims:
        if d in ref.coords:
            gv = np.asarray(got[d].values)
            rv = np.asarray(ref.coord(d))
            # frequency / bin-centre coordinates are host float64 arithmetic on both sides: bit for bit
            assert gv.shape == rv.shape and gv.dtype == rv.dtype, (d, gv.dtype, rv.dtype)
            assert np.array_equal(gv, rv, equal_nan=rv.dtype.kind in "fc"), d
            ra = ref.coord_attrs.get(d, {})
            for k, v in ra.items():
                assert k in got[d].attrs, (d, k)
                np.testing.assert_allclose(float(got[d].attrs[k]), float(v), rtol=1e-13)
    err = rel_err(got, ref)
    assert err < tol, f"rel err {err:.3e} >= {tol:.1e}"
    if tol > 1e-8:  # float32 tolerances: the finer norms
        binrel, l1 = fine_errors(got.values, ref.values)
        l1_tol = max(tol, TOL["float32"])  # (callers that hold the max norm tighter than 3e-4 do so on spectra led by one huge bin)
        assert l1 < l1_tol, f"L1 err {l1:.3e} >= {l1_tol:.1e}"
        assert binrel < bin_rel, f"worst per-bin rel err {binrel:.3e} >= {bin_rel:.1e}"
    return err


def _cube(rng, shape, dtype, trend=True):
    nt, ny, nx = shape
    v = rng.standard_normal(shape)
    if trend:
        ii, jj = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
        v = v + 0.05 * ii - 0.03 * jj + 2.0
    if np.dtype(dtype).kind == "c":
        v = v + 1j * rng.standard_normal(shape)
    return v.astype(dtype)


def _coords3(shape, dy=0.5, dx=2.0, y0=0.0, x0=3.0):
    nt, ny, nx = shape
    return {"time": np.arange(nt), "y": np.arange(ny) * dy + y0, "x": np.arange(nx) * dx + x0}


D3 = ("time", "y", "x")

# name -> (shape, function(da, od) -> (got, ref)); dtype is applied by the runner
CASES = {}


def case(name, shape=(3, 16, 24), dtypes=("float64", "float32"), trend=True):
    def deco(fn):
        CASES[name] = (shape, dtypes, trend, fn)
        return fn
    return deco


@case("fft2d_default")
def _(da, od): return xa.fft(da, dim=["y", "x"]), o.fft(od, dim=["y", "x"])
@case("fft2d_complex_in", dtypes=("complex128", "complex64"))
def _(da, od): return xa.fft(da, dim=["y", "x"]), o.fft(od, dim=["y", "x"])
@case("fft2d_noshift_nophase")
def _(da, od): return (xa.fft(da, dim=["y", "x"], shift=False, true_phase=False, true_amplitude=False),
                       o.fft(od, dim=["y", "x"], shift=False, true_phase=False, true_amplitude=False))
@case("dft2d_linear_hann")
def _(da, od): return (xa.dft(da, dim=["y", "x"], detrend="linear", window="hann"),
                       o.dft(od, dim=["y", "x"], detrend="linear", window="hann"))
@case("fft1d_x_linear_hann")
def _(da, od): return (xa.fft(da, dim=["x"], detrend="linear", window="hann"),
                       o.fft(od, dim=["x"], detrend="linear", window="hann"))
@case("fft1d_middle_axis_constant")
def _(da, od): return xa.fft(da, dim=["y"], detrend="constant"), o.fft(od, dim=["y"], detrend="constant")
@case("fft1d_first_axis", shape=(8, 6, 10))
def _(da, od): return xa.fft(da, dim=["time"], shift=False), o.fft(od, dim=["time"], shift=False)
@case("rfft2d_real_x")
def _(da, od): return xa.fft(da, dim=["y"], real_dim="x"), o.fft(od, dim=["y"], real_dim="x")
@case("rfft2d_real_y")
def _(da, od): return xa.fft(da, dim=["x"], real_dim="y"), o.fft(od, dim=["x"], real_dim="y")
@case("rfft1d_constant")
def _(da, od): return (xa.dft(da, dim="x", real_dim="x", detrend="constant"),
                       o.dft(od, dim="x", real_dim="x", detrend="constant"))
@case("fft2d_odd_sizes", shape=(2, 15, 9))
def _(da, od): return (xa.fft(da, dim=["y", "x"], detrend="linear", window="hamming"),
                       o.fft(od, dim=["y", "x"], detrend="linear", window="hamming"))
@case("fft2d_prime_sizes", shape=(2, 17, 13))
def _(da, od): return xa.fft(da, dim=["y", "x"]), o.fft(od, dim=["y", "x"])
@case("fft1d_mixed_2601", shape=(2, 1, 2601))
def _(da, od): return xa.fft(da, dim=["x"], detrend="constant"), o.fft(od, dim=["x"], detrend="constant")
@case("rfft1d_odd", shape=(2, 3, 45))
def _(da, od): return xa.fft(da, dim=["x"], real_dim="x"), o.fft(od, dim=["x"], real_dim="x")
@case("ps2d_linear_hann_density")
def _(da, od): return (xa.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"),
                       o.power_spectrum(od, dim=["y", "x"], detrend="linear", window="hann"))
@case("ps2d_constant_hann_spectrum_wc")
def _(da, od): return (xa.power_spectrum(da, dim=["y", "x"], detrend="constant", window="hann", scaling="spectrum",
                                         window_correction=True),
                       o.power_spectrum(od, dim=["y", "x"], detrend="constant", window="hann", scaling="spectrum",
                                        window_correction=True))
@case("ps2d_density_false")
def _(da, od): return (xa.power_spectrum(da, dim=["y", "x"], density=False, window="bartlett"),
                       o.power_spectrum(od, dim=["y", "x"], density=False, window="bartlett"))
@case("ps2d_real_dim")
def _(da, od): return (xa.power_spectrum(da, dim=["y"], real_dim="x", detrend="linear", window="hann"),
                       o.power_spectrum(od, dim=["y"], real_dim="x", detrend="linear", window="hann"))
@case("ps1d_real_dim_periodogram")
def _(da, od): return (xa.power_spectrum(da, dim="x", real_dim="x", detrend="constant"),
                       o.power_spectrum(od, dim="x", real_dim="x", detrend="constant"))
@case("ps1d_time_axis", shape=(20, 6, 10))
def _(da, od): return (xa.power_spectrum(da, dim=["time"], window="hann"),
                       o.power_spectrum(od, dim=["time"], window="hann"))
@case("ps2d_noshift_descending")
def _(da, od): return xa.power_spectrum(da, dim=["y", "x"], shift=False), o.power_spectrum(od, dim=["y", "x"], shift=False)
@case("iso_ps_constant_hann")
def _(da, od): return (xa.isotropic_power_spectrum(da, dim=["y", "x"], detrend="constant", window="hann"),
                       o.isotropic_power_spectrum(od, dim=["y", "x"], detrend="constant", window="hann"))
@case("iso_ps_truncate_nfactor2", shape=(2, 32, 32))
def _(da, od): return (xa.isotropic_power_spectrum(da, dim=["y", "x"], truncate=True, nfactor=2),
                       o.isotropic_power_spectrum(od, dim=["y", "x"], truncate=True, nfactor=2))
@case("isotropize_existing_ps")
def _(da, od): return (xa.isotropize(xa.power_spectrum(da, dim=["y", "x"]), ["freq_y", "freq_x"]),
                       o.isotropize(o.power_spectrum(od, dim=["y", "x"]), ["freq_y", "freq_x"]))
@case("detrend_linear_2d")
def _(da, od): return xa.detrend(da, ["y", "x"], "linear"), o.detrend(od, ["y", "x"], "linear").transpose(*D3)
@case("detrend_linear_1d")
def _(da, od): return xa.detrend(da, "x", "linear"), o.detrend(od, "x", "linear")
@case("detrend_constant_2d", dtypes=("float64", "float32", "complex128"))
def _(da, od): return xa.detrend(da, ["y", "x"], "constant"), o.detrend(od, ["y", "x"], "constant")


def run_case(name, dtype, seed=0):
    shape, dtypes, trend, fn = CASES[name]
    rng = np.random.default_rng(seed)
    data = _cube(rng, shape, dtype, trend)
    da, od = pair(data, D3, _coords3(shape))
    got, ref = fn(da, od)
    return check(got, ref, TOL[dtype])


def all_case_params():
    return [(n, dt) for n, (_, dts, _, _) in CASES.items() for dt in dts]


# ---- two-field cases (cross spectra) -----------------------------------------------------------------
def run_cross_case(kind, dtype, seed=1):
    rng = np.random.default_rng(seed)
    shape = (2, 16, 24)
    a = _cube(rng, shape, dtype)
    b = _cube(rng, shape, dtype)
    c1 = _coords3(shape)
    c2 = _coords3(shape, y0=1.5, x0=-4.0)
    da, od = pair(a, D3, c1)
    db, ob = pair(b, D3, c2)
    if kind == "true_phase_window":
        kw = dict(dim=["y", "x"], window="hann", detrend="constant")
    elif kind == "nophase_spectrum":
        kw = dict(dim=["y", "x"], true_phase=False, scaling="spectrum")
    elif kind == "real_dim":
        kw = dict(dim=["y"], real_dim="x", detrend="linear")
    elif kind == "one_dim":
        kw = dict(dim=["x"], window="hann", window_correction=True)
    elif kind in ("opposite_x", "opposite_y_mid_axis"):
        # the two fields' coordinates run in opposite directions: the reference flips each by its own coordinate, after the
        # window (xrft.py:425-441) -- per-field flip flags in the engine (XRFTHIP_FLIP0_* / XRFTHIP_FLIP_*)
        c3 = dict(c2)
        if kind == "opposite_x":
            c3["x"] = c2["x"][::-1].copy()
            kw = dict(dim=["y", "x"], window="hann", detrend="linear")
        else:
            c3["y"] = c2["y"][::-1].copy()
            kw = dict(dim=["y"], window="hann")
        db, ob = pair(b, D3, c3)
        e1 = check(xa.cross_spectrum(da, db, **kw), o.cross_spectrum(od, ob, **kw), TOL[dtype])
        e2 = check(xa.cross_spectrum(db, da, **kw), o.cross_spectrum(ob, od, **kw), TOL[dtype])
        return max(e1, e2)
    elif kind == "iso":
        got = xa.isotropic_cross_spectrum(da, db, dim=["y", "x"], window="hann", detrend="linear")
        ref = o.isotropic_cross_spectrum(od, ob, dim=["y", "x"], window="hann", detrend="linear")
        return check(got, ref, TOL[dtype])
    else:
        raise KeyError(kind)
    return check(xa.cross_spectrum(da, db, **kw), o.cross_spectrum(od, ob, **kw), TOL[dtype])


CROSS_KINDS = ["true_phase_window", "nophase_spectrum", "real_dim", "one_dim", "iso", "opposite_x", "opposite_y_mid_axis"]


# ---- true-phase cases with descending / offset coordinates --------------------------------------------
def run_true_phase_case(dtype, seed=2):
    rng = np.random.default_rng(seed)
    N = 20
    v = (rng.random(N) + 1j * rng.random(N)).astype("complex128" if dtype == "float64" else "complex64")
    x = np.arange(N // 2, -N // 2, -1) + 2  # descending (test_xrft.py:1336-1347)
    da, od = pair(v, ("x",), {"x": x})
    e1 = check(xa.dft(da, dim="x", true_phase=True), o.dft(od, dim="x", true_phase=True), TOL[dtype])
    v2 = rng.random((12, 10)).astype(dtype)
    c = {"y": np.arange(12, 0, -1) * 0.5, "x": np.arange(10) * 0.3 - 7.0}
    da, od = pair(v2, ("y", "x"), c)
    e2 = check(xa.fft(da), o.fft(od), TOL[dtype])
    return max(e1, e2)


# ---- widened rows: ifft / idft, cross_phase, chunks_to_segments ---------------------------------
def run_inverse_cases(dtype="float64"):
    """xrft.ifft / idft against the oracle (xrft.py:479-646) and as round trips (test_xrft.py:1253-1312)."""
    cdt = "complex128" if dtype == "float64" else "complex64"
    tol = TOL[dtype]
    rng = np.random.default_rng(8)
    errs = []
    # 1-D complex round trip, shifted and unshifted spectra (test_ifft_fft)
    N = 20
    v = (rng.random(N) + 1j * rng.random(N)).astype(cdt)
    s, so = pair(v, ("x",), {"x": np.arange(0, N)})
    for sh in (True, False):
        F, Fo = xa.fft(s, shift=sh), o.fft(so, shift=sh)
        errs.append(check(xa.ifft(F, shift=True), o.ifft(Fo, shift=True), tol))
        assert np.abs(xa.ifft(F, shift=True).values - v).max() < 50 * tol
    # idft(dft) with explicit and automatic lag (test_idft_dft)
    N, dx = 40, 0.37
    v = (rng.random(N) + 1j * rng.random(N)).astype(cdt)
    x = dx * (np.arange(-N // 2, -N // 2 + N) + 7)
    s, so = pair(v, ("x",), {"x": x})
    F, Fo = xa.dft(s, true_phase=True, true_amplitude=True), o.dft(so, true_phase=True, true_amplitude=True)
    lagv = float(x[N // 2])
    errs.append(check(xa.idft(F, shift=True, true_phase=True, true_amplitude=True, lag=lagv),
                      o.idft(Fo, shift=True, true_phase=True, true_amplitude=True, lag=lagv), tol))
    back = xa.idft(F, shift=True, true_phase=True, true_amplitude=True)
    assert np.abs(back.values - v).max() < 50 * tol and np.allclose(back["x"].values, x)
    # 2-D complex, batch, odd x odd sizes, default arguments
    v = (rng.standard_normal((3, 9, 15)) + 1j * rng.standard_normal((3, 9, 15))).astype(cdt)
    c = {"t": np.arange(3), "y": np.arange(9) * 0.5 - 2, "x": np.arange(15) * 0.25 + 1}
    s, so = pair(v, ("t", "y", "x"), c)
    F, Fo = xa.fft(s, dim=["y", "x"]), o.fft(so, dim=["y", "x"])
    errs.append(check(xa.ifft(F, dim=["freq_y", "freq_x"]), o.ifft(Fo, dim=["freq_y", "freq_x"]), tol))
    assert np.abs(xa.ifft(F, dim=["freq_y", "freq_x"]).values - v).max() < 100 * tol
    # real_dim: rfft -> irfft (the half spectrum is Hermitian-extended on the device)
    v = rng.standard_normal((2, 8, 12)).astype(dtype)
    c = {"t": np.arange(2), "y": np.arange(8) * 0.5, "x": np.arange(12) * 0.25}
    s, so = pair(v, ("t", "y", "x"), c)
    F, Fo = (f(a, dim=["y"], real_dim="x", true_phase=False, true_amplitude=False) for f, a in ((xa.fft, s), (o.fft, so)))
    kw = dict(dim=["freq_y"], real_dim="freq_x", true_phase=False, true_amplitude=False, lag=[0.0, 0.0], shift=True)
    errs.append(check(xa.ifft(F, **kw), o.ifft(Fo, **kw), tol))
    assert np.abs(xa.ifft(F, **kw).values - v).max() < 100 * tol
    # the same with the default true phase / amplitude (input phase table on the stored half of the real axis)
    F, Fo = xa.fft(s, dim=["y"], real_dim="x"), o.fft(so, dim=["y"], real_dim="x")
    errs.append(check(xa.ifft(F, dim=["freq_y"], real_dim="freq_x"), o.ifft(Fo, dim=["freq_y"], real_dim="freq_x"), tol))
    assert np.abs(xa.ifft(F, dim=["freq_y"], real_dim="freq_x").values - v).max() < 100 * tol
    F1, F1o = xa.fft(s, dim="x", real_dim="x"), o.fft(so, dim="x", real_dim="x")
    errs.append(check(xa.ifft(F1, dim="freq_x", real_dim="freq_x"), o.ifft(F1o, dim="freq_x", real_dim="freq_x"), tol))
    # not centred on zero frequency -> ValueError (test_idft_centered_coordinates)
    import pytest
    bad, _ = pair((rng.random(20) + 0j).astype(cdt), ("freq_x",), {"freq_x": np.arange(-10, 10) + 2})
    with pytest.raises(ValueError):
        xa.idft(bad)
    return max(errs)


def run_cross_phase_cases(dtype="float64"):
    """xrft.cross_phase (xrft.py:838-874; test_xrft.py:606-690)."""
    tol = 1e-9 if dtype == "float64" else 2e-3  # the angle of a near-zero cross spectrum amplifies rounding
    N = 32
    x = np.linspace(0, 1, num=N, endpoint=False)
    f, po = 6, np.pi / 2
    a1 = xa.DataArray(np.cos(2 * np.pi * f * x).astype(dtype), ("x",), {"x": x}, name="a")
    a2 = xa.DataArray(np.cos(2 * np.pi * f * x - po).astype(dtype), ("x",), {"x": x}, name="b")
    cp = xa.cross_phase(a1, a2, dim=["x"])
    assert cp.name == "a_b_phase" and cp.dims == ("freq_x",)
    i = int(np.argmin(np.abs(cp["freq_x"].values - f)))
    assert abs(float(cp.values[i]) - po) < (1e-6 if dtype == "float64" else 1e-3)
    rng = np.random.default_rng(4)
    shape = (2, 12, 10)
    v1, v2 = rng.standard_normal(shape).astype(dtype), rng.standard_normal(shape).astype(dtype)
    c1 = _coords3(shape)
    c2 = _coords3(shape, y0=1.5, x0=-4.0)
    d1, o1 = pair(v1, D3, c1)
    d2, o2 = pair(v2, D3, c2)
    got = xa.cross_phase(d1, d2, dim=["y", "x"], true_phase=True, window="hann")
    ref = o.cross_phase(o1, o2, dim=["y", "x"], true_phase=True, window="hann")
    assert got.dims == ref.dims
    d = np.angle(np.exp(1j * (got.values - ref.values)))  # compare angles modulo 2 pi
    assert np.abs(d).max() < tol, np.abs(d).max()
    return float(np.abs(d).max())


def run_segment_cases(dtype="float64"):
    """chunks_to_segments (xrft.py:106-136, 390-391; test_xrft.py:273-337): the chunk length is metadata here."""
    import pytest
    tol = TOL[dtype]
    rng = np.random.default_rng(12)
    N = 32
    v = rng.random((N, N, N)).astype(dtype)
    c = {"time": np.arange(N), "y": np.arange(N), "x": np.arange(N)}
    da, od = pair(v, D3, c)
    ft = xa.fft(da.chunk({"time": 16}), dim=["time"], shift=False, chunks_to_segments=True)
    assert ft.dims == ("time_segment", "freq_time", "y", "x")
    e1 = check(ft, o.fft(od.chunk({"time": 16}), dim=["time"], shift=False, chunks_to_segments=True), tol)
    ft2 = xa.fft(da.chunk({"y": 16, "x": 16}), dim=["y", "x"], shift=False, chunks_to_segments=True)
    assert ft2.dims == ("time", "y_segment", "freq_y", "x_segment", "freq_x")
    e2 = check(ft2, o.fft(od.chunk({"y": 16, "x": 16}), dim=["y", "x"], shift=False, chunks_to_segments=True), tol)
    ps = xa.power_spectrum(da.chunk({"y": 16, "x": 16}), dim=["y", "x"], window="hann", window_correction=True,
                           chunks_to_segments=True)
    e3 = check(ps, o.power_spectrum(od.chunk({"y": 16, "x": 16}), dim=["y", "x"], window="hann", window_correction=True,
                                    chunks_to_segments=True), tol)
    with pytest.raises(ValueError):
        xa.fft(da.chunk({"time": 20}), dim=["time"], detrend="linear", chunks_to_segments=True)
    with pytest.raises(ValueError):  # several chunks along a transform dim without chunks_to_segments (test_xrft.py:166-170)
        xa.fft(da.chunk({"x": 1}), dim=["x"])
    return max(e1, e2, e3)


# ---------------------------------------------------------------------------------- SURVEY 8 f4
def run_nd_cases(dtype):
    """Transforms over three and four axes (the reference hands all axes to fftn at once, xrft.py:439-447)."""
    tol = TOL[dtype]
    rng = np.random.default_rng(77)
    shape = (2, 6, 8, 10)
    v = rng.standard_normal(shape).astype(dtype)
    ii, jj, kk = np.meshgrid(np.arange(6), np.arange(8), np.arange(10), indexing="ij")
    v = v + (0.3 * ii - 0.2 * jj + 0.1 * kk + 2.0).astype(dtype)[None]
    coords = {"t": np.arange(2), "z": np.arange(6) * 0.5 + 1.0, "y": np.arange(8) * 2.0 - 3.0, "x": np.arange(10) * 0.25}
    da, od = pair(v, ("t", "z", "y", "x"), coords)
    d3 = ["z", "y", "x"]
    for kw in (dict(), dict(detrend="linear", window="hann"), dict(true_phase=False, shift=False, detrend="constant"),
               dict(true_amplitude=False, window="hamming")):
        check(xa.fft(da, dim=d3, **kw), o.fft(od, dim=d3, **kw), tol)
    check(xa.fft(da, dim=["x", "z", "y"], detrend="linear"), o.fft(od, dim=["x", "z", "y"], detrend="linear"), tol)
    check(xa.fft(da, dim=d3, real_dim="x", window="hann"), o.fft(od, dim=d3, real_dim="x", window="hann"), tol)
    check(xa.fft(da, detrend="constant"), o.fft(od, detrend="constant"), tol)  # all four axes
    for kw in (dict(), dict(detrend="linear", window="hann", window_correction=True), dict(scaling="spectrum", window="hann", window_correction=True),
               dict(scaling="false_density", detrend="constant")):
        check(xa.power_spectrum(da, dim=d3, **kw), o.power_spectrum(od, dim=d3, **kw), tol)
    w = rng.standard_normal(shape).astype(dtype)
    c2 = dict(coords, z=coords["z"] + 0.5)
    db, ob = pair(w, ("t", "z", "y", "x"), c2)
    for kw in (dict(), dict(detrend="linear", window="hann"), dict(true_phase=False)):
        check(xa.cross_spectrum(da, db, dim=d3, **kw), o.cross_spectrum(od, ob, dim=d3, **kw), tol)
    with pytest.raises(NotImplementedError):
        xa.fft(da, detrend="linear")  # 4-D linear detrend: "Only 1D, 2D, and 3D detrending"
    # real_dim in spectra over three axes, inverse transforms over three axes
    check(xa.power_spectrum(da, dim=d3, real_dim="x", window="hann"), o.power_spectrum(od, dim=d3, real_dim="x", window="hann"), tol)
    check(xa.cross_spectrum(da, db, dim=d3, real_dim="x"), o.cross_spectrum(od, ob, dim=d3, real_dim="x"), tol)
    fd = ["freq_z", "freq_y", "freq_x"]
    ft, oft = xa.fft(da, dim=d3), o.fft(od, dim=d3)
    check(xa.ifft(ft, dim=fd), o.ifft(oft, dim=fd), tol)
    check(xa.ifft(ft, dim=fd, true_phase=False, shift=True, lag=[0.0, 0.0, 0.0]), o.ifft(oft, dim=fd, true_phase=False, shift=True, lag=[0.0, 0.0, 0.0]), tol)
    ft, oft = xa.fft(da, dim=d3, real_dim="x"), o.fft(od, dim=d3, real_dim="x")
    check(xa.ifft(ft, dim=fd, real_dim="freq_x"), o.ifft(oft, dim=fd, real_dim="freq_x"), tol)
    back = xa.ifft(xa.fft(da, dim=d3), dim=fd)
    assert np.abs(back.values.real - v).max() < (1e-10 if dtype == "float64" else 2e-4)


def run_detrend3_cases(dtype):
    tol = TOL["float32" if dtype == "float32" else "float64"]
    rng = np.random.default_rng(78)
    for shape, dims in (((3, 6, 5, 4), ["z", "y", "x"]), ((2, 9, 16, 12), ["x", "z", "y"]), ((4, 1, 7, 3), ["z", "y", "x"])):
        v = rng.standard_normal(shape)
        ii, jj, kk = np.meshgrid(*[np.arange(n) for n in shape[1:]], indexing="ij")
        v = v + (0.3 * ii - 0.7 * jj + 0.11 * kk + 5.0)[None]
        if dtype.startswith("complex"):
            v = v + 1j * (rng.standard_normal(shape) - 0.2 * ii[None] + 0.05 * kk[None])
        v = v.astype(dtype)
        da, od = pair(v, ("t", "z", "y", "x"))
        for kind in ("constant", "linear"):
            if dtype.startswith("complex") and kind == "linear":
                ref = o.detrend(o.OArr(v.real, od.dims), dims, kind).transpose("t", "z", "y", "x").values + 1j * o.detrend(
                    o.OArr(v.imag, od.dims), dims, kind).transpose("t", "z", "y", "x").values
            else:
                r = o.detrend(od, dims, kind)
                ref = r.transpose("t", "z", "y", "x").values if r.dims != od.dims else r.values
            got = xa.detrend(da, dims, kind)
            assert tuple(got.dims) == ("t", "z", "y", "x")
            err = np.abs(got.values - ref).max() / max(np.abs(ref).max(), 1e-300)
            assert err < tol, (shape, dims, kind, err)


def run_pad_cases():
    rng = np.random.default_rng(79)
    v = rng.standard_normal((3, 5, 6))
    coords = {"t": np.arange(3), "y": np.arange(5) * 0.5 - 1.0, "x": np.arange(6) * 2.0 + 10.0}
    for data in (v, None):
        if data is None:
            import torch

            arr = xa.DataArray(torch.from_numpy(v.copy()), ("t", "y", "x"), coords)
        else:
            arr = xa.DataArray(v, ("t", "y", "x"), coords)
        od = o.OArr(v, ("t", "y", "x"), coords)
        for kw, mode in ((dict(x=2, y=1), "constant"), (dict(x=(1, 4)), "constant"), (dict(y=(0, 3), x=(2, 0)), "edge"),
                         (dict(x=3), "wrap"), (dict(y=2, x=2), "reflect"), (dict(x=(2, 1)), "symmetric")):
            got = xa.pad(arr, mode=mode, **kw)
            ref = o.pad(od, mode=mode, **kw)
            npt.assert_array_equal(got.values, ref.values)
            for d in ("y", "x"):
                npt.assert_array_equal(got[d].values, ref.coord(d))
                if d in kw:
                    assert got[d].attrs["pad_width"] == kw[d]
            back = xa.unpad(got)
            npt.assert_array_equal(back.values, v)
            for d in ("y", "x"):
                npt.assert_array_equal(back[d].values, coords[d])
                assert "pad_width" not in back[d].attrs
        got = xa.pad(arr, x=2, constant_values=7.5)
        npt.assert_array_equal(got.values, o.pad(od, x=2, constant_values=7.5).values)
        npt.assert_array_equal(xa.unpad(xa.pad(arr, x=2, y=1), x=1, y=1).values, np.pad(v, ((0, 0), (0, 0), (1, 1))))
        npt.assert_array_equal(xa.pad(arr, x=2, mode="mean").values, np.pad(v, ((0, 0), (0, 0), (2, 2)), mode="mean"))
        npt.assert_array_equal(xa.pad(arr, x=2, mode="linear_ramp", end_values=1.0).values,
                               np.pad(v, ((0, 0), (0, 0), (2, 2)), mode="linear_ramp", end_values=1.0))
    # the padded array transforms like any other (test_padding.py:207-235: fft of a padded array, unpad after ifft)
    da1 = xa.DataArray(v[0, 0], ("x",), {"x": coords["x"]})
    p = xa.pad(da1, x=4)
    ft = xa.fft(p, true_phase=True)
    back = xa.unpad(xa.ifft(ft, lag=[ft["freq_x"].attrs["direct_lag"]]), x=4)
    npt.assert_allclose(back.values.real, v[0, 0], atol=1e-12)
    npt.assert_allclose(back["x"].values, coords["x"], atol=1e-12)


def run_bluestein_cases(dtype):
    """Lengths with a prime factor above 128 (chirp-z inside the tile kernel); numpy's pocketfft takes any length."""
    tol = TOL[dtype]
    # float32 chirp-z inside the tile: round 3 carried a 4e-3 per-bin exception here; measured since (profiles/r04_bluestein_f32.txt) these
    # cases hold 6e-5 in float32 arithmetic, ERA5-like slabs 1.2e-4: the per-bin bound of every other path
    br = BIN_REL
    rng = np.random.default_rng(131)
    for n in (131, 257, 262, 1801, 4099):
        v = rng.standard_normal((3, n)).astype(dtype) + 0.01 * np.arange(n, dtype=dtype)[None]
        da, od = pair(v, ("t", "x"), {"t": np.arange(3), "x": np.arange(n) * 0.5 + 2.0})
        for kw in (dict(), dict(detrend="linear", window="hann"), dict(true_phase=False, shift=False)):
            check(xa.fft(da, dim="x", **kw), o.fft(od, dim="x", **kw), tol, br)
        check(xa.power_spectrum(da, dim="x", window="hann"), o.power_spectrum(od, dim="x", window="hann"), tol, br)
        check(xa.fft(da, dim="x", real_dim="x"), o.fft(od, dim="x", real_dim="x"), tol, br)
    z = (rng.standard_normal((2, 139)) + 1j * rng.standard_normal((2, 139))).astype("complex128" if dtype == "float64" else "complex64")
    da, od = pair(z, ("t", "x"), {"t": np.arange(2), "x": np.arange(139) * 1.0})
    check(xa.fft(da, dim="x"), o.fft(od, dim="x"), tol, br)
    check(xa.ifft(xa.fft(da, dim="x"), dim="freq_x"), o.ifft(o.fft(od, dim="x"), dim="freq_x"), tol, br)
    for shape in ((2, 131, 24), (2, 20, 262), (1, 149, 137)):
        v = rng.standard_normal(shape).astype(dtype)
        c = {"t": np.arange(shape[0]), "y": np.arange(shape[1]) * 1.0, "x": np.arange(shape[2]) * 2.0}
        da, od = pair(v, ("t", "y", "x"), c)
        check(xa.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"),
              o.power_spectrum(od, dim=["y", "x"], detrend="linear", window="hann"), tol, br)
        check(xa.fft(da, dim=["y", "x"]), o.fft(od, dim=["y", "x"]), tol, br)


def run_composite_lengths(dtype):
    """Lengths whose factorisation uses every in-register butterfly (2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 16) and a prime one (7, 11)."""
    tol = TOL[dtype]
    rng = np.random.default_rng(15)
    cdt = "complex128" if dtype == "float64" else "complex64"
    for n in (6, 9, 10, 12, 15, 18, 36, 45, 60, 90, 100, 120, 144, 150, 225, 360, 720, 1440, 2160, 77, 1155):
        z = (rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))).astype(cdt)
        da, od = pair(z, ("t", "x"), {"t": np.arange(2), "x": np.arange(n) * 0.5})
        check(xa.fft(da, dim="x", true_phase=False), o.fft(od, dim="x", true_phase=False), tol)
        r = rng.standard_normal((2, n)).astype(dtype)
        da, od = pair(r, ("t", "x"), {"t": np.arange(2), "x": np.arange(n) * 0.5})
        check(xa.power_spectrum(da, dim="x", window="hann"), o.power_spectrum(od, dim="x", window="hann"), tol)
    v = rng.standard_normal((2, 90, 120)).astype(dtype)
    da, od = pair(v, ("t", "y", "x"), {"t": np.arange(2), "y": np.arange(90) * 0.25, "x": np.arange(120) * 0.25})
    check(xa.power_spectrum(da, dim=["y", "x"], detrend="constant", window="hann"),
          o.power_spectrum(od, dim=["y", "x"], detrend="constant", window="hann"), tol)


# ---- long 1-D real float32 sequences: the two y-first passes as the two steps of a four-step transform (fasty.h, FS) ----
def run_fourstep_1d(n=65536, nt=3):
    """xrft.fft / dft / power_spectrum along one long axis (BASELINE.json configs[1] is (1024, 65536)): plain, shifted or not,
    true phase, detrended; a window takes the generic path.  The float32 reference detrends in float32 (scipy), which costs
    it 5e-5 on a 65536-point line: the detrended cases are held against the oracle on float64 input."""
    rng = np.random.default_rng(31)
    v = (rng.standard_normal((nt, n)) + 2.0 + 1e-4 * np.arange(n)).astype("float32")
    c = {"t": np.arange(nt), "x": np.arange(n) * 0.25 + 5.0}
    da, od = pair(v, ("t", "x"), c)
    od64 = o.OArr(v.astype("float64"), ("t", "x"), c)
    worst = 0.0
    # 8192 ... 65536 samples = one row per workgroup, in registers, ONE pass (csrc/fastr.h); longer rows: the two four-step passes (fasty.h)
    tag = "[fastr]" if n <= 65536 else "four-step]"
    for kw in (dict(), dict(true_phase=False, shift=False), dict(true_amplitude=False, shift=False)):
        worst = max(worst, check(xa.fft(da, dim=["x"], **kw), o.fft(od, dim=["x"], **kw), 3e-6))
    assert tag in next(reversed(xa.api._plan_cache.values())).describe()
    if n <= 65536:  # real_dim: the half spectrum k = 0..n/2, its power spectrum counted twice inside (xrft.py:400-404, 673-682)
        for kw in (dict(), dict(true_phase=False)):
            worst = max(worst, check(xa.fft(da, dim=["x"], real_dim="x", **kw), o.fft(od, dim=["x"], real_dim="x", **kw), 3e-6))
            assert tag in next(reversed(xa.api._plan_cache.values())).describe()
        worst = max(worst, check(xa.power_spectrum(da, dim=["x"], real_dim="x"), o.power_spectrum(od, dim=["x"], real_dim="x"), 3e-6))
        assert tag in next(reversed(xa.api._plan_cache.values())).describe()
        g = xa.power_spectrum(da, dim=["x"], real_dim="x", detrend="linear", window="hann")
        assert tag in next(reversed(xa.api._plan_cache.values())).describe()
        r = o.power_spectrum(od64, dim=["x"], real_dim="x", detrend="linear", window="hann")
        assert float(np.abs(g.values - r.values).max() / np.abs(r.values).max()) < 2e-5
    worst = max(worst, check(xa.dft(da, dim="x"), o.dft(od, dim="x"), 3e-6))
    for kw in (dict(), dict(scaling="spectrum", shift=False)):
        worst = max(worst, check(xa.power_spectrum(da, dim=["x"], **kw), o.power_spectrum(od, dim=["x"], **kw), 3e-6))
    for det in ("constant", "linear"):
        g = xa.fft(da, dim=["x"], detrend=det)
        r = o.fft(od64, dim=["x"], detrend=det)
        err = float(np.abs(g.values - r.values).max() / np.abs(r.values).max())
        assert err < 2e-5, (det, err)  # (the trend reaches 100x the noise at 2^20 samples: float32 input)
        g = xa.power_spectrum(da, dim=["x"], detrend=det)
        r = o.power_spectrum(od64, dim=["x"], detrend=det)
        assert float(np.abs(g.values - r.values).max() / np.abs(r.values).max()) < 2e-5
    # a window of the whole sequence is not separable over the [n1][256] view: it rides on a slab-shaped table, and the transforms
    # that carry the residual trend back are per column (fasty.h, W2D) -- still the two four-step kernels (round 2: generic passes)
    for kw in (dict(window="hann"), dict(window="hann", detrend="linear"), dict(window="hamming", detrend="constant", shift=False)):
        g = xa.power_spectrum(da, dim=["x"], **kw)
        assert tag in next(reversed(xa.api._plan_cache.values())).describe(), kw
        r = o.power_spectrum(od64, dim=["x"], **kw)
        e = float(np.abs(g.values - r.values).max() / np.abs(r.values).max())
        binrel, l1 = fine_errors(g.values, r.values)
        assert e < 2e-5 and l1 < 3e-4 and binrel < BIN_REL, (kw, e, binrel, l1)
        worst = max(worst, e)
    for kw in (dict(window="hann", detrend="linear"), dict(window="blackman", true_phase=False, shift=False)):
        g = xa.fft(da, dim=["x"], **kw)
        assert tag in next(reversed(xa.api._plan_cache.values())).describe(), kw
        r = o.fft(od64, dim=["x"], **kw)
        e = float(np.abs(g.values - r.values).max() / np.abs(r.values).max())
        assert e < 2e-5, (kw, e)
    return worst


def run_fastm_cases(shape=(2, 360, 360), full=True, cross=True, dtype="float64"):
    """float64 / float32 slabs on the regular lat/lon lengths (360 / 720 / 1440; BASELINE.json configs[4] is (64, 1440, 720) float64
    with a linear detrend and a Hann window): the mixed-radix y-first kernels (csrc/fastm.h) against the oracle -- power spectra with every
    detrend / window / shift combination, the complex spectrum with true phase on offset coordinates, cross spectrum, cross phase."""
    rng = np.random.default_rng(41)
    tol = TOL[dtype]
    a = _cube(rng, shape, dtype)
    c1 = _coords3(shape, y0=1.0, x0=-3.0)
    da, od = pair(a, D3, c1)
    worst = 0.0

    def on_fastm():  # (the two-pass y-first pipeline: the table kernels of fastm.h, or -- 3000 / 3600 / 4320 since round 5 -- fastn.h's lengths-as-data kernels on that axis)
        d_ = next(reversed(xa.api._plan_cache.values())).describe()
        return "[fastm]" in d_ or "[fastn]" in d_

    worst = max(worst, check(xa.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"),
                             o.power_spectrum(od, dim=["y", "x"], detrend="linear", window="hann"), tol))
    assert on_fastm()
    if not full:
        return worst
    for kw in (dict(), dict(detrend="constant", window="hamming", scaling="spectrum", window_correction=True), dict(detrend="linear"),
               dict(shift=False, window="bartlett", density=False)):
        worst = max(worst, check(xa.power_spectrum(da, dim=["y", "x"], **kw), o.power_spectrum(od, dim=["y", "x"], **kw), tol))
        assert on_fastm()
    for kw in (dict(), dict(shift=False, true_phase=False, true_amplitude=False), dict(detrend="linear", window="hann")):
        worst = max(worst, check(xa.fft(da, dim=["y", "x"], **kw), o.fft(od, dim=["y", "x"], **kw), tol))
        assert on_fastm()
    if not cross:  # (1440 x 1440: one row pair per workgroup leaves no room for the second field -- the generic kernels take it)
        return worst
    b = _cube(rng, shape, dtype)
    db, ob = pair(b, D3, _coords3(shape, y0=-2.5, x0=4.0))
    for kw in (dict(window="hann", detrend="linear"), dict(true_phase=False, scaling="spectrum", shift=False)):
        worst = max(worst, check(xa.cross_spectrum(da, db, dim=["y", "x"], **kw), o.cross_spectrum(od, ob, dim=["y", "x"], **kw), tol))
        assert on_fastm()
    # real_dim: half output along x (rows of nx/2 + 1 samples)
    for fn, ofn, kwr in ((xa.power_spectrum, o.power_spectrum, dict(detrend="linear", window="hann")), (xa.fft, o.fft, dict(detrend="constant")),
                         (xa.power_spectrum, o.power_spectrum, dict(shift=False, scaling="spectrum"))):
        worst = max(worst, check(fn(da, dim=["y"], real_dim="x", **kwr), ofn(od, dim=["y"], real_dim="x", **kwr), tol))
        assert on_fastm()
    worst = max(worst, check(xa.cross_spectrum(da, db, dim=["y"], real_dim="x", window="hann"), o.cross_spectrum(od, ob, dim=["y"], real_dim="x", window="hann"), tol))
    assert on_fastm()
    # isotropic spectra (doc/MITgcm_example.ipynb: isotropic_powerspectrum with detrend='linear', window=True): the spectrum is
    # stored by the same kernels and summed by the bit-reproducible radial pass
    kwi = dict(dim=["y", "x"], detrend="linear", window="hann")
    ips = xa.isotropic_power_spectrum(da, **kwi)
    assert on_fastm()
    worst = max(worst, check(ips, o.isotropic_power_spectrum(od, **kwi), tol))
    assert np.array_equal(xa.isotropic_power_spectrum(da, **kwi).values, ips.values)
    ics = xa.isotropic_cross_spectrum(da, db, **kwi)
    assert on_fastm()
    worst = max(worst, check(ics, o.isotropic_cross_spectrum(od, ob, **kwi), tol))
    g = xa.cross_phase(da, db, dim=["y", "x"], detrend="constant")
    r = o.cross_phase(od, ob, dim=["y", "x"], detrend="constant")
    dphi = np.abs(np.angle(np.exp(1j * (g.values - r.values))))
    mag = np.abs(o.cross_spectrum(od, ob, dim=["y", "x"], detrend="constant").values)
    # (the angle of a near-zero cross spectrum amplifies rounding -- the detrended mean bin is pure rounding, its angle 0 or pi:
    # the error is held relative to the sample's magnitude, and absolutely wherever the sample is not tiny)
    lim = (1e-10, 1e-6) if dtype == "float64" else (3e-4, 3e-1)  # (float32: 1e-6 of the maximum is already rounding noise)
    big = mag > (1e-6 if dtype == "float64" else 1e-2) * mag.max()
    assert (dphi * mag).max() / mag.max() < lim[0] and dphi[big].max() < lim[1], ((dphi * mag).max() / mag.max(), dphi[big].max())
    assert on_fastm()
    return worst


def adversarial_detrend_fields(n, rng):
    """float32 slabs (n x n) that stress the float32 detrending of the two-pass kernels, which subtract a per-column line
    ESTIMATED from a few reference rows before the transform (round 2: rows 0, 1, n-2, n-1; now the medians of the three rows
    around n/4 and around 3n/4) and add the difference to the exact least-squares plane (xrft/detrend.py:100-113) back in the
    spectral domain (csrc/fasty.h, fastm.h): outliers in those rows, offsets and trends far above the signal, constant columns."""
    ii, jj = np.meshgrid(np.arange(n, dtype=np.float64), np.arange(n, dtype=np.float64), indexing="ij")
    noise = rng.standard_normal((n, n))
    out = {}
    v = noise.copy(); v[0, :] += 1e3; v[1, ::2] -= 5e2; v[-1, :] += 7e2; v[-2, ::3] -= 1e3
    out["spikes_in_the_edge_rows"] = v
    v = noise.copy(); v[n // 4, :] += 1e3; v[3 * n // 4, ::2] -= 8e2; v[3 * n // 4 + 1, 1::2] += 6e2
    out["spikes_in_the_quarter_rows"] = v
    v = noise.copy(); v[:2, :] += 50.0; v[-2:, :] -= 80.0
    out["steps_in_the_edge_rows"] = v
    out["offset_1e6"] = noise + 1e6
    v = noise.copy(); v[:, n // 3] = 4.0; v[:, 0] = 0.0; v[:, -1] = -2.5
    out["constant_columns"] = v
    out["trend_1e4_times_noise"] = noise + 1e4 * (0.7 * ii / n - 0.4 * jj / n) + 3e3
    v = noise.copy(); v[: n // 2, :] += 1e2
    out["half_slab_step"] = v
    return {k: v.astype(np.float32) for k, v in out.items()}


def run_adversarial_detrend(n, detrends=("linear", "constant"), window="hann", seed=77, only=None):
    """power_spectrum of the adversarial slabs against the oracle, all norms of `check`; returns the worst errors.  The oracle
    gets the same float32 samples as float64: the reference's own float32 arithmetic is no yardstick here (`da - da.mean()` in
    float32 leaves the ky = 0 row of the 1e6-offset slab 35x off; DESIGN.md 2, deviations: the trend is fitted in float64)."""
    rng = np.random.default_rng(seed)
    fields = adversarial_detrend_fields(n, rng)
    names = [k for k in fields if only is None or k in only]
    v = np.stack([fields[k] for k in names])
    c = {"time": np.arange(len(names)), "y": np.arange(n) * 0.5, "x": np.arange(n) * 2.0}
    worst = {}
    for det in detrends:
        kw = dict(dim=["y", "x"], detrend=det, window=window)
        got = xa.power_spectrum(xa.DataArray(v, D3, c), **kw)
        ref = o.power_spectrum(o.OArr(v.astype(np.float64), D3, c), **kw)
        g, r = np.asarray(got.values), np.asarray(ref.values)
        for t, k in enumerate(names):
            e = float(np.abs(g[t] - r[t]).max() / np.abs(r[t]).max())
            binrel, l1 = fine_errors(g[t], r[t])
            worst[(det, k)] = (e, binrel, l1)
            assert e < 1e-3 and l1 < 1e-3 and binrel < BIN_REL, (det, k, e, binrel, l1)
    return worst


def run_radial_sum_cases(big=False):
    """Radial bin sums (xrft.isotropize / isotropic_*_spectrum; reference xrft.py:877-1010) outside the specialised kernels: the
    stand-alone sum and the generic plans.  Values against numpy.bincount / the oracle, any number of bins (the C ABI used to
    stop at 4096; the tables of a launch cover a window of bins), and bit-identical repeats (integer fixed-point sums)."""
    import torch

    from xrft_amd import engine

    rng = np.random.default_rng(51)
    nb = 7000  # more bins than one launch's LDS window (5461 real / 3276 complex)
    ny, nx, nt = 48, 160, 3
    bm = rng.integers(-1, nb, size=(ny, nx)).astype(np.int32)
    dev = xa.api._to_device(np.zeros(1, dtype=np.float32)).device
    bmd = torch.from_numpy(bm).to(dev)
    for dt in ("float32", "float64", "complex64", "complex128"):
        v = rng.standard_normal((nt, ny, nx)) * np.exp(3.0 * rng.standard_normal((nt, ny, nx)))  # ten decades of dynamic range
        if dt.startswith("complex"):
            v = v + 1j * rng.standard_normal((nt, ny, nx))
        v = v.astype(dt)
        t = torch.from_numpy(v).to(dev)
        got = engine.isotropize(t, bmd, nb)
        again = engine.isotropize(t, bmd, nb)
        assert torch.equal(got, again), dt
        g = got.cpu().numpy()
        ok = bm.ravel() >= 0
        for b in range(nt):
            w = v[b].ravel()[ok].astype("complex128" if dt.startswith("complex") else "float64")
            ref = np.bincount(bm.ravel()[ok], weights=w.real, minlength=nb)
            if dt.startswith("complex"):
                ref = ref + 1j * np.bincount(bm.ravel()[ok], weights=w.imag, minlength=nb)
            mag = np.bincount(bm.ravel()[ok], weights=np.abs(w), minlength=nb)
            assert np.all(np.abs(g[b] - ref) <= 1e-12 * np.maximum(mag, 1e-300)), (dt, b, np.abs(g[b] - ref).max())
    # inf / nan members are what they are in a floating-point sum (the fixed-point tables used to turn a NaN into +inf):
    # nan poisons its bin, +inf alone gives +inf, +inf and -inf give nan; the other bins and slabs stay exact
    for dt in ("float32", "float64", "complex128"):
        v = rng.standard_normal((2, ny, nx)).astype(dt)
        bm2 = (np.arange(ny * nx).reshape(ny, nx) % 50).astype(np.int32)
        v[0, 0, 3] = np.nan; v[0, 0, 4] = np.inf; v[0, 0, 5] = -np.inf; v[0, 0, 6] = np.inf; v[0, 1, 6] = -np.inf
        if dt == "complex128":
            v[1, 0, 7] = complex(1.0, np.inf)
        g = engine.isotropize(torch.from_numpy(v).to(dev), torch.from_numpy(bm2).to(dev), 50).cpu().numpy()
        with np.errstate(invalid="ignore"):
            vv = v.astype("complex128" if dt.startswith("complex") else "float64")
            ref = np.stack([np.array([vv[b].ravel()[bm2.ravel() == k].sum() for k in range(50)]) for b in range(2)])
        assert np.array_equal(np.isnan(g.real), np.isnan(ref.real)) and np.array_equal(np.isnan(g.imag), np.isnan(ref.imag)), dt
        assert np.array_equal(np.isinf(g), np.isinf(ref)) and np.array_equal(np.sign(g.real[np.isinf(g.real)]), np.sign(ref.real[np.isinf(ref.real)])), dt
        fin = np.isfinite(ref)
        npt.assert_allclose(g[fin], ref[fin], rtol=1e-10, atol=1e-10)
    # generic plans (float64, lengths the specialised kernels do not take): values vs the oracle, repeats bit for bit
    shape = (3, 48, 40)
    a = _cube(rng, shape, "float64")
    b = _cube(rng, shape, "float64")
    da, od = pair(a, D3, _coords3(shape))
    db, ob = pair(b, D3, _coords3(shape))
    kw = dict(dim=["y", "x"], detrend="linear", window="hann")
    ips = xa.isotropic_power_spectrum(da, **kw)
    worst = check(ips, o.isotropic_power_spectrum(od, **kw), TOL["float64"])
    assert np.array_equal(xa.isotropic_power_spectrum(da, **kw).values, ips.values)
    ics = xa.isotropic_cross_spectrum(da, db, **kw)
    worst = max(worst, check(ics, o.isotropic_cross_spectrum(od, ob, **kw), TOL["float64"]))
    assert np.array_equal(xa.isotropic_cross_spectrum(da, db, **kw).values, ics.values)
    iso = xa.isotropize(xa.power_spectrum(da, dim=["y", "x"]), ["freq_y", "freq_x"])
    assert np.array_equal(xa.isotropize(xa.power_spectrum(da, dim=["y", "x"]), ["freq_y", "freq_x"]).values, iso.values)
    if big:  # more than 4096 radial bins through the public call: nfactor = 1 on a 4400^2 spectrum
        n = 4400
        ps = (rng.standard_normal((1, n, n)) ** 2).astype("float32")
        c = {"t": np.arange(1), "freq_y": np.fft.fftshift(np.fft.fftfreq(n, 0.5)), "freq_x": np.fft.fftshift(np.fft.fftfreq(n, 0.5))}
        dps, ops = pair(ps, ("t", "freq_y", "freq_x"), c)
        got = xa.isotropize(dps, ["freq_y", "freq_x"], nfactor=1)
        ref = o.isotropize(ops, ["freq_y", "freq_x"], nfactor=1)
        assert got.sizes["freq_r"] == ref.values.shape[-1] > 4000
        worst = max(worst, check(got, ref, 1e-6))
    return worst


def check_values(got, ref, tol):
    """check() when the oracle ran on a float64 copy of float32 data: dims and coordinates as usual, values by relative error."""
    assert tuple(got.dims) == tuple(ref.dims)
    g, r = np.asarray(got.values), np.asarray(ref.values)
    if g.dtype == r.dtype:
        return check(got, ref, tol)
    for d in got.dims:
        assert np.array_equal(np.asarray(got[d].values), np.asarray(ref.coord(d)), equal_nan=True), d
    err = float(np.abs(g - r).max() / max(float(np.abs(r).max()), 1e-300))
    assert g.shape == r.shape and err < tol, f"rel err {err:.3e} >= {tol:.1e}"
    binrel, l1 = fine_errors(g, r)
    assert l1 < max(tol, TOL["float32"]) and binrel < BIN_REL, f"L1 err {l1:.3e} (< {max(tol, TOL['float32']):.1e}), worst per-bin rel err {binrel:.3e} (< {BIN_REL:.1e})"
    return err


def run_yonly_any_length_cases(shape=(3, 96, 40), dtype="float32"):
    """One transform axis that is not the contiguous one on ANY smooth length (the reference's `dim="time"` calls on lengths outside fastm.h's table:
    48, 96, 120, 150, 250, 1250, odd lengths): csrc/fastg.h fastgy_kernel (run-time radices, two real columns per complex sequence) against the oracle --
    fft (true phase on an offset coordinate, also ifftshifted), power spectrum, every per-column detrend, window, shift; a column count that does not fill
    the last workgroup; the first axis of a 3-D array."""
    rng = np.random.default_rng(67)
    tol = TOL[dtype]
    a = _cube(rng, shape, dtype)
    da, od = pair(a, D3, _coords3(shape, y0=2.5, x0=-1.0))
    od_det = od if dtype == "float64" else o.OArr(a.astype("float64"), D3, _coords3(shape, y0=2.5, x0=-1.0))  # (as run_yonly_fast_cases)
    worst = 0.0

    def on_fast():
        return "[fastg y-only]" in next(reversed(xa.api._plan_cache.values())).describe()

    for kw in (dict(), dict(detrend="linear", window="hann"), dict(detrend="constant", shift=False), dict(true_phase=False, true_amplitude=False, window="hamming")):
        worst = max(worst, check_values(xa.fft(da, dim=["y"], **kw), o.fft(od_det if "detrend" in kw else od, dim=["y"], **kw), tol))
        assert on_fast(), kw
    for kw in (dict(detrend="linear", window="hann"), dict(shift=False, scaling="spectrum", detrend="constant"), dict()):
        worst = max(worst, check_values(xa.power_spectrum(da, dim=["y"], **kw), o.power_spectrum(od_det if "detrend" in kw else od, dim=["y"], **kw), tol))
        assert on_fast(), kw
    # real_dim along the axis (ABI 0.1.4: the half output of the one-pass kernels -- no transposed copy; xrft.py:400-404, 673-682)
    for kw in (dict(detrend="linear", window="hann"), dict(scaling="spectrum", detrend="constant"), dict()):
        worst = max(worst, check_values(xa.power_spectrum(da, dim=["y"], real_dim="y", **kw), o.power_spectrum(od_det if "detrend" in kw else od, dim=["y"], real_dim="y", **kw), tol))
        assert on_fast(), kw
    worst = max(worst, check_values(xa.fft(da, dim=["y"], real_dim="y", detrend="linear", true_phase=False), o.fft(od_det, dim=["y"], real_dim="y", detrend="linear", true_phase=False), tol))
    assert on_fast()
    # complex input (the later stages of N-D transforms, xrft.fft of complex data): one sequence per column, any column count
    cdt = "complex128" if dtype == "float64" else "complex64"
    z = (a + 1j * _cube(rng, shape, dtype)).astype(cdt)[:, :, : max(1, shape[2] - 1)]  # (an odd column count where the shape allows)
    cz = _coords3((shape[0], shape[1], z.shape[2]), y0=2.5, x0=-1.0)
    dz, oz = pair(z, D3, cz)
    oz_det = oz if dtype == "float64" else o.OArr(z.astype("complex128"), D3, cz)
    for kw in (dict(), dict(detrend="linear", window="hann", shift=False)):
        worst = max(worst, check_values(xa.fft(dz, dim=["y"], **kw), o.fft(oz_det if "detrend" in kw else oz, dim=["y"], **kw), tol))
        assert on_fast(), kw
    worst = max(worst, check_values(xa.power_spectrum(dz, dim=["y"], detrend="constant"), o.power_spectrum(oz_det, dim=["y"], detrend="constant"), tol))
    assert on_fast()
    # and back: xrft.ifft along the axis where it lies (conj in, conj out, the fftshifted input rotated on load, the lag's phase on the input)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for kw in (dict(), dict(true_phase=False, shift=False), dict(true_phase=False), dict(shift=False)):
            F, Fo = xa.fft(da, dim=["y"], **kw), o.fft(od, dim=["y"], **kw)
            worst = max(worst, check_values(xa.ifft(F, dim=["freq_y"], **kw), o.ifft(Fo, dim=["freq_y"], **kw), tol))
            assert on_fast(), kw
    # two fields: cross spectrum and cross phase along the axis (a column of each field = the two halves of one packed sequence; any column count)
    b2 = _cube(rng, shape, dtype)[:, :, : max(1, shape[2] - 1)]
    a2 = a[:, :, : b2.shape[2]]
    c2a, c2b = _coords3(a2.shape, y0=2.5, x0=-1.0), _coords3(a2.shape, y0=-1.5, x0=-1.0)
    d2a, o2a = pair(a2, D3, c2a)
    d2b, o2b = pair(b2, D3, c2b)
    o2a_det = o2a if dtype == "float64" else o.OArr(a2.astype("float64"), D3, c2a)
    o2b_det = o2b if dtype == "float64" else o.OArr(b2.astype("float64"), D3, c2b)
    for kw in (dict(detrend="linear", window="hann"), dict(true_phase=False, shift=False, scaling="spectrum")):
        det = "detrend" in kw
        worst = max(worst, check_values(xa.cross_spectrum(d2a, d2b, dim=["y"], **kw), o.cross_spectrum(o2a_det if det else o2a, o2b_det if det else o2b, dim=["y"], **kw), tol))
        assert on_fast(), kw
    gph = xa.cross_phase(d2a, d2b, dim=["y"], detrend="constant")
    rph = o.cross_phase(o2a, o2b, dim=["y"], detrend="constant")
    assert on_fast()
    dphi = np.abs(np.angle(np.exp(1j * (gph.values - rph.values))))
    mag = np.abs(o.cross_spectrum(o2a, o2b, dim=["y"], detrend="constant").values)
    assert (dphi * mag).max() / mag.max() < (1e-10 if dtype == "float64" else 3e-4), (dphi * mag).max() / mag.max()
    # the same samples with the transform axis FIRST, (time, y, x): columns = y x
    at = np.ascontiguousarray(a.transpose(1, 0, 2))
    ct = {"time": np.arange(shape[1]) * 0.5 + 2.5, "y": np.arange(shape[0]) * 1.0, "x": np.arange(shape[2]) * 2.0}
    dt_, ot_ = pair(at, D3, ct)
    ot_det = ot_ if dtype == "float64" else o.OArr(at.astype("float64"), D3, ct)
    worst = max(worst, check_values(xa.power_spectrum(dt_, dim=["time"], detrend="linear", window="hann"), o.power_spectrum(ot_det, dim=["time"], detrend="linear", window="hann"), tol))
    assert on_fast()
    return worst


def run_yonly_fast_cases(shape=(3, 360, 40), dtype="float64"):
    """One transform axis that is not the contiguous one (spectra along "time" of a (batch, time, space) array), lengths of the
    fastm table: csrc/fastm.h fastm_yonly_kernel against the oracle -- fft (true phase on an offset, also ifftshifted,
    coordinate), power spectrum, every detrend along the axis, window, shift."""
    rng = np.random.default_rng(61)
    # (float32 at 2048+ points with a detrend: the float32 reference's own float32 detrend leaves up to 4e-4 of max in the mean bin;
    # BASELINE.json's float32 bar is 1e-3)
    tol = TOL[dtype] if (dtype == "float64" or shape[1] < 2048) else 1e-3
    a = _cube(rng, shape, dtype)
    da, od = pair(a, D3, _coords3(shape, y0=2.5, x0=-1.0))
    # (the float32 reference detrends in float32 -- scipy -- which leaves 1e-4 of max in the mean bin of a 4096-point line; the
    # detrended float32 cases are held against the oracle on the same data in float64: DESIGN.md 2, deviations)
    od_det = od if dtype == "float64" else o.OArr(a.astype("float64"), D3, _coords3(shape, y0=2.5, x0=-1.0))
    worst = 0.0

    def on_fast():
        return "[fastm y-only]" in next(reversed(xa.api._plan_cache.values())).describe()

    for kw in (dict(), dict(detrend="linear", window="hann"), dict(detrend="constant", shift=False), dict(true_phase=False, true_amplitude=False, window="hamming")):
        worst = max(worst, check_values(xa.fft(da, dim=["y"], **kw), o.fft(od_det if "detrend" in kw else od, dim=["y"], **kw), tol))
        assert on_fast(), kw
    for kw in (dict(detrend="linear", window="hann"), dict(shift=False, scaling="spectrum", detrend="constant"), dict()):
        worst = max(worst, check_values(xa.power_spectrum(da, dim=["y"], **kw), o.power_spectrum(od_det if "detrend" in kw else od, dim=["y"], **kw), tol))
        assert on_fast(), kw
    # real_dim along the axis (ABI 0.1.4: the half output, no transposed copy; xrft.py:400-404, 673-682)
    for kw in (dict(detrend="linear", window="hann"), dict(scaling="spectrum", detrend="constant"), dict()):
        worst = max(worst, check_values(xa.power_spectrum(da, dim=["y"], real_dim="y", **kw), o.power_spectrum(od_det if "detrend" in kw else od, dim=["y"], real_dim="y", **kw), tol))
        assert on_fast(), kw
    worst = max(worst, check_values(xa.fft(da, dim=["y"], real_dim="y", detrend="linear"), o.fft(od_det, dim=["y"], real_dim="y", detrend="linear"), tol))
    assert on_fast()
    # complex input (the later stages of N-D transforms): one sequence per column
    cdt = "complex128" if dtype == "float64" else "complex64"
    z = (a + 1j * _cube(rng, shape, dtype)).astype(cdt)
    dz, oz = pair(z, D3, _coords3(shape, y0=2.5, x0=-1.0))
    oz_det = oz if dtype == "float64" else o.OArr(z.astype("complex128"), D3, _coords3(shape, y0=2.5, x0=-1.0))  # (as od_det)
    for kw in (dict(), dict(detrend="linear", window="hann", shift=False)):
        worst = max(worst, check_values(xa.fft(dz, dim=["y"], **kw), o.fft(oz_det if "detrend" in kw else oz, dim=["y"], **kw), tol))
        assert on_fast(), kw
    worst = max(worst, check_values(xa.power_spectrum(dz, dim=["y"], detrend="constant"), o.power_spectrum(oz_det, dim=["y"], detrend="constant"), tol))
    assert on_fast()
    # two fields: cross spectrum and cross phase along the axis (a column of each field = the two halves of one packed sequence)
    b = _cube(rng, shape, dtype)
    db, ob = pair(b, D3, _coords3(shape, y0=-1.5, x0=-1.0))
    ob_det = ob if dtype == "float64" else o.OArr(b.astype("float64"), D3, _coords3(shape, y0=-1.5, x0=-1.0))  # (as od_det)
    for kw in (dict(detrend="linear", window="hann"), dict(true_phase=False, shift=False, scaling="spectrum")):
        det = "detrend" in kw
        worst = max(worst, check_values(xa.cross_spectrum(da, db, dim=["y"], **kw), o.cross_spectrum(od_det if det else od, ob_det if det else ob, dim=["y"], **kw), tol))
        assert on_fast(), kw
    g = xa.cross_phase(da, db, dim=["y"], detrend="constant")
    r = o.cross_phase(od, ob, dim=["y"], detrend="constant")
    assert on_fast()
    dphi = np.abs(np.angle(np.exp(1j * (g.values - r.values))))
    mag = np.abs(o.cross_spectrum(od, ob, dim=["y"], detrend="constant").values)
    lim = 1e-10 if dtype == "float64" else 3e-4
    assert (dphi * mag).max() / mag.max() < lim, (dphi * mag).max() / mag.max()
    # the first axis of a 3-D array: batch = 1, inner = ny * nx
    worst = max(worst, check(xa.power_spectrum(da.transpose("y", "time", "x"), dim=["y"], detrend="linear", window="hann"),
                             o.power_spectrum(od_det.transpose("y", "time", "x"), dim=["y"], detrend="linear", window="hann"), tol))
    assert on_fast()
    return worst


def run_fastn_cases(shape=(2, 1215, 700), dtype="float32", cross=True):
    """csrc/fastn.h -- the y-first two-pass pipeline with the LENGTHS AS DATA (run-time radices 2 ... 20 incl. 7 / 11 / 13, odd lengths, ragged column blocks,
    the chirp convolution for a column length with a large prime factor) -- against the oracle, every mode that the table kernels of fastm.h take."""
    rng = np.random.default_rng(91)
    tol = TOL[dtype]
    a = _cube(rng, shape, dtype)
    da, od = pair(a, D3, _coords3(shape, y0=1.0, x0=-3.0))
    worst = 0.0

    def on_fast():
        return any("[fastn]" in p.describe() for p in xa.api._plan_cache.values())

    for kw in (dict(detrend="linear", window="hann"), dict(detrend="constant", shift=False, scaling="spectrum"), dict(window="hamming", window_correction=True), dict()):
        xa.api._plan_cache.clear()
        worst = max(worst, check(xa.power_spectrum(da, dim=["y", "x"], **kw), o.power_spectrum(od, dim=["y", "x"], **kw), tol))
        assert on_fast(), (shape, kw, [p.describe() for p in xa.api._plan_cache.values()])
    xa.api._plan_cache.clear()
    worst = max(worst, check(xa.power_spectrum(da, dim=["y"], real_dim="x", detrend="linear"), o.power_spectrum(od, dim=["y"], real_dim="x", detrend="linear"), tol))
    assert on_fast()
    for kw in (dict(detrend="linear", window="hann"), dict(true_phase=False, shift=False), dict(real_dim="x", detrend="constant")):
        xa.api._plan_cache.clear()
        dims = ["y"] if "real_dim" in kw else ["y", "x"]
        worst = max(worst, check(xa.fft(da, dim=dims, **kw), o.fft(od, dim=dims, **kw), tol))
        assert on_fast(), kw
    for kw in (dict(detrend="linear", window="hann"), dict(truncate=True)):
        xa.api._plan_cache.clear()
        r1 = xa.isotropic_power_spectrum(da, dim=["y", "x"], **kw)
        worst = max(worst, check(r1, o.isotropic_power_spectrum(od, dim=["y", "x"], **kw), tol))
        assert on_fast(), kw
        assert np.array_equal(np.asarray(xa.isotropic_power_spectrum(da, dim=["y", "x"], **kw).values), np.asarray(r1.values))  # bit-identical repeats
    if cross:
        b = _cube(rng, shape, dtype, trend=False)
        c2 = _coords3(shape, y0=2.5, x0=1.0)
        db, ob = pair(b, D3, c2)
        for kw in (dict(detrend="linear", window="hann"), dict(true_phase=False)):
            xa.api._plan_cache.clear()
            worst = max(worst, check(xa.cross_spectrum(da, db, dim=["y", "x"], **kw), o.cross_spectrum(od, ob, dim=["y", "x"], **kw), tol))
            assert on_fast(), kw
        xa.api._plan_cache.clear()
        db0, ob0 = pair(b, D3, _coords3(shape, y0=1.0, x0=-3.0))
        worst = max(worst, check(xa.isotropic_cross_spectrum(da, db0, dim=["y", "x"], window="hann"), o.isotropic_cross_spectrum(od, ob0, dim=["y", "x"], window="hann"), tol))
        assert on_fast()
    return worst


def run_inverse_one_pass_cases(shape=(2, 360, 250), dtype="float64"):
    """xrft.ifft (xrft.py:479-646) on the one-pass kernels of csrc/fastg.h: a two-axis inverse transform as two one-axis stages (y where it lies, then the rows),
    one axis along a first / middle axis with no transposed copy, along the contiguous axis, small slabs in one pass -- every true_phase / shift combination,
    against the oracle."""
    import warnings

    rng = np.random.default_rng(79)
    tol = TOL[dtype]
    a = _cube(rng, shape, dtype)
    da, od = pair(a, D3, _coords3(shape, y0=1.0, x0=-3.0))
    worst = 0.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for kw in (dict(), dict(true_phase=False, shift=False), dict(true_phase=False), dict(shift=False)):
            F, Fo = xa.fft(da, dim=["y", "x"], **kw), o.fft(od, dim=["y", "x"], **kw)
            xa.api._plan_cache.clear()
            worst = max(worst, check_values(xa.ifft(F, dim=["freq_y", "freq_x"], **kw), o.ifft(Fo, dim=["freq_y", "freq_x"], **kw), tol))
            tags = [p.describe() for p in xa.api._plan_cache.values()]
            assert (any("[fastg y-only]" in t or "[fastm y-only]" in t for t in tags) and any("[fastg rows]" in t or "[fastm x-only]" in t or "[fasty complex rows]" in t for t in tags)
                    or any("[fastg] one pass" in t for t in tags) or any("[fasty complex]" in t for t in tags)), tags  # (complex64 power-of-two slabs: csrc/fasty_c2c.h)
            for d, tag in (("y", ("[fastg y-only]", "[fastm y-only]")), ("x", ("[fastg rows]", "[fastm x-only]", "[fasty complex rows]"))):  # (rows of a table length: the table kernel takes the inverse, too -- round 5)
                F1, F1o = xa.fft(da, dim=[d], **kw), o.fft(od, dim=[d], **kw)
                xa.api._plan_cache.clear()
                worst = max(worst, check_values(xa.ifft(F1, dim=["freq_" + d], **kw), o.ifft(F1o, dim=["freq_" + d], **kw), tol))
                assert any(t_ in p.describe() for p in xa.api._plan_cache.values() for t_ in tag), (d, kw)
            if shape[2] % 2 == 0:  # the half spectrum back to real samples (irfft / irfftn, real_dim): along the rows, and over two axes where the slab fits a workgroup
                Fr, Fro = xa.fft(da, dim=["x"], real_dim="x", **kw), o.fft(od, dim=["x"], real_dim="x", **kw)
                xa.api._plan_cache.clear()
                worst = max(worst, check_values(xa.ifft(Fr, dim=["freq_x"], real_dim="freq_x", **kw), o.ifft(Fro, dim=["freq_x"], real_dim="freq_x", **kw), tol))
                # (float32 rows of 512 .. 4096 samples: the c2r row pass; a table length: two half rows per transform in fastm_xonly_kernel)
                assert any("[fastg rows]" in p.describe() or "[fasty complex rows]" in p.describe() or "[fastm x-only]" in p.describe() for p in xa.api._plan_cache.values()), kw
                if shape[1] == 1440 and shape[2] == 720:  # (the C5 grid back from its half spectra, irfftn: both stages on the table kernels -- 361 complex columns, the last block short)
                    Fn, Fno = xa.fft(da, dim=["y", "x"], real_dim="x", **kw), o.fft(od, dim=["y", "x"], real_dim="x", **kw)
                    xa.api._plan_cache.clear()
                    worst = max(worst, check_values(xa.ifft(Fn, dim=["freq_y", "freq_x"], real_dim="freq_x", **kw), o.ifft(Fno, dim=["freq_y", "freq_x"], real_dim="freq_x", **kw), tol))
                    tags = [p.describe() for p in xa.api._plan_cache.values()]
                    assert any("[fastm y-only]" in t for t in tags) and any("[fastm x-only]" in t for t in tags), tags
                Fr2, Fr2o = xa.fft(da, dim=["y"], real_dim="x", **kw), o.fft(od, dim=["y"], real_dim="x", **kw)
                worst = max(worst, check_values(xa.ifft(Fr2, dim=["freq_y"], real_dim="freq_x", **kw), o.ifft(Fr2o, dim=["freq_y"], real_dim="freq_x", **kw), tol))
    return worst


def run_complex_rows_cases(n, nt=3, seed=91):
    """Complex64 rows of n = 2048 .. 16384 points along the contiguous axis in ONE pass (csrc/fastr.h, fastc_kernel) and of 2^16 .. 2^20 points in the two passes of
    csrc/fasty_c2c.h on the [n / 256][256] view (the four-step form): xrft.fft of complex data, xrft.ifft of its spectrum (true phase on / off, shift on / off, explicit
    lag), power spectrum with a window (the long rows: without -- a window takes them to the generic passes) -- against the oracle (xrft.py:439-447, :586-621)."""
    rng = np.random.default_rng(seed + n)
    tol = TOL["complex64"]
    z = (rng.standard_normal((nt, n)) + 1j * rng.standard_normal((nt, n))).astype(np.complex64)
    c = {"t": np.arange(nt), "x": np.arange(n) * 0.5 + 3.0}
    da, od = pair(z, ("t", "x"), c)
    worst = 0.0

    def tag():
        return next(reversed(xa.api._plan_cache.values())).describe()

    for kw in (dict(), dict(true_phase=False, true_amplitude=False), dict(shift=False), dict(true_phase=False, shift=False)):
        F, Fo = xa.fft(da, dim="x", **kw), o.fft(od, dim="x", **kw)
        assert "complex rows" in tag(), tag()
        worst = max(worst, check(F, Fo, tol))
        ikw = dict(kw)
        if kw.get("true_phase", True):
            ikw["lag"] = float(c["x"][n // 2])  # (the lag that makes ifft(fft(z)) the round trip: xrft.py:215-234)
        G, Go = xa.ifft(F, dim="freq_x", **ikw), o.ifft(Fo, dim="freq_x", **ikw)
        assert "complex rows" in tag() and "inverse" in tag(), tag()
        worst = max(worst, check_values(G, Go, tol))
        if not kw:  # (the round trip of the default call: true phase, shifted spectrum)
            assert np.abs(np.asarray(G.values) - z).max() < 100 * tol * np.abs(z).max()
    P, Po = xa.power_spectrum(da, dim="x", window="hann"), o.power_spectrum(od, dim="x", window="hann")
    assert "complex rows" in tag() or n >= 65536, tag()
    worst = max(worst, check(P, Po, tol))
    if n >= 65536:
        P, Po = xa.power_spectrum(da, dim="x"), o.power_spectrum(od, dim="x")
        assert "complex rows, four-step" in tag(), tag()
        worst = max(worst, check(P, Po, tol))
    return worst


def run_complex_two_pass_cases(ny, nx, nt=2, variant=0, seed=97):
    """Complex64 slabs, both lengths a power of two 256 .. 4096, through the two-pass pipeline of csrc/fasty_c2c.h: xrft.fft of complex data over (y, x) with and
    without a window, xrft.ifft of its spectrum (the fftshifted input rotated on load, the lag's phase on the input, conjugate in / out), power spectrum --
    against the oracle (xrft.py:439-447, :586-621), and as a round trip."""
    rng = np.random.default_rng(seed + ny + 3 * nx)
    tol = TOL["complex64"]
    kw = (dict(), dict(true_phase=False, true_amplitude=False), dict(shift=False), dict(true_phase=False, shift=False))[variant % 4]
    z = (rng.standard_normal((nt, ny, nx)) + 1j * rng.standard_normal((nt, ny, nx))).astype(np.complex64)
    c = {"t": np.arange(nt), "y": np.arange(ny) * 0.25 - 7.0, "x": np.arange(nx) * 0.5 + 3.0}
    da, od = pair(z, ("t", "y", "x"), c)

    def tag():
        return next(reversed(xa.api._plan_cache.values())).describe()

    F, Fo = xa.fft(da, dim=["y", "x"], **kw), o.fft(od, dim=["y", "x"], **kw)
    assert "[fasty complex]" in tag(), tag()
    worst = check(F, Fo, tol)
    Fw, Fwo = xa.fft(da, dim=["y", "x"], window="hann", **kw), o.fft(od, dim=["y", "x"], window="hann", **kw)
    worst = max(worst, check(Fw, Fwo, tol))
    ikw = dict(kw)
    if kw.get("true_phase", True):
        ikw["lag"] = [float(c["y"][ny // 2]), float(c["x"][nx // 2])]
    G, Go = xa.ifft(F, dim=["freq_y", "freq_x"], **ikw), o.ifft(Fo, dim=["freq_y", "freq_x"], **ikw)
    assert "[fasty complex]" in tag() and "inverse" in tag(), tag()
    worst = max(worst, check_values(G, Go, tol))
    if not kw:  # (the round trip of the default call: true phase, shifted spectrum)
        assert np.abs(np.asarray(G.values) - z).max() < 100 * tol * np.abs(z).max()
    P, Po = xa.power_spectrum(da, dim=["y", "x"]), o.power_spectrum(od, dim=["y", "x"])
    assert "[fasty complex]" in tag(), tag()
    worst = max(worst, check(P, Po, tol))
    return worst


def run_c2r_two_pass_cases(ny, nx, nt=2, variant=0, seed=101):
    """xrft.ifft with real_dim (irfftn / irfft, xrft.py:612-616) of float32 half spectra whose lengths are powers of two: the complex two-pass pipeline with the c2r
    row pass (csrc/fasty_c2c.h fastyc_rows_c2r_kernel: ny = 256 .. 4096, nx = 512 .. 4096), and that row pass alone along the contiguous axis -- against the oracle
    and as a round trip of the real field."""
    rng = np.random.default_rng(seed + ny + 5 * nx)
    tol = TOL["float32"]
    kw = (dict(), dict(true_phase=False, true_amplitude=False), dict(shift=False), dict(true_phase=False, shift=False))[variant % 4]
    v = rng.standard_normal((nt, ny, nx)).astype(np.float32)
    c = {"t": np.arange(nt), "y": np.arange(ny) * 0.25 - 7.0, "x": np.arange(nx) * 0.5 + 3.0}
    da, od = pair(v, ("t", "y", "x"), c)

    def tag():
        return next(reversed(xa.api._plan_cache.values())).describe()

    F, Fo = xa.fft(da, dim=["y"], real_dim="x", **kw), o.fft(od, dim=["y"], real_dim="x", **kw)
    G, Go = xa.ifft(F, dim=["freq_y"], real_dim="freq_x", **kw), o.ifft(Fo, dim=["freq_y"], real_dim="freq_x", **kw)
    assert "[fasty complex]" in tag() and "real samples" in tag(), tag()
    assert np.asarray(G.values).dtype == np.float32
    worst = check_values(G, Go, tol)
    F1, F1o = xa.fft(da, dim="x", real_dim="x", **kw), o.fft(od, dim="x", real_dim="x", **kw)
    G1, G1o = xa.ifft(F1, dim="freq_x", real_dim="freq_x", **kw), o.ifft(F1o, dim="freq_x", real_dim="freq_x", **kw)
    assert "[fasty complex rows]" in tag() and "real samples" in tag(), tag()
    worst = max(worst, check_values(G1, G1o, tol))
    if not kw:  # the default call returns the field it started from
        assert np.abs(np.asarray(G.values) - v).max() < 100 * tol * np.abs(v).max() and np.abs(np.asarray(G1.values) - v).max() < 100 * tol * np.abs(v).max()
    return worst


def run_rows_any_length_cases(shape=(37, 250), dtype="float32"):
    """One transform axis, the contiguous one, on ANY smooth length outside the tables (96, 250, 750, 1250, 6000, odd 125 / 243 / 729): csrc/fastg.h on groups of
    rows (run-time radices, a mean / line per row in the workgroup; the last group short) against the oracle -- fft with true phase, real_dim (half
    output), power spectrum, every detrend, window, shift."""
    rng = np.random.default_rng(73)
    tol = TOL[dtype]
    n = shape[-1]
    v = (rng.standard_normal(shape) + 2.0 + 3.0 * np.arange(n) / n).astype(dtype)
    dims = ("a", "b", "x")[-len(shape):]
    c = {d: np.arange(s) for d, s in zip(dims[:-1], shape[:-1])}
    c["x"] = np.arange(n) * 0.5 - 7.0
    da, od = pair(v, dims, c)
    worst = 0.0

    def on_fast():
        return "[fastg rows]" in next(reversed(xa.api._plan_cache.values())).describe()

    for kw in (dict(), dict(detrend="linear", window="hann"), dict(detrend="constant", shift=False), dict(true_phase=False, true_amplitude=False, window="hamming"),
               dict(real_dim="x", detrend="linear")):
        worst = max(worst, check(xa.fft(da, dim=["x"], **kw), o.fft(od, dim=["x"], **kw), tol))
        assert on_fast(), kw
    for kw in (dict(detrend="linear", window="hann"), dict(shift=False, scaling="spectrum", detrend="constant"), dict(real_dim="x", window="hann"), dict()):
        worst = max(worst, check(xa.power_spectrum(da, dim=["x"], **kw), o.power_spectrum(od, dim=["x"], **kw), tol))
        assert on_fast(), kw
    # two fields: the cross spectrum along the axis (both fields' rows in the workgroup's LDS)
    w = (rng.standard_normal(shape) - 1.0).astype(dtype)
    c2 = dict(c); c2["x"] = c["x"] + 1.25
    db, ob = pair(w, dims, c2)
    for kw in (dict(detrend="linear", window="hann"), dict(true_phase=False, shift=False), dict(real_dim="x", detrend="constant")):
        worst = max(worst, check(xa.cross_spectrum(da, db, dim=["x"], **kw), o.cross_spectrum(od, ob, dim=["x"], **kw), tol))
        assert on_fast(), kw
    return worst


def run_rows_rader_cases(shape=(37, 365), dtype="float32"):
    """One transform axis, the contiguous one, on a length with ONE prime factor 17 ... 127 (365 = 5 x 73 daily samples of (station, time) rows, 730, 1460, 366): the
    prime-factor / Rader form of csrc/fastg.h's fastgy_kernel with the lanes along the samples (FORM 3) against the oracle -- fft (true phase, ifftshift), power spectrum,
    every detrend, window, shift; an odd number of rows (the last sequence holds one row); complex rows; xrft.ifft; the cross spectrum and cross phase of two fields.
    real_dim (half output) included."""
    import warnings

    rng = np.random.default_rng(79)
    tol = TOL[dtype]
    n = shape[-1]
    v = (rng.standard_normal(shape) + 2.0 + 3.0 * np.arange(n) / n).astype(dtype)
    dims = ("a", "b", "x")[-len(shape):]
    c = {d: np.arange(s) for d, s in zip(dims[:-1], shape[:-1])}
    c["x"] = np.arange(n) * 0.5 - 7.0
    da, od = pair(v, dims, c)
    od_det = od if dtype == "float64" else o.OArr(v.astype("float64"), dims, c)
    worst = 0.0

    def on_fast():
        return "[fastg rows Rader]" in next(reversed(xa.api._plan_cache.values())).describe()

    for kw in (dict(), dict(detrend="linear", window="hann"), dict(detrend="constant", shift=False), dict(true_phase=False, true_amplitude=False, window="hamming")):
        worst = max(worst, check_values(xa.fft(da, dim=["x"], **kw), o.fft(od_det if "detrend" in kw else od, dim=["x"], **kw), tol))
        assert on_fast(), kw
    for kw in (dict(detrend="linear", window="hann"), dict(shift=False, scaling="spectrum", detrend="constant"), dict()):
        worst = max(worst, check_values(xa.power_spectrum(da, dim=["x"], **kw), o.power_spectrum(od_det if "detrend" in kw else od, dim=["x"], **kw), tol))
        assert on_fast(), kw
    worst = max(worst, check(xa.power_spectrum(da, dim=["x"], real_dim="x", window="hann"), o.power_spectrum(od, dim=["x"], real_dim="x", window="hann"), tol))
    assert on_fast()  # (the half output, ABI 0.1.4)
    worst = max(worst, check_values(xa.fft(da, dim=["x"], real_dim="x", detrend="linear"), o.fft(od_det, dim=["x"], real_dim="x", detrend="linear"), tol))
    assert on_fast()
    # complex rows, and back
    cdt = "complex128" if dtype == "float64" else "complex64"
    z = (v + 1j * rng.standard_normal(shape)).astype(cdt)
    dz, oz = pair(z, dims, c)
    oz_det = oz if dtype == "float64" else o.OArr(z.astype("complex128"), dims, c)
    for kw in (dict(), dict(detrend="linear", window="hann", shift=False)):
        worst = max(worst, check_values(xa.fft(dz, dim=["x"], **kw), o.fft(oz_det if "detrend" in kw else oz, dim=["x"], **kw), tol))
        assert on_fast(), kw
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for kw in (dict(), dict(true_phase=False, shift=False), dict(true_phase=False), dict(shift=False)):
            F, Fo = xa.fft(da, dim=["x"], **kw), o.fft(od, dim=["x"], **kw)
            worst = max(worst, check_values(xa.ifft(F, dim=["freq_x"], **kw), o.ifft(Fo, dim=["freq_x"], **kw), tol))
            assert on_fast(), kw
    # two fields
    w = (rng.standard_normal(shape) - 1.0).astype(dtype)
    c2 = dict(c); c2["x"] = c["x"] + 1.25
    db, ob = pair(w, dims, c2)
    ob_det = ob if dtype == "float64" else o.OArr(w.astype("float64"), dims, c2)
    for kw in (dict(detrend="linear", window="hann"), dict(true_phase=False, shift=False)):
        det = "detrend" in kw
        worst = max(worst, check_values(xa.cross_spectrum(da, db, dim=["x"], **kw), o.cross_spectrum(od_det if det else od, ob_det if det else ob, dim=["x"], **kw), tol))
        assert on_fast(), kw
    ph, pho = xa.cross_phase(da, db, dim=["x"]), o.cross_phase(od, ob, dim=["x"])
    assert on_fast()
    dphi = np.angle(np.exp(1j * (np.asarray(ph.values, dtype="float64") - np.asarray(pho.values, dtype="float64"))))
    amp = np.abs(np.asarray(o.cross_spectrum(od, ob, dim=["x"], true_phase=True).values))
    assert np.abs(dphi[amp > 1e-3 * amp.max()]).max() < (1e-9 if dtype == "float64" else 2e-3)
    return worst


def run_xonly_fast_cases(shape=(5, 360), dtype="float64"):
    """One short transform axis, the contiguous one (spectra along the last axis of (..., n) arrays, n in the fastm table):
    csrc/fastm.h fastm_xonly_kernel (rows packed in pairs; an odd number of rows leaves the last pair half empty) against the
    oracle -- fft with true phase, real_dim (half output), power spectrum, every detrend, window, shift."""
    rng = np.random.default_rng(71)
    tol = TOL[dtype] if (dtype == "float64" or shape[-1] < 2048) else 1e-3  # (see run_yonly_fast_cases)
    n = shape[-1]
    v = (rng.standard_normal(shape) + 2.0 + 3.0 * np.arange(n) / n).astype(dtype)
    dims = ("a", "b", "x")[-len(shape):]
    c = {d: np.arange(s) for d, s in zip(dims[:-1], shape[:-1])}
    c["x"] = np.arange(n) * 0.5 - 7.0
    da, od = pair(v, dims, c)
    worst = 0.0

    def on_fast():  # (real float32 rows of 4096 samples: the register-resident one-pass kernel, csrc/fastr.h; else rows packed in pairs)
        d = next(reversed(xa.api._plan_cache.values())).describe()
        # (round 6: complex64 rows of 256 .. 4096 points without a detrend: the row pass of csrc/fasty_c2c.h on the input's own rows)
        return "[fastm x-only]" in d or (n == 4096 and dtype == "float32" and "[fastr]" in d) or (dtype == "float32" and "[fasty complex rows]" in d)

    for kw in (dict(), dict(detrend="linear", window="hann"), dict(detrend="constant", shift=False), dict(true_phase=False, true_amplitude=False, window="hamming"),
               dict(real_dim="x", detrend="linear")):
        worst = max(worst, check(xa.fft(da, dim=["x"], **kw), o.fft(od, dim=["x"], **kw), tol))
        assert on_fast(), kw
    for kw in (dict(detrend="linear", window="hann"), dict(shift=False, scaling="spectrum", detrend="constant"), dict(real_dim="x", window="hann"), dict()):
        worst = max(worst, check(xa.power_spectrum(da, dim=["x"], **kw), o.power_spectrum(od, dim=["x"], **kw), tol))
        assert on_fast(), kw
    cdt = "complex128" if dtype == "float64" else "complex64"
    z = (v + 1j * rng.standard_normal(shape)).astype(cdt)
    dz, oz = pair(z, dims, c)
    for kw in (dict(), dict(detrend="linear", window="hann", shift=False)):
        worst = max(worst, check(xa.fft(dz, dim=["x"], **kw), o.fft(oz, dim=["x"], **kw), tol))
        assert on_fast(), kw
    worst = max(worst, check(xa.power_spectrum(dz, dim=["x"], detrend="constant"), o.power_spectrum(oz, dim=["x"], detrend="constant"), tol))
    assert on_fast()
    w = (rng.standard_normal(shape) - 1.0).astype(dtype)
    c2 = dict(c); c2["x"] = c["x"] + 1.25
    db, ob = pair(w, dims, c2)
    for kw in (dict(detrend="linear", window="hann"), dict(true_phase=False, shift=False), dict(real_dim="x", detrend="constant")):
        worst = max(worst, check(xa.cross_spectrum(da, db, dim=["x"], **kw), o.cross_spectrum(od, ob, dim=["x"], **kw), tol))
        assert on_fast(), kw
    g = xa.cross_phase(da, db, dim=["x"], window="hann")
    r = o.cross_phase(od, ob, dim=["x"], window="hann")
    assert on_fast()
    dphi = np.abs(np.angle(np.exp(1j * (g.values - r.values))))
    mag = np.abs(o.cross_spectrum(od, ob, dim=["x"], window="hann").values)
    lim = 1e-10 if dtype == "float64" else 3e-4
    assert (dphi * mag).max() / mag.max() < lim, (dphi * mag).max() / mag.max()
    return worst


def run_long_prime_cases(lengths=((9001, "float64"), (10007, "float32"), (9001, "complex128"))):
    """One transform axis (the last) whose length has a prime factor above 128 and is too long for Bluestein inside one LDS tile:
    Bluestein through global memory (api._bluestein_1d: xrfthip_table_mul around two smooth-length plans).  numpy.fft takes any
    length (xrft.py:398-447): fft with every option, power spectrum, real_dim, a descending coordinate."""
    rng = np.random.default_rng(81)
    worst = 0.0
    for n, dt in lengths:
        v = rng.standard_normal((3, n)) + 0.5 + 2.0 * np.arange(n) / n
        if dt.startswith("complex"):
            v = v + 1j * rng.standard_normal((3, n))
        v = v.astype(dt)
        c = {"t": np.arange(3), "x": np.arange(n) * 0.5 - 11.0}
        da, od = pair(v, ("t", "x"), c)
        tol = TOL[dt]
        # float32 Bluestein through global memory: three length-32768 transforms and two chirp products per result, which in float32 held
        # the bins 1e-3 of the peak to 4e-3 only (round 3); float32 data now take this path in float64 (api._bluestein_1d): BIN_REL
        br = BIN_REL
        for kw in (dict(), dict(detrend="linear", window="hann"), dict(shift=False, true_phase=False)):
            worst = max(worst, check(xa.fft(da, dim=["x"], **kw), o.fft(od, dim=["x"], **kw), tol, br))
        worst = max(worst, check(xa.power_spectrum(da, dim=["x"], detrend="constant", window="hann"),
                                 o.power_spectrum(od, dim=["x"], detrend="constant", window="hann"), tol, br))
        if not dt.startswith("complex"):
            worst = max(worst, check(xa.power_spectrum(da, dim=["x"], real_dim="x"), o.power_spectrum(od, dim=["x"], real_dim="x"), tol, br))
            worst = max(worst, check(xa.fft(da, dim=["x"], real_dim="x", detrend="linear"), o.fft(od, dim=["x"], real_dim="x", detrend="linear"), tol, br))
        c2 = dict(c)
        c2["x"] = c["x"][::-1].copy()
        da2, od2 = pair(v, ("t", "x"), c2)
        worst = max(worst, check(xa.fft(da2, dim=["x"], window="hann"), o.fft(od2, dim=["x"], window="hann"), tol, br))
        for kw in (dict(), dict(true_phase=False, shift=False)):  # and back: the inverse transform with conjugated chirps
            F, Fo = xa.fft(da, dim=["x"], **kw), o.fft(od, dim=["x"], **kw)
            worst = max(worst, check(xa.ifft(F, dim=["freq_x"], **kw), o.ifft(Fo, dim=["freq_x"], **kw), tol, br))
        if dt.startswith("complex"):
            continue
        # the half spectrum back to 2 (n // 2) real samples (irfft; 2 x 5003 points for n = 10007: Bluestein again)
        F, Fo = xa.fft(da, dim=["x"], real_dim="x"), o.fft(od, dim=["x"], real_dim="x")
        worst = max(worst, check(xa.ifft(F, dim=["freq_x"], real_dim="freq_x"), o.ifft(Fo, dim=["freq_x"], real_dim="freq_x"), tol, br))
        # two transform axes, one of them such a length (either position): no two-axis plan exists, the axes go one at a time
        tol2 = max(tol, 1e-8)  # (the cube's trend grows with the index: 10^3 times the noise at 10^5 samples -- the plane fit's own rounding)
        for shape in ((2, 6, n), (2, n, 6)):
            a = _cube(rng, shape, dt)
            da3, od3 = pair(a, D3, _coords3(shape))
            for kw in (dict(), dict(detrend="linear", window="hann")):
                worst = max(worst, check(xa.fft(da3, dim=["y", "x"], **kw), o.fft(od3, dim=["y", "x"], **kw), tol2 if kw else tol, br))
            worst = max(worst, check(xa.power_spectrum(da3, dim=["y", "x"], detrend="linear", window="hann"),
                                     o.power_spectrum(od3, dim=["y", "x"], detrend="linear", window="hann"), tol2, br))
            worst = max(worst, check(xa.power_spectrum(da3, dim=["y"], real_dim="x", detrend="constant"),
                                     o.power_spectrum(od3, dim=["y"], real_dim="x", detrend="constant"), tol, br))
            b = _cube(rng, shape, dt)
            db3, ob3 = pair(b, D3, _coords3(shape, y0=1.0))
            worst = max(worst, check(xa.cross_spectrum(da3, db3, dim=["y", "x"], window="hann"), o.cross_spectrum(od3, ob3, dim=["y", "x"], window="hann"), tol, br))
            # ... and the calls ADVICE r2 found raising: the inverse over both axes, and the cross phase (the angle of the composed cross spectrum)
            F3, Fo3 = xa.fft(da3, dim=["y", "x"]), o.fft(od3, dim=["y", "x"])
            worst = max(worst, check(xa.ifft(F3, dim=["freq_y", "freq_x"]), o.ifft(Fo3, dim=["freq_y", "freq_x"]), tol, br))
            gph = xa.cross_phase(da3, db3, dim=["y", "x"], window="hann")
            rcs = o.cross_spectrum(od3, ob3, dim=["y", "x"], window="hann")
            dphi = np.abs(np.angle(np.exp(1j * (np.asarray(gph.values, dtype=np.float64) - np.angle(rcs.values)))))
            assert (dphi * np.abs(rcs.values)).max() / np.abs(rcs.values).max() < (1e-9 if dt == "float64" else 3e-4)
            # isotropic spectra: the full spectrum, then isotropize -- what the reference does literally (xrft.py:1085-1095)
            kwi = dict(dim=["y", "x"], detrend="linear", window="hann", truncate=True)
            worst = max(worst, check(xa.isotropic_power_spectrum(da3, **kwi), o.isotropic_power_spectrum(od3, **kwi), tol2, br))
            worst = max(worst, check(xa.isotropic_cross_spectrum(da3, db3, dim=["y", "x"], window="hann"),
                                     o.isotropic_cross_spectrum(od3, ob3, dim=["y", "x"], window="hann"), tol2, br))
    return worst


def run_reduce_axis_cases():
    """xrfthip_reduce_axis (the sum / mean over a batch dimension as ONE library kernel: float64 accumulation in index order) against
    numpy, every dtype, first / middle / last axis, bit-identical repeats; DataArray.mean / .sum of device data go through it."""
    import torch

    from xrft_amd import engine

    rng = np.random.default_rng(91)
    dev = xa.api._to_device(np.zeros(1, dtype=np.float32)).device
    for dt in ("float32", "float64", "complex64", "complex128"):
        v = rng.standard_normal((5, 7, 33))
        if dt.startswith("complex"):
            v = v + 1j * rng.standard_normal(v.shape)
        v = v.astype(dt)
        t = torch.from_numpy(v).to(dev)
        for ax in (0, 1, 2):
            got = engine.reduce_axis(t, ax, 1.0 / v.shape[ax])
            assert torch.equal(got, engine.reduce_axis(t, ax, 1.0 / v.shape[ax]))
            ref = v.astype("complex128" if dt.startswith("complex") else "float64").mean(axis=ax)
            npt.assert_allclose(got.cpu().numpy(), ref, rtol=3e-6 if dt in ("float32", "complex64") else 1e-13, atol=1e-6 if dt in ("float32", "complex64") else 1e-14)
    da = xa.DataArray(torch.from_numpy(rng.standard_normal((6, 9))).to(dev), ("time", "freq_r"), {"time": np.arange(6), "freq_r": np.arange(9) * 0.1})
    m = da.mean("time")
    assert m.dims == ("freq_r",) and np.array_equal(m["freq_r"].values, np.arange(9) * 0.1)
    npt.assert_allclose(m.values, da.values.mean(axis=0), rtol=1e-13)
    npt.assert_allclose(da.sum("freq_r").values, da.values.sum(axis=1), rtol=1e-13)


def run_mid_layout_cases(dtype="float64", shape=(24, 5, 20)):
    """Two transform axes that are NOT adjacent -- dim = ["t", "x"] of a (t, y, x) array, (a, t, y, x, i) with dims in front of, between and behind them --
    where they lie (xrfthip_desc.mid: [batch][n0][mid][n1][inner], a plane detrend per (batch, mid, inner) element first; no transposed copies) against
    the oracle.  xrft.py:395-409."""
    rng = np.random.default_rng(98)
    tol = TOL[dtype]
    nt, ny, nx = shape
    ii, jj = np.meshgrid(np.arange(nt), np.arange(nx), indexing="ij")
    v = (rng.standard_normal(shape) + (0.05 * ii - 0.03 * jj + 2.0)[:, None, :] * (1.0 + np.arange(ny))[None, :, None]).astype(dtype)
    c = {"t": np.arange(nt) * 0.5 + 1.0, "y": np.arange(ny), "x": np.arange(nx) * 2.0 - 3.0}
    da, od = pair(v, ("t", "y", "x"), c)
    worst = 0.0

    def on_inner():
        d_ = next(reversed(xa.api._plan_cache.values())).describe()
        return "[inner layout]" in d_ and "[mid %d]" % ny in d_

    for kw in (dict(), dict(detrend="linear", window="hann"), dict(detrend="constant", shift=False, true_phase=False), dict(window="hamming", true_amplitude=False)):
        worst = max(worst, check(xa.fft(da, dim=["t", "x"], **kw), o.fft(od, dim=["t", "x"], **kw), tol))
        assert on_inner(), kw
    for kw in (dict(detrend="linear", window="hann"), dict(scaling="spectrum", shift=False), dict(detrend="constant", window="hann", window_correction=True)):
        worst = max(worst, check(xa.power_spectrum(da, dim=["t", "x"], **kw), o.power_spectrum(od, dim=["t", "x"], **kw), tol))
        assert on_inner(), kw
    worst = max(worst, check(xa.fft(da, dim=["x", "t"], detrend="linear", window="hann"), o.fft(od, dim=["x", "t"], detrend="linear", window="hann"), tol))
    assert on_inner()
    c2 = dict(c); c2["t"] = c["t"][::-1].copy()  # a descending coordinate (flip)
    da2, od2 = pair(v, ("t", "y", "x"), c2)
    worst = max(worst, check(xa.fft(da2, dim=["t", "x"], window="hann"), o.fft(od2, dim=["t", "x"], window="hann"), tol))
    z = (v + 1j * rng.standard_normal(shape)).astype("complex128" if dtype == "float64" else "complex64")
    dz, oz = pair(z, ("t", "y", "x"), c)
    worst = max(worst, check(xa.fft(dz, dim=["t", "x"], detrend="constant"), o.fft(oz, dim=["t", "x"], detrend="constant"), tol))
    assert on_inner()
    # dims in front of, between and behind the two axes
    w = rng.standard_normal((2, 12, 3, 10, 4)).astype(dtype)
    c5 = {"a": np.arange(2), "t": np.arange(12) * 1.0, "y": np.arange(3), "x": np.arange(10) * 0.25, "i": np.arange(4)}
    d5, o5 = pair(w, ("a", "t", "y", "x", "i"), c5)
    worst = max(worst, check(xa.power_spectrum(d5, dim=["t", "x"], detrend="linear", window="hann"), o.power_spectrum(o5, dim=["t", "x"], detrend="linear", window="hann"), tol))
    d_ = next(reversed(xa.api._plan_cache.values())).describe()
    assert "[batch 2][ny 12][mid 3][nx 10][inner 4]" in d_, d_
    return worst


def run_inner_layout_cases(dtype="float64", shape=(24, 20, 6)):
    """Two ADJACENT transform axes that are not the trailing ones -- dim = ["y", "x"] of a (y, x, t) array, (t, y, x, z), the two in
    either order -- through the engine's inner layout (xrfthip_desc.inner: x where it lies, then y, a detrend pass first; no
    transposed copies) against the oracle; and xrft.detrend over such axes (xrfthip_detrend_inner).  xrft.py:395-409."""
    rng = np.random.default_rng(97)
    tol = TOL[dtype]
    ny, nx, nt = shape
    ii, jj = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    v = (rng.standard_normal(shape) + (0.05 * ii - 0.03 * jj + 2.0)[:, :, None] * (1.0 + np.arange(nt))[None, None, :]).astype(dtype)
    c = {"y": np.arange(ny) * 0.5 + 1.0, "x": np.arange(nx) * 2.0 - 3.0, "t": np.arange(nt)}
    da, od = pair(v, ("y", "x", "t"), c)
    worst = 0.0

    def on_inner():
        return "[inner layout]" in next(reversed(xa.api._plan_cache.values())).describe()

    for kw in (dict(), dict(detrend="linear", window="hann"), dict(detrend="constant", shift=False, true_phase=False), dict(window="hamming", true_amplitude=False)):
        worst = max(worst, check(xa.fft(da, dim=["y", "x"], **kw), o.fft(od, dim=["y", "x"], **kw), tol))
        assert on_inner(), kw
    for kw in (dict(detrend="linear", window="hann"), dict(scaling="spectrum", shift=False), dict(detrend="constant", window="hann", window_correction=True)):
        worst = max(worst, check(xa.power_spectrum(da, dim=["y", "x"], **kw), o.power_spectrum(od, dim=["y", "x"], **kw), tol))
        assert on_inner(), kw
    # the same axes named in the other order: the plan's first axis is the one that comes first in memory
    worst = max(worst, check(xa.fft(da, dim=["x", "y"], detrend="linear", window="hann"), o.fft(od, dim=["x", "y"], detrend="linear", window="hann"), tol))
    assert on_inner()
    worst = max(worst, check(xa.power_spectrum(da, dim=["x", "y"], window="hann"), o.power_spectrum(od, dim=["x", "y"], window="hann"), tol))
    # a descending coordinate (flip) and complex input
    c2 = dict(c); c2["x"] = c["x"][::-1].copy()
    da2, od2 = pair(v, ("y", "x", "t"), c2)
    worst = max(worst, check(xa.fft(da2, dim=["y", "x"], window="hann"), o.fft(od2, dim=["y", "x"], window="hann"), tol))
    assert on_inner()
    z = (v + 1j * rng.standard_normal(shape)).astype("complex128" if dtype == "float64" else "complex64")
    dz, oz = pair(z, ("y", "x", "t"), c)
    worst = max(worst, check(xa.fft(dz, dim=["y", "x"], detrend="constant"), o.fft(oz, dim=["y", "x"], detrend="constant"), tol))
    assert on_inner()
    # leading AND trailing dims
    w = rng.standard_normal((3, 12, 10, 4)).astype(dtype)
    c4 = {"t": np.arange(3), "y": np.arange(12) * 1.0, "x": np.arange(10) * 0.25, "z": np.arange(4)}
    d4, o4 = pair(w, ("t", "y", "x", "z"), c4)
    worst = max(worst, check(xa.power_spectrum(d4, dim=["y", "x"], detrend="linear", window="hann"), o.power_spectrum(o4, dim=["y", "x"], detrend="linear", window="hann"), tol))
    assert on_inner()
    # stand-alone detrend where the axes lie
    for det in ("constant", "linear"):
        worst = max(worst, check(xa.detrend(da, ["y", "x"], det), o.detrend(od, ["y", "x"], det).transpose("y", "x", "t"), tol))
        worst = max(worst, check(xa.detrend(d4, ["y", "x"], det), o.detrend(o4, ["y", "x"], det).transpose("t", "y", "x", "z"), tol))
        worst = max(worst, check(xa.detrend(d4, "y", det), o.detrend(o4, "y", det).transpose("t", "y", "x", "z"), tol))
    # many independent elements (the per-element coefficients no longer fit the LDS of the pass that subtracts the planes), rows of a
    # length the 256-element stride does not divide, more than one leading batch element
    wb = (rng.standard_normal((2, 7, 5, 1801)) + 0.3 * np.arange(7)[None, :, None, None] - 0.2 * np.arange(5)[None, None, :, None] + 4.0).astype(dtype)
    cb = {"t": np.arange(2), "y": np.arange(7) * 1.0, "x": np.arange(5) * 1.0, "z": np.arange(1801)}
    db_, ob_ = pair(wb, ("t", "y", "x", "z"), cb)
    for det in ("constant", "linear"):
        worst = max(worst, check(xa.detrend(db_, ["y", "x"], det), o.detrend(ob_, ["y", "x"], det).transpose("t", "y", "x", "z"), tol))
    return worst


def run_fused_inner_cases(dtype="float64", shapes=((24, 20, 6), (16, 48, 3), (36, 30, 17), (40, 16, 33), (2, 28, 18, 5), (20, 24, 8), (2, 32, 16, 12))):
    """dim = ["y", "x"] of a real (..., y, x, t) array with both lengths >= 16 and smooth: the engine's two FUSED passes over the inner layout (csrc/fastn.h:
    fastn_cols_kernel on the [ny][nx t] view, fastn_fit_inner_kernel, fastn_irows_kernel; describe() says [fastn fused]) against the oracle -- odd and even
    element counts, a ragged last element block, a leading batch, every option the path takes.  xrft.py:395-409, 421-447."""
    rng = np.random.default_rng(5)
    tol = TOL[dtype]
    worst = 0.0
    for shape in shapes:
        ny, nx, nt = shape[-3:]
        ii, jj = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
        v = (rng.standard_normal(shape) + (0.05 * ii - 0.03 * jj + 2.0)[:, :, None] * (1.0 + np.arange(nt))[None, None, :]).astype(dtype)
        dims = ("y", "x", "t") if len(shape) == 3 else ("b", "y", "x", "t")
        c = {"y": np.arange(ny) * 0.5 + 1.0, "x": np.arange(nx) * 2.0 - 3.0, "t": np.arange(nt)}
        if len(shape) == 4:
            c["b"] = np.arange(shape[0])
        da, od = pair(v, dims, c)
        for kw in (dict(), dict(detrend="linear", window="hann"), dict(detrend="constant", shift=False, true_phase=False), dict(window="hamming", true_amplitude=False)):
            worst = max(worst, check(xa.fft(da, dim=["y", "x"], **kw), o.fft(od, dim=["y", "x"], **kw), tol))
            d = next(reversed(xa.api._plan_cache.values())).describe()
            assert "[fastn fused]" in d and "[inner layout]" in d, (shape, kw, d)
        for kw in (dict(detrend="linear", window="hann"), dict(scaling="spectrum", shift=False), dict(detrend="constant", window="hann", window_correction=True)):
            worst = max(worst, check(xa.power_spectrum(da, dim=["y", "x"], **kw), o.power_spectrum(od, dim=["y", "x"], **kw), tol))
            assert "[fastn fused]" in next(reversed(xa.api._plan_cache.values())).describe(), (shape, kw)
        # the axes named in the other order; a length with a factor the butterflies do not hold falls back to the composite of one-axis plans
        worst = max(worst, check(xa.power_spectrum(da, dim=["x", "y"], window="hann"), o.power_spectrum(od, dim=["x", "y"], window="hann"), tol))
        # the cross spectrum (and the cross phase) of two fields with different coordinate origins where the axes lie (round 6: pass 1 and the plane fit per field, pass 2 on both)
        w = (rng.standard_normal(shape) - (0.02 * jj + 1.0)[:, :, None]).astype(dtype)
        c2 = dict(c); c2["x"] = c["x"] + 1.25
        db, ob = pair(w, dims, c2)
        big = int(np.prod(shape)) > 1500000  # (the large shapes of the GPU suite: one option set per form -- the oracle's float64 passes are the cost)
        for kw in ((dict(detrend="linear", window="hann"),) if big else (dict(), dict(detrend="linear", window="hann"), dict(true_phase=False, shift=False), dict(scaling="spectrum", detrend="constant")) + ((dict(real_dim="x", detrend="linear"),) if nx % 2 == 0 else ())):
            worst = max(worst, check(xa.cross_spectrum(da, db, dim=["y", "x"], **kw), o.cross_spectrum(od, ob, dim=["y", "x"], **kw), tol))
            assert "[fastn fused]" in next(reversed(xa.api._plan_cache.values())).describe(), (shape, kw)
        if big:
            if nx % 2 == 0:
                worst = max(worst, check(xa.power_spectrum(da, dim=["y", "x"], real_dim="x", detrend="linear", window="hann"), o.power_spectrum(od, dim=["y", "x"], real_dim="x", detrend="linear", window="hann"), tol))
                assert "[fastn fused]" in next(reversed(xa.api._plan_cache.values())).describe(), shape
            continue
        gph, rph = xa.cross_phase(da, db, dim=["y", "x"], window="hann"), o.cross_phase(od, ob, dim=["y", "x"], window="hann")
        assert "[fastn fused]" in next(reversed(xa.api._plan_cache.values())).describe()
        mag = np.abs(o.cross_spectrum(od, ob, dim=["y", "x"], window="hann").values)
        dphi = np.abs(np.angle(np.exp(1j * (np.asarray(gph.values) - np.asarray(rph.values)))))
        assert (dphi * mag).max() / mag.max() < (1e-10 if dtype == "float64" else 3e-4)
        if nx % 2 == 0:  # real_dim along the second axis (round 6): rows of nx / 2 + 1 samples out of pass 2, the kept half counted twice in a power spectrum -- no transposed copy
            for fn, ofn, kws in ((xa.fft, o.fft, (dict(), dict(detrend="linear", window="hann"), dict(true_phase=False, true_amplitude=False))),
                                 (xa.power_spectrum, o.power_spectrum, (dict(), dict(detrend="linear", window="hann"), dict(scaling="spectrum", detrend="constant")))):
                for kw in kws:
                    worst = max(worst, check(fn(da, dim=["y", "x"], real_dim="x", **kw), ofn(od, dim=["y", "x"], real_dim="x", **kw), tol))
                    assert "[fastn fused]" in next(reversed(xa.api._plan_cache.values())).describe(), (shape, kw)
        if ny % 2 == 0:  # ... and along the axis of the pair that comes FIRST in memory (XRFTHIP_HALF_Y: rows ky = 0 .. ny/2, no twin rows)
            for fn, ofn, kw in ((xa.fft, o.fft, dict(detrend="linear", window="hann")), (xa.power_spectrum, o.power_spectrum, dict()), (xa.power_spectrum, o.power_spectrum, dict(scaling="spectrum", detrend="constant"))):
                worst = max(worst, check(fn(da, dim=["x", "y"], real_dim="y", **kw), ofn(od, dim=["x", "y"], real_dim="y", **kw), tol))
                assert "[fastn fused]" in next(reversed(xa.api._plan_cache.values())).describe(), (shape, kw)
            worst = max(worst, check(xa.cross_spectrum(da, db, dim=["x", "y"], real_dim="y", detrend="linear"), o.cross_spectrum(od, ob, dim=["x", "y"], real_dim="y", detrend="linear"), tol))
            assert "[fastn fused]" in next(reversed(xa.api._plan_cache.values())).describe(), shape
    return worst


def run_fused_mid_cases(dtype="float64", shapes=((24, 5, 20), (16, 3, 48), (36, 7, 30), (2, 28, 4, 18), (40, 2, 16), (146, 3, 24))):
    """dim = ["t", "x"] of a real (..., t, y, x) array -- wavenumber-frequency spectra of (time, lat, lon) fields, the two transform axes NOT adjacent -- as the engine's two
    FUSED passes (csrc/fastn.h: fastn_cols_kernel on the [nt][ny nx] view, a plane per (slab, y), fastn_irows_kernel with the lanes along x; describe() says [fastn fused])
    against the oracle.  xrft.py:395-409, 421-447."""
    rng = np.random.default_rng(6)
    tol = TOL[dtype]
    worst = 0.0
    for shape in shapes:
        nt, ny, nx = shape[-3:]
        ii, jj = np.meshgrid(np.arange(nt), np.arange(nx), indexing="ij")
        # (a trend per y element of 1 ... 2 times the base plane, at most ~40 x the noise: the residual's relative error grows with trend / noise -- the plane fit's own
        # conditioning, the same in the oracle)
        v = (rng.standard_normal(shape) + (0.05 * ii - 0.03 * jj + 2.0)[:, None, :] * (min(1.0, 500.0 / nt) * (1.0 + np.arange(ny) / ny))[None, :, None]).astype(dtype)
        dims = ("t", "y", "x") if len(shape) == 3 else ("b", "t", "y", "x")
        c = {"t": np.arange(nt) * 0.5 + 1.0, "y": np.arange(ny), "x": np.arange(nx) * 2.0 - 3.0}
        if len(shape) == 4:
            c["b"] = np.arange(shape[0])
        da, od = pair(v, dims, c)
        for kw in (dict(), dict(detrend="linear", window="hann"), dict(detrend="constant", shift=False, true_phase=False), dict(window="hamming", true_amplitude=False)):
            worst = max(worst, check(xa.fft(da, dim=["t", "x"], **kw), o.fft(od, dim=["t", "x"], **kw), tol))
            d = next(reversed(xa.api._plan_cache.values())).describe()
            assert "[fastn fused]" in d and "[mid %d]" % ny in d, (shape, kw, d)
        for kw in (dict(detrend="linear", window="hann"), dict(scaling="spectrum", shift=False), dict(detrend="constant", window="hann", window_correction=True)):
            worst = max(worst, check(xa.power_spectrum(da, dim=["t", "x"], **kw), o.power_spectrum(od, dim=["t", "x"], **kw), tol))
            assert "[fastn fused]" in next(reversed(xa.api._plan_cache.values())).describe(), (shape, kw)
        worst = max(worst, check(xa.power_spectrum(da, dim=["x", "t"], window="hann"), o.power_spectrum(od, dim=["x", "t"], window="hann"), tol))
        w = (rng.standard_normal(shape) + 0.5).astype(dtype)  # (two fields: the cross spectrum over the non-adjacent pair)
        c2 = dict(c); c2["t"] = c["t"] + 0.75
        db, ob = pair(w, dims, c2)
        big = int(np.prod(shape)) > 1500000
        for kw in ((dict(detrend="linear", window="hann"),) if big else (dict(), dict(detrend="linear", window="hann"), dict(true_phase=False, shift=False))):
            worst = max(worst, check(xa.cross_spectrum(da, db, dim=["t", "x"], **kw), o.cross_spectrum(od, ob, dim=["t", "x"], **kw), tol))
            assert "[fastn fused]" in next(reversed(xa.api._plan_cache.values())).describe(), (shape, kw)
        if big:
            continue
        if nx % 2 == 0:  # real_dim = the contiguous axis of the pair (round 6): the half output of the fused passes
            for fn, ofn, kws in ((xa.fft, o.fft, (dict(), dict(detrend="linear", window="hann"))), (xa.power_spectrum, o.power_spectrum, (dict(), dict(detrend="constant", window="hann")))):
                for kw in kws:
                    worst = max(worst, check(fn(da, dim=["t", "x"], real_dim="x", **kw), ofn(od, dim=["t", "x"], real_dim="x", **kw), tol))
                    assert "[fastn fused]" in next(reversed(xa.api._plan_cache.values())).describe(), (shape, kw)
        if nt % 2 == 0:  # real_dim = the FIRST axis of the pair (time): XRFTHIP_HALF_Y
            worst = max(worst, check(xa.power_spectrum(da, dim=["x", "t"], real_dim="t", detrend="linear", window="hann"), o.power_spectrum(od, dim=["x", "t"], real_dim="t", detrend="linear", window="hann"), tol))
            assert "[fastn fused]" in next(reversed(xa.api._plan_cache.values())).describe(), shape
    return worst


def run_fused_radial_code_forms(n=256):
    """The fused radial sums of the y-first float32 kernels (csrc/fasty.h): a radial bin map (what isotropic_*_spectrum hands
    over) is summed by a per-bin gather with no atomics; any other map through int64 fixed-point tables, its codes compact (first bin
    + step mask per 16 samples) or full (4 bytes per sample).  All three against numpy.bincount of the stored spectrum, through the C ABI's plan; a NaN in the data poisons its slab's bins, as a
    floating-point sum would."""
    import torch

    from xrft_amd import _lib, engine

    rng = np.random.default_rng(131)
    dev = xa.api._to_device(np.zeros(1, dtype=np.float32)).device
    nt, nb = 3, n // 4
    v = rng.standard_normal((nt, n, n)).astype(np.float32)
    t = torch.from_numpy(v).to(dev)
    f = np.fft.fftfreq(n)
    kr = np.sqrt(f[:, None] ** 2 + f[None, :] ** 2)
    radial = np.minimum((kr / kr.max() * nb).astype(np.int32), nb - 1)
    anymap = rng.integers(-1, nb, size=(n, n)).astype(np.int32)
    import os

    for name, bm, want, gather in (("radial", radial, "per-bin gather", "1"), ("radial", radial, "bin codes: compact", "0"), ("random", anymap, "bin codes: full", "1")):
        os.environ["XRFTHIP_ISO_GATHER"] = gather  # (read once, when a plan is created)
        for mode in (_lib.OUT_POWER, _lib.OUT_CROSS):
            plan = engine.SpectralPlan(2, nt, n, n, torch.float32, out_mode=mode, flags=_lib.ISO, scale=1.0, binmap=bm, nbins=nb)
            # (a radial map of a power spectrum of slabs up to 256 x 256 is summed inside the one-pass kernel, csrc/fasts.h; any other map
            # and the cross spectra take the two-pass kernels this test is about)
            assert ("[fasts]" if (n <= 256 and mode == _lib.OUT_POWER and name == "radial") else want) in plan.describe(), (name, plan.describe())
            t2 = torch.from_numpy(np.roll(v, 3, axis=2).copy()).to(dev) if mode == _lib.OUT_CROSS else None
            out, iso = plan.execute(t, t2)
            out2, iso2 = plan.execute(t, t2)
            assert torch.equal(iso, iso2)
            spec = out.cpu().numpy().astype(np.complex128 if mode == _lib.OUT_CROSS else np.float64)
            ok = bm.ravel() >= 0
            for b in range(nt):
                w = spec[b].ravel()[ok]
                ref = np.bincount(bm.ravel()[ok], weights=w.real, minlength=nb).astype(np.complex128)
                if mode == _lib.OUT_CROSS:
                    ref = ref + 1j * np.bincount(bm.ravel()[ok], weights=w.imag, minlength=nb)
                mag = np.bincount(bm.ravel()[ok], weights=np.abs(w), minlength=nb)
                got = iso.cpu().numpy()[b]
                assert np.all(np.abs(got - (ref if mode == _lib.OUT_CROSS else ref.real)) <= 2e-6 * np.maximum(mag, 1e-300)), (name, mode, b)
        # a NaN sample: every bin of that slab is NaN (the transform spreads it), the other slabs are untouched
        vn = v.copy(); vn[1, 5, 7] = np.nan
        plan = engine.SpectralPlan(2, nt, n, n, torch.float32, out_mode=_lib.OUT_POWER, flags=_lib.ISO | _lib.NO_SPECTRUM_OUT, scale=1.0, binmap=bm, nbins=nb)
        _, iso = plan.execute(torch.from_numpy(vn).to(dev))
        g = iso.cpu().numpy()
        used = np.bincount(bm.ravel()[bm.ravel() >= 0], minlength=nb) > 0
        assert np.all(np.isnan(g[1][used])) and np.all(np.isfinite(g[0])) and np.all(np.isfinite(g[2])), name
    os.environ.pop("XRFTHIP_ISO_GATHER", None)


def run_fastg_cases(shape=(3, 50, 50), dtype="float32", full=True):
    """Small real slabs of any smooth shape (50 x 50 boxes -- the reference's documented workload --, 96 x 96, 100 x 100, odd row counts), both
    precisions: the LDS-resident one-pass kernel with run-time radices (csrc/fastg.h) against the oracle -- power spectra with every detrend /
    window / shift combination, the complex spectrum with and without the true phase (an odd ny makes the ifftshift a rotation, not (-1)^k)."""
    rng = np.random.default_rng(53)
    tol = TOL[dtype]
    a = _cube(rng, shape, dtype)
    da, od = pair(a, D3, _coords3(shape, y0=1.0, x0=-3.0))
    worst = 0.0

    def on_fastg():
        return "[fastg]" in next(reversed(xa.api._plan_cache.values())).describe()

    worst = max(worst, check(xa.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"),
                             o.power_spectrum(od, dim=["y", "x"], detrend="linear", window="hann"), tol))
    assert on_fastg()
    if not full:
        return worst
    for kw in (dict(), dict(detrend="constant", window="hamming", scaling="spectrum", window_correction=True), dict(detrend="linear"),
               dict(shift=False, window="bartlett", density=False)):
        worst = max(worst, check(xa.power_spectrum(da, dim=["y", "x"], **kw), o.power_spectrum(od, dim=["y", "x"], **kw), tol))
        assert on_fastg()
    for kw in (dict(), dict(true_phase=False), dict(shift=False, true_phase=False, true_amplitude=False), dict(detrend="linear", window="hann")):
        worst = max(worst, check(xa.fft(da, dim=["y", "x"], **kw), o.fft(od, dim=["y", "x"], **kw), tol))
        assert on_fastg()
    # cross spectrum of two fields (xrft.py:753-835): both tiles in the workgroup's LDS when they fit, F0 conj(F1) on the way out
    b = (np.roll(a, 2, axis=-1) * 0.5 + _cube(rng, shape, dtype, trend=False) * 0.3).astype(dtype)
    db, ob = pair(b, D3, _coords3(shape, y0=1.5, x0=-2.0))  # (other origins: a true-phase factor that is not 1)
    two_fit = 2 * shape[1] * (shape[2] // 2 + 2 if shape[2] % 2 == 0 else shape[2] + 1) * (8 if dtype == "float32" else 16) < 140 * 1024
    for kw in (dict(), dict(true_phase=False, detrend="linear", window="hann"), dict(shift=False, scaling="spectrum", detrend="constant")):
        worst = max(worst, check(xa.cross_spectrum(da, db, dim=["y", "x"], **kw), o.cross_spectrum(od, ob, dim=["y", "x"], **kw), tol))
        assert on_fastg() or not two_fit
    # real_dim: the half spectrum along x, unshifted, the doubled interior columns of a power spectrum (xrft.py:400-404, 673-682)
    for kw in (dict(), dict(detrend="linear", window="hann"), dict(scaling="spectrum", detrend="constant")):
        worst = max(worst, check(xa.power_spectrum(da, dim=["y"], real_dim="x", **kw), o.power_spectrum(od, dim=["y"], real_dim="x", **kw), tol))
        assert on_fastg()
    for kw in (dict(), dict(true_phase=False, detrend="linear", window="hann")):
        worst = max(worst, check(xa.fft(da, dim=["y"], real_dim="x", **kw), o.fft(od, dim=["y"], real_dim="x", **kw), tol))
        assert on_fastg()
    # the radial sums inside the same pass (xrft.py:895-906, 1013-1095): per-bin position lists, any bin map
    for kw in (dict(detrend="linear", window="hann"), dict(truncate=True), dict(detrend="constant", window="hamming", nfactor=2)):
        xa.api._plan_cache.clear()
        worst = max(worst, check(xa.isotropic_power_spectrum(da, dim=["y", "x"], **kw), o.isotropic_power_spectrum(od, dim=["y", "x"], **kw), max(tol, 1e-9)))
        assert any("[fastg]" in p.describe() and "radial sums" in p.describe() for p in xa.api._plan_cache.values())
    # ... and of a cross spectrum (complex sums; a sample of the right half plane is the conjugate of the stored product); two fields with different
    # origins carry a true-phase factor per sample: the other paths, the same numbers
    dc, oc = pair(b, D3, _coords3(shape, y0=1.0, x0=-3.0))
    for other, oth_o, fused in ((dc, oc, True), (db, ob, False)):
        for kw in (dict(detrend="linear", window="hann"), dict(truncate=True)):
            xa.api._plan_cache.clear()
            worst = max(worst, check(xa.isotropic_cross_spectrum(da, other, dim=["y", "x"], **kw), o.isotropic_cross_spectrum(od, oth_o, dim=["y", "x"], **kw), max(tol, 1e-9)))
            assert (not fused) or (not two_fit) or any("[fastg cross" in p.describe() and "radial sums" in p.describe() for p in xa.api._plan_cache.values())
    import torch

    from xrft_amd import _lib, engine

    nt, ny, nx = shape
    nb = max(2, min(ny, nx) // 4)
    tdt = torch.float64 if dtype == "float64" else torch.float32
    t = xa.api._to_device(np.ascontiguousarray(a))
    anymap = rng.integers(-1, nb, size=(ny, nx)).astype(np.int32)
    if two_fit:  # the C ABI's plan: complex sums of any bin map against numpy.bincount of the stored cross spectrum
        t2 = xa.api._to_device(np.ascontiguousarray(b))
        plan = engine.SpectralPlan(2, nt, ny, nx, tdt, out_mode=_lib.OUT_CROSS, flags=_lib.ISO, scale=0.5, binmap=anymap, nbins=nb)
        assert "[fastg cross" in plan.describe()
        out, iso = plan.execute(t, t2)
        assert torch.equal(iso, plan.execute(t, t2)[1])
        spec = out.cpu().numpy().astype(np.complex128)
        ok = anymap.ravel() >= 0
        for bb in range(nt):
            w = spec[bb].ravel()[ok]
            ref = np.bincount(anymap.ravel()[ok], weights=w.real, minlength=nb) + 1j * np.bincount(anymap.ravel()[ok], weights=w.imag, minlength=nb)
            mag = np.bincount(anymap.ravel()[ok], weights=np.abs(w), minlength=nb)
            assert np.all(np.abs(iso.cpu().numpy()[bb] - ref) <= (1e-13 if dtype == "float64" else 2e-6) * np.maximum(mag, 1e-300))
    for flags in (_lib.ISO, _lib.ISO | _lib.NO_SPECTRUM_OUT):
        plan = engine.SpectralPlan(2, nt, ny, nx, tdt, out_mode=_lib.OUT_POWER, flags=flags, scale=0.5, binmap=anymap, nbins=nb)
        assert "[fastg]" in plan.describe()
        out, iso = plan.execute(t)
        _, iso2 = plan.execute(t)
        assert torch.equal(iso, iso2)  # no atomics: the same bits
        if out is not None:
            spec = out.cpu().numpy().astype(np.float64)
            ok = anymap.ravel() >= 0
            for b in range(nt):
                ref = np.bincount(anymap.ravel()[ok], weights=spec[b].ravel()[ok], minlength=nb)
                assert np.all(np.abs(iso.cpu().numpy()[b] - ref) <= (1e-13 if dtype == "float64" else 2e-6) * np.maximum(ref, 1e-300))
            keep = iso
        else:
            assert torch.equal(iso, keep)
    vn = np.array(a, copy=True)
    vn[nt - 1, 1, 2] = np.nan  # poisons its own slab only
    plan = engine.SpectralPlan(2, nt, ny, nx, tdt, out_mode=_lib.OUT_POWER, flags=_lib.ISO | _lib.NO_SPECTRUM_OUT, scale=1.0, binmap=anymap, nbins=nb)
    g = plan.execute(xa.api._to_device(vn))[1].cpu().numpy()
    used = np.bincount(anymap.ravel()[anymap.ravel() >= 0], minlength=nb) > 0
    assert np.all(np.isnan(g[nt - 1][used])) and (nt == 1 or np.all(np.isfinite(g[: nt - 1])))
    return worst


def run_fastm_radial_code_forms(ny=360, nx=240, dtype="float64"):
    """The fused radial sums of the mixed-radix kernels (csrc/fastm.h): a radial bin map is gathered per bin from the spectra in LDS with no
    atomics, any other map (and a radial one with XRFTHIP_ISO_GATHER=0) goes through the int64 fixed-point tables.  Both against
    numpy.bincount of the stored spectrum through the C ABI's plan, power and cross spectra, repeats bit for bit; a NaN poisons its own slab only."""
    import os

    import torch

    from xrft_amd import _lib, engine

    rng = np.random.default_rng(137)
    dev = xa.api._to_device(np.zeros(1, dtype=np.float32)).device
    tdt = torch.float64 if dtype == "float64" else torch.float32
    nt, nb = 3, min(ny, nx) // 4
    v = rng.standard_normal((nt, ny, nx)).astype(dtype)
    t = torch.from_numpy(v).to(dev)
    kr = np.sqrt((np.fft.fftfreq(ny) * 1.3)[:, None] ** 2 + np.fft.fftfreq(nx)[None, :] ** 2)
    radial = np.minimum((kr / kr.max() * nb).astype(np.int32), nb - 1)
    anymap = rng.integers(-1, nb, size=(ny, nx)).astype(np.int32)
    tol = 1e-12 if dtype == "float64" else 2e-6
    try:
        for name, bm, want, gather in (("radial", radial, "per-bin gather", "1"), ("radial", radial, "fixed-point tables", "0"), ("random", anymap, "fixed-point tables", "1")):
            os.environ["XRFTHIP_ISO_GATHER"] = gather  # (read when the bin map is set)
            for mode in (_lib.OUT_POWER, _lib.OUT_CROSS):
                plan = engine.SpectralPlan(2, nt, ny, nx, tdt, out_mode=mode, flags=_lib.ISO, scale=1.0, binmap=bm, nbins=nb)
                assert "[fastm radial sums]" in plan.describe() and want in plan.describe(), (name, plan.describe())
                t2 = torch.from_numpy(np.roll(v, 3, axis=2).copy()).to(dev) if mode == _lib.OUT_CROSS else None
                out, iso = plan.execute(t, t2)
                out2, iso2 = plan.execute(t, t2)
                assert torch.equal(iso, iso2)
                spec = out.cpu().numpy().astype(np.complex128 if mode == _lib.OUT_CROSS else np.float64)
                ok = bm.ravel() >= 0
                for b in range(nt):
                    w = spec[b].ravel()[ok]
                    ref = np.bincount(bm.ravel()[ok], weights=w.real, minlength=nb).astype(np.complex128)
                    if mode == _lib.OUT_CROSS:
                        ref = ref + 1j * np.bincount(bm.ravel()[ok], weights=w.imag, minlength=nb)
                    mag = np.bincount(bm.ravel()[ok], weights=np.abs(w), minlength=nb)
                    got = iso.cpu().numpy()[b]
                    assert np.all(np.abs(got - (ref if mode == _lib.OUT_CROSS else ref.real)) <= tol * np.maximum(mag, 1e-300)), (name, mode, b)
            vn = v.copy(); vn[1, 5, 7] = np.nan
            plan = engine.SpectralPlan(2, nt, ny, nx, tdt, out_mode=_lib.OUT_POWER, flags=_lib.ISO | _lib.NO_SPECTRUM_OUT, scale=1.0, binmap=bm, nbins=nb)
            _, iso = plan.execute(torch.from_numpy(vn).to(dev))
            g = iso.cpu().numpy()
            used = np.bincount(bm.ravel()[bm.ravel() >= 0], minlength=nb) > 0
            assert np.all(np.isnan(g[1][used])) and np.all(np.isfinite(g[0])) and np.all(np.isfinite(g[2])), name
    finally:
        os.environ.pop("XRFTHIP_ISO_GATHER", None)


def run_nan_in_isotropic_spectra():
    """A NaN sample poisons its own slab's isotropic spectrum and nothing else, on every path that takes radial sums (the
    reference sums in floating point, xrft.py:895-906; ADVICE r2: fixed-point tables gave +inf or 0): fasty.h (256^2 float32),
    fastm.h fused (360^2 float64), the generic plans + the stand-alone pass (48 x 40 float64)."""
    rng = np.random.default_rng(141)
    for shape, dt in (((3, 256, 256), "float32"), ((3, 360, 360), "float64"), ((3, 48, 40), "float64")):
        v = rng.standard_normal(shape).astype(dt)
        v[1, 5, 7] = np.nan
        c = _coords3(shape)
        da = xa.DataArray(v, D3, c)
        for fn in (lambda: xa.isotropic_power_spectrum(da, dim=["y", "x"], window="hann"),
                   lambda: xa.isotropic_cross_spectrum(da, xa.DataArray(np.roll(v, 2, axis=2).copy(), D3, c), dim=["y", "x"], window="hann")):
            g = np.asarray(fn().values)
            assert np.all(np.isnan(g[1])), (shape, dt)
            assert np.all(np.isfinite(g[0])) and np.all(np.isfinite(g[2])), (shape, dt)


def run_small_slab_walk_cases(shapes=((7, 256, 256), (5, 128, 256), (9, 64, 64)), grid="2"):
    """csrc/fasts.h with a resident set of workgroups walking the slabs (XRFTHIP_FASTS_GRID: what a long batch of 256 x 256 slabs gets by default): every
    workgroup asks for its next slab while the staged rows of the current one leave.  Power spectrum, its radial sums and the complex form (which does
    not prefetch) against the oracle, and bit for bit against the one-workgroup-per-slab launch."""
    import os
    rng = np.random.default_rng(77)
    worst = 0.0
    old = os.environ.get("XRFTHIP_FASTS_GRID")
    try:
        for shape in shapes:
            nt, ny, nx = shape
            v = (rng.standard_normal(shape) * (1.0 + np.arange(nt))[:, None, None] + 0.02 * np.arange(ny)[None, :, None] - 0.01 * np.arange(nx)[None, None, :] + 2.0).astype("float32")
            c = {"t": np.arange(nt), "y": np.arange(ny) * 0.5, "x": np.arange(nx) * 0.25}
            da, od = pair(v, ("t", "y", "x"), c)
            res = {}
            for g in ("0", grid):
                os.environ["XRFTHIP_FASTS_GRID"] = g
                xa.api.clear_plan_cache()
                ps = xa.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")
                assert "[fasts]" in next(reversed(xa.api._plan_cache.values())).describe()
                iso = xa.isotropic_power_spectrum(da, dim=["y", "x"], detrend="constant", window="hann")
                ft = xa.fft(da, dim=["y", "x"], detrend="linear")
                res[g] = (np.asarray(ps.values), np.asarray(iso.values), np.asarray(ft.values))
                if g != "0":
                    worst = max(worst, check(ps, o.power_spectrum(od, dim=["y", "x"], detrend="linear", window="hann"), 3e-4))
                    worst = max(worst, check(iso, o.isotropic_power_spectrum(od, dim=["y", "x"], detrend="constant", window="hann"), 3e-4))
                    worst = max(worst, check(ft, o.fft(od, dim=["y", "x"], detrend="linear"), 3e-4))
            for a_, b_ in zip(res["0"], res[grid]):
                assert np.array_equal(a_, b_)
    finally:
        if old is None:
            os.environ.pop("XRFTHIP_FASTS_GRID", None)
        else:
            os.environ["XRFTHIP_FASTS_GRID"] = old
        xa.api.clear_plan_cache()
    return worst


def run_long_rows_resident_cases(n=32768, nt=600):
    """Rows of 16384 / 32768 float32 samples in a batch long enough for the resident set of csrc/fastr.h (fastr2_kernel: 256 / 512 workgroups walk the rows,
    classes of workgroups start a few microseconds apart): fft with and without the true phase, dft, power spectrum -- rows from the start, the middle and
    the end of the batch against the oracle, and every row bit for bit against the one-workgroup-per-row launch (XRFTHIP_FASTR_GRID=0)."""
    import os
    rng = np.random.default_rng(n + nt)
    v = (rng.standard_normal((nt, n)) + 1.0).astype("float32")
    c = {"t": np.arange(nt), "x": np.arange(n) * 0.5 + 2.0}
    da, _ = pair(v, ("t", "x"), c)
    pick = sorted(set([0, 1, 255, 256, 257, 511, 512, nt // 2, nt - 2, nt - 1]) & set(range(nt)))
    od = o.OArr(v[pick].astype("float64"), ("t", "x"), {"t": np.arange(len(pick)), "x": c["x"]})
    calls = ((xa.fft, o.fft, dict(dim=["x"])), (xa.fft, o.fft, dict(dim=["x"], true_phase=False, shift=False)), (xa.dft, o.dft, dict(dim="x")),
             (xa.power_spectrum, o.power_spectrum, dict(dim=["x"], detrend="linear", window="hann")))
    worst = 0.0
    old = {k: os.environ.get(k) for k in ("XRFTHIP_FASTR_GRID", "XRFTHIP_FASTR_STAGGER")}
    try:
        res = {}
        for mode in ("default", "per-row"):
            if mode == "per-row":
                os.environ["XRFTHIP_FASTR_GRID"] = "0"
                os.environ["XRFTHIP_FASTR_STAGGER"] = "0"
            xa.api.clear_plan_cache()
            res[mode] = []
            for fn, ofn, kw in calls:
                g = fn(da, **kw)
                assert "[fastr]" in next(reversed(xa.api._plan_cache.values())).describe()
                gv = np.asarray(g.values)
                res[mode].append(gv)
                if mode == "default":
                    r = np.asarray(ofn(od, **kw).values)
                    err = float(np.abs(gv[pick] - r).max() / np.abs(r).max())
                    assert err < 2e-5, (kw, err)
                    worst = max(worst, err)
        for a_, b_ in zip(res["default"], res["per-row"]):
            assert np.array_equal(a_, b_)
    finally:
        for k, val in old.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val
        xa.api.clear_plan_cache()
    return worst


def run_inverse_non_trailing_pairs(cfgs=(((32, 48, 6), ("y", "x", "t"), ["y", "x"], "float32"), ((36, 40, 7), ("y", "x", "t"), ["x", "y"], "float64"),
                                          ((32, 6, 48), ("t", "y", "x"), ["t", "x"], "float64"), ((2, 24, 20, 3), ("b", "y", "x", "t"), ["y", "x"], "complex128"))):
    """xrft.ifft over two axes that are not the trailing pair (xrft.py:586-621): ifftn is separable, each axis has a plan that runs where it lies -- one axis at a time,
    no transposed copy; the result is contiguous in the input's layout.  Every true_phase / shift combination against the oracle."""
    import warnings
    rng = np.random.default_rng(17)
    worst = 0.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for shape, dims, td, dt in cfgs:
            v = rng.standard_normal(shape)
            if dt.startswith("complex"):
                v = v + 1j * rng.standard_normal(shape)
            v = v.astype(dt)
            c = {d: np.arange(n) * 0.5 + 1.0 for d, n in zip(dims, shape)}
            da, od = pair(v, dims, c)
            tol = 1e-10 if dt in ("float64", "complex128") else 3e-4
            fd = ["freq_" + d for d in td]
            for kw in (dict(), dict(true_phase=False, shift=False), dict(shift=False), dict(true_phase=False)):
                F, Fo = xa.fft(da, dim=td, **kw), o.fft(od, dim=td, **kw)
                g, r = xa.ifft(F, dim=fd, **kw), o.ifft(Fo, dim=fd, **kw)
                worst = max(worst, check_values(g, r, tol))
                assert tuple(g.dims) == tuple(dims)
                data = g.data
                assert data.is_contiguous() if hasattr(data, "is_contiguous") else np.asarray(data).flags["C_CONTIGUOUS"]
    return worst
