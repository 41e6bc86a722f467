"""Shared parity cases: every case builds seeded inputs, runs the product API (xrft_amd, bound to whichever
library the calling test selected) and the CPU oracle, and returns (got, ref).  Used by

  tests/test_emulated_api.py   CPU: product host code + kernels compiled for the emulator (index arithmetic)
  tests/test_gpu_parity.py     GPU: the real libxrft_hip.so through the C ABI  (-m gpu)

Tolerances follow BASELINE.json: 1e-6 relative in float64, 1e-3 in float32 (relative to max |reference|);
the tests assert much tighter float64 agreement (1e-10).
"""
import warnings

import numpy as np

import xrft_amd as xa
from oracle import xrft_oracle as o

warnings.simplefilter("ignore")

TOL = {"float64": 1e-10, "float32": 3e-4, "complex128": 1e-10, "complex64": 3e-4}


def pair(data, dims, coords=None):
    return xa.DataArray(data, dims, coords), o.OArr(data, dims, coords)


def rel_err(got, ref):
    g = np.asarray(got.values)
    r = np.asarray(ref.values)
    assert g.shape == r.shape, (g.shape, r.shape)
    den = max(float(np.abs(r).max()), 1e-300)
    return float(np.abs(g - r).max()) / den


def check(got, ref, tol):
    assert tuple(got.dims) == tuple(ref.dims), (got.dims, ref.dims)
    for d in ref.dims:
        if d in ref.coords:
            gv = np.asarray(got[d].values)
            rv = np.asarray(ref.coord(d))
            if rv.dtype.kind in "fc":
                np.testing.assert_allclose(gv.astype(np.float64), rv.astype(np.float64), rtol=1e-13, atol=0,
                                           equal_nan=True)
            else:
                assert np.array_equal(gv, rv)
            ra = ref.coord_attrs.get(d, {})
            for k, v in ra.items():
                assert k in got[d].attrs, (d, k)
                np.testing.assert_allclose(float(got[d].attrs[k]), float(v), rtol=1e-13)
    err = rel_err(got, ref)
    assert err < tol, f"rel err {err:.3e} >= {tol:.1e}"
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
    elif kind == "iso":
        got = xa.isotropic_cross_spectrum(da, db, dim=["y", "x"], window="hann", detrend="linear")
        ref = o.isotropic_cross_spectrum(od, ob, dim=["y", "x"], window="hann", detrend="linear")
        return check(got, ref, TOL[dtype])
    else:
        raise KeyError(kind)
    return check(xa.cross_spectrum(da, db, **kw), o.cross_spectrum(od, ob, **kw), TOL[dtype])


CROSS_KINDS = ["true_phase_window", "nophase_spectrum", "real_dim", "one_dim", "iso"]


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
