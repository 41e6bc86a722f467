"""CPU-side functional tests of the PRODUCT host code and kernels: the same sources as libxrft_hip.so, compiled by
g++ against tests/emu/hip_emu.h (fibers stand in for GPU threads), driven through the same C ABI / ctypes binding
and compared with the CPU oracle.  This exercises index arithmetic and host logic only -- it is NOT a product
fallback and says nothing about the GPU build; the GPU parity tests are tests/test_gpu_parity.py (-m gpu)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emu"))
import build_emu  # noqa: E402

from xrft_amd import _lib, api  # noqa: E402

import cases  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def emulated_library():
    api._plan_cache.clear()
    _lib._load_for_testing(build_emu.build())
    yield
    api._plan_cache.clear()
    _lib._state.update(dll=None, path=None, device="cuda")


@pytest.mark.parametrize("name,dtype", cases.all_case_params())
def test_case(name, dtype):
    cases.run_case(name, dtype)


@pytest.mark.parametrize("kind", cases.CROSS_KINDS)
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_cross(kind, dtype):
    cases.run_cross_case(kind, dtype)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_true_phase(dtype):
    cases.run_true_phase_case(dtype)


def test_golden_ps2d(golden_dir):
    """Committed oracle fixtures (tests/golden/case_ps2d_f64.npz) through the product path."""
    import xrft_amd as xa

    z = np.load(os.path.join(golden_dir, "case_ps2d_f64.npz"))
    data = z["data"]
    nt, ny, nx = data.shape
    da = xa.DataArray(data, ("time", "y", "x"), {"time": np.arange(nt), "y": np.arange(ny) * 0.5, "x": np.arange(nx) * 2.0})
    for n, combo in enumerate(z["combos"]):
        det, win, scaling, wc = str(combo).split("|")
        ps = xa.power_spectrum(da, dim=["y", "x"], detrend=None if det == "None" else det,
                               window=None if win == "None" else win, scaling=scaling, window_correction=bool(int(wc)))
        ref = z[f"ps_{n}"]
        assert np.abs(ps.values - ref).max() / np.abs(ref).max() < 1e-10, combo
    assert np.array_equal(ps["freq_y"].values, z["freq_y"]) and np.array_equal(ps["freq_x"].values, z["freq_x"])


def test_four_step_paths(monkeypatch):
    """Force the four-step decompositions (normally only used when a sequence does not fit one LDS tile)."""
    import xrft_amd as xa
    from oracle import xrft_oracle as o

    rng = np.random.default_rng(5)
    monkeypatch.setenv("XRFTHIP_X_FOURSTEP_MIN", "2")
    monkeypatch.setenv("XRFTHIP_Y_FOURSTEP_MIN", "2")
    api._plan_cache.clear()
    for shape, dt in [((2, 12, 60), "float64"), ((2, 16, 36), "complex128"), ((3, 1, 120), "float64")]:
        v = rng.standard_normal(shape)
        if dt.startswith("complex"):
            v = v + 1j * rng.standard_normal(shape)
        c = {"t": np.arange(shape[0]), "y": np.arange(shape[1]) * 0.5, "x": np.arange(shape[2]) * 0.25 + 1.0}
        da, od = cases.pair(v.astype(dt), ("t", "y", "x"), c)
        dims = ["y", "x"] if shape[1] > 1 else ["x"]
        cases.check(xa.fft(da, dim=dims, detrend="linear", window="hann"), o.fft(od, dim=dims, detrend="linear", window="hann"), 1e-10)
        cases.check(xa.power_spectrum(da, dim=dims), o.power_spectrum(od, dim=dims), 1e-10)
        if dt == "float64":
            cases.check(xa.power_spectrum(da, dim=dims[:-1], real_dim="x"), o.power_spectrum(od, dim=dims[:-1], real_dim="x"), 1e-10)
    api._plan_cache.clear()


def test_groups_and_batches(monkeypatch):
    """Several slab groups per call (workspace re-use) and a batch that is not a multiple of the group size."""
    import xrft_amd as xa
    from oracle import xrft_oracle as o

    monkeypatch.setenv("XRFTHIP_GROUP", "2")
    api._plan_cache.clear()
    rng = np.random.default_rng(6)
    v = rng.standard_normal((5, 8, 12)) + np.arange(12) * 0.1
    c = {"t": np.arange(5), "y": np.arange(8.0), "x": np.arange(12.0)}
    da, od = cases.pair(v, ("t", "y", "x"), c)
    cases.check(xa.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann"),
                o.power_spectrum(od, dim=["y", "x"], detrend="linear", window="hann"), 1e-10)
    cases.check(xa.isotropic_power_spectrum(da, dim=["y", "x"], detrend="constant"),
                o.isotropic_power_spectrum(od, dim=["y", "x"], detrend="constant"), 1e-10)
    da2, od2 = cases.pair(rng.standard_normal((5, 8, 12)), ("t", "y", "x"), c)
    cases.check(xa.cross_spectrum(da, da2, dim=["y", "x"], detrend="linear"),
                o.cross_spectrum(od, od2, dim=["y", "x"], detrend="linear"), 1e-10)
    api._plan_cache.clear()


def test_unsupported_length_is_loud_at_the_c_abi():
    """A plan the kernels cannot serve -- two transform axes, one a prime whose Bluestein transform does not fit the LDS -- is refused
    by xrfthip_plan_create with XRFTHIP_UNSUPPORTED_LENGTH (never computed some other way behind the caller's back); the Python
    layer then transforms the axes one at a time (test_long_prime_lengths_through_global_bluestein)."""
    import torch

    from xrft_amd import engine

    with pytest.raises(_lib.XrftHipError) as ei:
        engine.SpectralPlan(ndim=2, batch=2, ny=8, nx=10007, dtype=torch.float64, out_mode=_lib.OUT_POWER, detrend=_lib.DETREND_NONE, flags=0, scale=1.0)
    assert ei.value.status == _lib.UNSUPPORTED_LENGTH


def test_bluestein_precision_policy():
    """float32 data on a Bluestein length inside one tile stay in float32 (1.2e-4 per bin on the GPU); a length that goes through global
    memory (api._bluestein_1d) runs in float64 by default, and the policy switch keeps float32 arithmetic there (max-norm bound only)."""
    import xrft_amd as xa
    rng = np.random.default_rng(3)
    v = rng.standard_normal((2, 262)).astype("float32")
    da = xa.DataArray(v, ("t", "x"), {"t": np.arange(2), "x": np.arange(262) * 1.0})
    assert xa.bluestein_in_float64() is True
    g = xa.fft(da, dim="x", true_phase=False, true_amplitude=False)
    plan = next(reversed(api._plan_cache.values()))
    assert plan.uses_bluestein() and plan.dtype == torch.float32 and g.data.dtype == torch.complex64
    ref = np.fft.fftshift(np.fft.fft(v.astype("float64"), axis=1), axes=1)
    assert np.abs(g.values - ref).max() / np.abs(ref).max() < 1e-6
    n = 10007  # prime, beyond the tile
    v = rng.standard_normal((2, n)).astype("float32")
    da = xa.DataArray(v, ("t", "x"), {"t": np.arange(2), "x": np.arange(n) * 1.0})
    ref = np.fft.fftshift(np.fft.fft(v.astype("float64"), axis=1), axes=1)
    g64 = xa.fft(da, dim="x", true_phase=False, true_amplitude=False)
    assert g64.data.dtype == torch.complex64
    try:
        xa.bluestein_in_float64(False)
        g32 = xa.fft(da, dim="x", true_phase=False, true_amplitude=False)
        assert g32.data.dtype == torch.complex64
    finally:
        xa.bluestein_in_float64(True)
    e64 = np.abs(g64.values - ref).max() / np.abs(ref).max()
    e32 = np.abs(g32.values - ref).max() / np.abs(ref).max()
    assert e64 < 2e-7 and e64 <= e32 < 1e-5, (e64, e32)


def test_long_prime_lengths_through_global_bluestein():
    cases.run_long_prime_cases(lengths=((9001, "float64"),))  # (float32 and complex input of such lengths: the GPU suite; the float32 policy: test_bluestein_precision_policy)


@pytest.mark.parametrize("ny,nx,nt,shift,det,win", [
    (4096, 4096, 1, True, "linear", "hann"),
    (4096, 4096, 1, False, None, None),
    (1024, 1024, 3, True, "linear", "hann"),
    (1024, 2048, 2, False, "constant", "hamming"),
    (2048, 1024, 2, True, "linear", None),
    (2048, 2048, 1, True, None, "hann"),
    (1024, 4096, 1, True, "linear", "hann"),
    (4096, 1024, 1, False, "linear", "hann"),
    (256, 256, 5, True, "linear", "hann"),      # (256 x 256: one slab per workgroup, one pass -- csrc/fasts.h)
    (256, 256, 2, False, None, None),
    (256, 256, 3, True, "constant", "hamming"),
    (256, 256, 2, False, "linear", None),
    (128, 128, 3, True, "linear", "hann"),      # (64 | 128 | 256 points per axis: the same kernel, smaller workgroups)
    (64, 64, 5, True, "linear", "hann"),
    (64, 64, 2, False, None, None),
    (128, 256, 2, True, "constant", "hamming"),
    (256, 64, 2, False, "linear", "hann"),
    (64, 128, 3, True, "linear", None),
    (128, 64, 2, True, None, "hann"),
    (256, 128, 2, True, "linear", "hann"),
    (64, 256, 2, True, "linear", "hann"),
    (512, 256, 3, True, "linear", "hamming"),
    (256, 1024, 2, False, "constant", None),
    (512, 512, 2, True, None, "hann"),
])
def test_fastp2_path(ny, nx, nt, shift, det, win):
    """The specialised power-of-two float32 power-spectrum kernels (fasty.h) against numpy in float64."""
    import scipy.signal as sps
    import xrft_amd as xa

    rng = np.random.default_rng(42)
    v = rng.standard_normal((nt, ny, nx)).astype(np.float32)
    v += (0.01 * np.arange(ny, dtype=np.float32))[None, :, None] + (-0.02 * np.arange(nx, dtype=np.float32) + 3)[None, None, :]
    v *= (1 + np.arange(nt, dtype=np.float32))[:, None, None]
    c = {"t": np.arange(nt), "y": np.arange(ny) * 1.0, "x": np.arange(nx) * 1.0}
    ps = xa.power_spectrum(xa.DataArray(v, ("t", "y", "x"), c), dim=["y", "x"], detrend=det, window=win, shift=shift)
    plan = next(reversed(api._plan_cache.values()))
    assert "[fast" in plan.describe()
    for t in range(nt):
        x = v[t].astype(np.float64)
        if det == "constant":
            x = x - x.mean()
        elif det == "linear":
            ii, jj = np.meshgrid(np.arange(ny) - (ny - 1) / 2, np.arange(nx) - (nx - 1) / 2, indexing="ij")
            x = x - (x.mean() + (ii * x).sum() / (ii * ii).sum() * ii + (jj * x).sum() / (jj * jj).sum() * jj)
        if win:
            x = x * getattr(sps.windows, win)(ny, sym=False)[:, None] * getattr(sps.windows, win)(nx, sym=False)[None, :]
        F = np.fft.fft2(x)
        if shift:
            F = np.fft.fftshift(F)
        ref = np.abs(F) ** 2 / (ny * nx)
        g = ps.values[t].astype(np.float64)
        assert np.abs(g - ref).max() / ref.max() < 2e-5
        assert np.abs(g - ref).sum() / ref.sum() < 5e-6
    api._plan_cache.clear()


@pytest.mark.parametrize("ny,nx,nt,det,win,truncate", [(1024, 1024, 3, "linear", "hann", True), (1024, 2048, 2, None, None, False),
                                                     (2048, 1024, 1, "constant", "hann", True), (256, 256, 4, "linear", "hann", True),
                                                     (512, 256, 3, None, "hann", False),
                                                     # one slab per workgroup (csrc/fasts.h): the radial sums from the staged rows
                                                     (256, 256, 2, None, None, False), (128, 128, 5, "linear", "hann", True),
                                                     (64, 64, 9, "linear", "hann", False), (128, 256, 3, "constant", "hamming", True),
                                                     (256, 64, 3, "linear", "hann", False), (64, 128, 4, None, "hann", True)])
def test_fastp2_isotropic(ny, nx, nt, det, win, truncate):
    """isotropic_power_spectrum through the specialised kernels: radial sums taken inside the column pass."""
    import xrft_amd as xa
    from oracle import xrft_oracle as o

    rng = np.random.default_rng(7)
    v = rng.standard_normal((nt, ny, nx)).astype(np.float32)
    v += (0.01 * np.arange(ny, dtype=np.float32))[None, :, None] + (-0.02 * np.arange(nx, dtype=np.float32) + 3)[None, None, :]
    v *= (1 + np.arange(nt, dtype=np.float32))[:, None, None]
    c = {"t": np.arange(nt), "y": np.arange(ny) * 1.0, "x": np.arange(nx) * 1.0}
    got = xa.isotropic_power_spectrum(xa.DataArray(v, ("t", "y", "x"), c), dim=["y", "x"], detrend=det, window=win, truncate=truncate)
    assert any("[fast" in p.describe() for p in api._plan_cache.values())
    if max(ny, nx) <= 256:
        assert any("[fasts]" in p.describe() for p in api._plan_cache.values())
    ref = o.isotropic_power_spectrum(o.OArr(v, ("t", "y", "x"), c), dim=["y", "x"], detrend=det, window=win, truncate=truncate)
    cases.check(got, ref, 3e-4)
    api._plan_cache.clear()


def _p2_fields(ny, nx, nt, seed, dx=1.0, x0=0.0):
    import xrft_amd as xa
    from oracle import xrft_oracle as o

    rng = np.random.default_rng(seed)
    v = rng.standard_normal((nt, ny, nx)).astype(np.float32)
    v += (0.01 * np.arange(ny, dtype=np.float32))[None, :, None] + (-0.02 * np.arange(nx, dtype=np.float32) + 3)[None, None, :]
    v *= (1 + np.arange(nt, dtype=np.float32))[:, None, None]
    c = {"t": np.arange(nt), "y": np.arange(ny) * dx + x0, "x": np.arange(nx) * 2 * dx - x0}
    return xa.DataArray(v, ("t", "y", "x"), c), o.OArr(v, ("t", "y", "x"), c)


@pytest.mark.parametrize("ny,nx,kw", [
    (1024, 1024, dict(detrend="linear", window="hann")),                      # true_phase=True: ifftshift sign + phase tables
    (1024, 2048, dict(true_phase=False, detrend="constant")),
    (2048, 1024, dict(shift=False, window="hamming", true_phase=True)),
    (256, 512, dict(detrend="linear", window="hann")),
    (512, 256, dict(true_phase=False)),
    # one slab per workgroup (csrc/fasts.h): complex rows staged in halves, the mirror rows conjugated
    (256, 256, dict(detrend="linear", window="hann")),
    (256, 256, dict(true_phase=False, shift=False)),
    (128, 128, dict(detrend="linear", window="hann")),
    (64, 64, dict(true_phase=False, detrend="constant")),
    (128, 256, dict(shift=False, window="hamming")),
    (256, 64, dict(detrend="linear", window="hann")),
    (64, 128, dict(true_phase=False)),
])
def test_fastp2_complex_fft(ny, nx, kw):
    """xrft.fft of a real float32 slab through the specialised kernels (complex result, Hermitian half mirrored)."""
    import xrft_amd as xa
    from oracle import xrft_oracle as o

    da, od = _p2_fields(ny, nx, 2, 11, x0=3.0)
    got = xa.fft(da, dim=["y", "x"], **kw)
    assert any("[fast" in p.describe() for p in api._plan_cache.values())
    if max(ny, nx) <= 256:
        assert any("[fasts]" in p.describe() for p in api._plan_cache.values())
    cases.check(got, o.fft(od, dim=["y", "x"], **kw), 3e-4)
    api._plan_cache.clear()


@pytest.mark.parametrize("ny,nx,kw", [(256, 512, dict(window="hann", detrend="linear")), (512, 256, dict(true_phase=False)),
                                       (256, 256, dict(real_dim="x", dim=["y"]))])
def test_fastp2_cross_phase(ny, nx, kw):
    """cross_phase: the specialised cross pipeline with the angle taken in the untile pass."""
    import xrft_amd as xa
    from oracle import xrft_oracle as o

    da, od = _p2_fields(ny, nx, 2, 41, x0=2.0)
    db, ob = _p2_fields(ny, nx, 2, 42, x0=-1.0)
    kw = dict(kw)
    dim = kw.pop("dim", ["y", "x"])
    got = xa.cross_phase(da, db, dim=dim, **kw)
    assert any("[fast" in p.describe() for p in api._plan_cache.values())
    ref = o.cross_phase(od, ob, dim=dim, **kw)
    assert got.dims == ref.dims and got.values.shape == ref.values.shape
    d = np.angle(np.exp(1j * (got.values.astype(np.float64) - ref.values)))
    # the angle of a near-zero cross spectrum amplifies rounding: compare where the cross spectrum is not tiny
    cs = np.abs(o.cross_spectrum(od, ob, dim=dim, **kw).values)
    ok = cs > 1e-3 * cs.max()
    assert np.abs(d[ok]).max() < 5e-3, np.abs(d[ok]).max()
    api._plan_cache.clear()


@pytest.mark.parametrize("ny,nx", [(1024, 1024), (1024, 2048), (256, 512)])
def test_fastp2_real_dim(ny, nx):
    """real_dim: the half spectrum leaves the specialised kernels as it is (no mirror), kept bins count twice."""
    import xrft_amd as xa
    from oracle import xrft_oracle as o

    da, od = _p2_fields(ny, nx, 2, 31, x0=1.5)
    db, ob = _p2_fields(ny, nx, 2, 32, x0=1.5)
    for fn, ofn, args, oargs in (
            (xa.power_spectrum, o.power_spectrum, (da,), (od,)),
            (xa.fft, o.fft, (da,), (od,)),
            (xa.cross_spectrum, o.cross_spectrum, (da, db), (od, ob))):
        for kw in (dict(detrend="linear", window="hann"), dict()):
            got = fn(*args, dim=["y"], real_dim="x", **kw)
            assert any("[fast" in p.describe() for p in api._plan_cache.values())
            cases.check(got, ofn(*oargs, dim=["y"], real_dim="x", **kw), 3e-4)
            api._plan_cache.clear()


@pytest.mark.parametrize("ny,nx,kw", [
    (1024, 1024, dict(detrend="linear", window="hann")),
    (2048, 1024, dict(true_phase=False)),
    (256, 256, dict(detrend="linear", window="hann")),
    (512, 512, dict()),
])
def test_fastp2_cross_spectrum(ny, nx, kw):
    import xrft_amd as xa
    from oracle import xrft_oracle as o

    da, od = _p2_fields(ny, nx, 2, 12)
    db, ob = _p2_fields(ny, nx, 2, 13)
    got = xa.cross_spectrum(da, db, dim=["y", "x"], **kw)
    assert any("[fast" in p.describe() for p in api._plan_cache.values())
    cases.check(got, o.cross_spectrum(od, ob, dim=["y", "x"], **kw), 3e-4)
    api._plan_cache.clear()


@pytest.mark.parametrize("ny,nx,kw", [
    (1024, 1024, dict(detrend="linear", window="hann", truncate=True)),
    (1024, 2048, dict(window="hann", truncate=False)),
    (256, 512, dict(detrend="linear", window="hann", truncate=True)),
    (512, 512, dict(truncate=True)),
])
def test_fastp2_isotropic_cross(ny, nx, kw):
    import xrft_amd as xa
    from oracle import xrft_oracle as o

    da, od = _p2_fields(ny, nx, 2, 14)
    db, ob = _p2_fields(ny, nx, 2, 15)
    got = xa.isotropic_cross_spectrum(da, db, dim=["y", "x"], **kw)
    assert any("[fast" in p.describe() for p in api._plan_cache.values())
    cases.check(got, o.isotropic_cross_spectrum(od, ob, dim=["y", "x"], **kw), 3e-4)
    api._plan_cache.clear()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_composite_radix_lengths(dtype):
    cases.run_composite_lengths(dtype)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_bluestein_lengths(dtype):
    cases.run_bluestein_cases(dtype)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_inverse_transforms(dtype):
    cases.run_inverse_cases(dtype)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_cross_phase(dtype):
    cases.run_cross_phase_cases(dtype)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_chunks_to_segments(dtype):
    cases.run_segment_cases(dtype)


def test_concurrent_callers_share_a_plan():
    """The API functions are pure, like the reference's: several threads may call them at once (dask-style workers).  The
    plan cache is locked and one plan enqueues one call at a time with a workspace per stream."""
    import threading

    import xrft_amd as xa

    rng = np.random.default_rng(3)
    c = {"t": np.arange(2), "y": np.arange(24) * 1.0, "x": np.arange(40) * 0.5}
    inputs = [rng.standard_normal((2, 24, 40)) for _ in range(4)]
    want = [xa.power_spectrum(xa.DataArray(v, ("t", "y", "x"), c), dim=["y", "x"], detrend="linear", window="hann").values for v in inputs]
    got = [None] * 4
    errs = []

    def work(i):
        try:
            for _ in range(5):
                got[i] = xa.power_spectrum(xa.DataArray(inputs[i], ("t", "y", "x"), c), dim=["y", "x"], detrend="linear", window="hann").values
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    assert not errs
    for g_, w_ in zip(got, want):
        np.testing.assert_allclose(g_, w_, rtol=1e-12)  # (the detrend sums are accumulated with atomics: order varies)


@pytest.mark.parametrize("dtype", ["float64", "float32", "complex128"])
def test_inplace_middle_and_first_axis(dtype):
    """Single-axis transforms along a middle / the first axis take the XRFTHIP_AXIS_Y plan (no transposed copy): fft, power
    and cross spectra with per-column detrend and window, against the oracle."""
    import xrft_amd as xa
    from oracle import xrft_oracle as o

    rng = np.random.default_rng(1)
    v = rng.standard_normal((6, 20, 12))
    if dtype.startswith("complex"):
        v = v + 1j * rng.standard_normal(v.shape)
    v = v.astype(dtype)
    c = {"t": np.arange(6) * 2.0, "y": np.arange(20) * 0.5, "x": np.arange(12) * 0.25 + 1}
    da, od = cases.pair(v, ("t", "y", "x"), c)
    tol = 1e-5 if dtype == "float32" else 1e-10
    for dim in ("y", "t"):
        for det, win in ((None, None), ("linear", "hann"), ("constant", None)):
            cases.check(xa.fft(da, dim=[dim], detrend=det, window=win), o.fft(od, dim=[dim], detrend=det, window=win), tol)
            assert "y:col-only" in next(reversed(api._plan_cache.values())).describe()
            cases.check(xa.power_spectrum(da, dim=[dim], detrend=det, window=win), o.power_spectrum(od, dim=[dim], detrend=det, window=win), tol)
    da2, od2 = cases.pair((v * 2 + 1).astype(dtype), ("t", "y", "x"), c)
    cases.check(xa.cross_spectrum(da, da2, dim=["y"], window="hann"), o.cross_spectrum(od, od2, dim=["y"], window="hann"), tol)
    cases.check(xa.fft(da, dim=["y"], true_phase=False, shift=False), o.fft(od, dim=["y"], true_phase=False, shift=False), tol)


def test_torch_conj_and_neg_views_are_resolved():
    """torch's lazy conjugate / negative bits are materialised before the raw pointer reaches the library (x.conj(), x.mH)."""
    import torch

    import xrft_amd as xa

    rng = np.random.default_rng(2)
    z = rng.standard_normal((3, 16, 24)) + 1j * rng.standard_normal((3, 16, 24))
    c = {"t": np.arange(3), "y": np.arange(16.0), "x": np.arange(24.0)}
    tz = torch.from_numpy(z)
    for view, ref in ((torch.conj(tz), np.conj(z)), (torch.conj(tz).imag, -z.imag), (-tz.real, -z.real)):
        got = xa.fft(xa.DataArray(view, ("t", "y", "x"), c), dim=["y", "x"], true_phase=False, true_amplitude=False, shift=False)
        want = np.fft.fftn(ref, axes=(1, 2))
        assert np.abs(got.values - want).max() / np.abs(want).max() < 1e-12


def test_fftmod_backend_object():
    """xrft_amd.fftmod -- the module the reference's `_fft_module` seam can return (xrft.py:32-36): fftn / rfftn / ifftn /
    irfftn / fftshift / ifftshift with the reference's call shapes against numpy.fft."""
    import fftmod_cases

    from xrft_amd import fftmod

    assert fftmod_cases.run_all(fftmod) < 2e-5


@pytest.mark.parametrize("n", [4096, 8192, 16384, 32768, 65536, 262144])
def test_fourstep_1d_fast_path(n):
    cases.run_fourstep_1d(n, nt=2)


@pytest.mark.parametrize("shape,full,dtype", [((2, 360, 360), True, "float64"), ((1, 1440, 720), False, "float64"), ((1, 720, 1440), False, "float64"),
                                               ((2, 360, 360), True, "float32"), ((1, 720, 1440), False, "float32"), ((1, 1440, 720), False, "float32"),
                                               ((2, 180, 360), True, "float64"), ((2, 240, 480), False, "float32"), ((1, 960, 480), False, "float64"),
                                               ((2, 256, 256), True, "float64"), ((1, 1024, 512), False, "float64"), ((1, 480, 960), False, "float32"),
                                               ((1, 1000, 1000), False, "float32"), ((1, 500, 1200), False, "float64")])
def test_fastm_latlon_lengths(shape, full, dtype):
    """The mixed-radix y-first kernels (csrc/fastm.h; BASELINE.json configs[4] is (64, 1440, 720) float64)."""
    cases.run_fastm_cases(shape, full, True, dtype)


def test_radial_sums_any_nbins_and_bit_identical_repeats():
    """Stand-alone and generic-plan radial sums: > 4096 bins, values vs numpy / the oracle, repeats bit for bit (the emulator runs
    the workgroups of a launch on several OS threads, so an order-dependent sum would show)."""
    cases.run_radial_sum_cases()


@pytest.mark.parametrize("shape,dtype", [((3, 360, 40), "float64"), ((2, 256, 24), "float32"), ((2, 1024, 16), "float64"), ((1, 2048, 8), "float32"),
                                         ((2, 1440, 8), "float64"), ((2, 240, 32), "float32"), ((3, 100, 16), "float64"), ((2, 1000, 8), "float32"),
                                         ((2, 128, 40), "float32"), ((2, 1200, 8), "float64"), ((2, 500, 24), "float64"), ((2, 800, 16), "float32"),
                                         ((1, 4096, 8), "float32"), ((1, 2048, 4), "float64")])
def test_one_axis_not_contiguous_fast_kernel(shape, dtype):
    """fastm_yonly_kernel (csrc/fastm.h): fft / power_spectrum along a middle or first axis, in place in memory order."""
    cases.run_yonly_fast_cases(shape, dtype)


@pytest.mark.parametrize("shape,dtype", [((5, 360), "float64"), ((3, 7, 256), "float32"), ((9, 1000), "float32"), ((2, 1440), "float64"), ((1, 2048), "float32"),
                                         ((6, 100), "float64"), ((4, 128), "float32"), ((3, 1200), "float64"), ((3, 4096), "float32"), ((2, 2048), "float64"),
                                         ((5, 512), "float32"), ((33, 1024), "float32")])
def test_short_contiguous_axis_fast_kernel(shape, dtype):
    """fastm_xonly_kernel (csrc/fastm.h): fft / power_spectrum along the last axis, rows packed in pairs."""
    cases.run_xonly_fast_cases(shape, dtype)


@pytest.mark.parametrize("n", [256, 360])
def test_adversarial_detrend_float32(n):
    """Outliers in the rows the float32 kernels estimate the trend from, offsets / trends far above the signal, constant
    columns (fasty.h at 256, fastm.h float32 at 360): every norm of cases.check, per bin down to 1e-6 of the peak."""
    cases.run_adversarial_detrend(n)


def test_reduce_axis_kernel():
    cases.run_reduce_axis_cases()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_two_adjacent_axes_that_are_not_the_trailing_ones(dtype):
    cases.run_inner_layout_cases(dtype)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_inner_layout_as_two_fused_passes(dtype):
    cases.run_fused_inner_cases(dtype)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_two_axes_that_are_not_adjacent_as_two_fused_passes(dtype):
    cases.run_fused_mid_cases(dtype)


def test_real_dim_along_one_axis_is_a_half_output_plan_or_a_refusal():
    """ABI 0.1.4: XRFTHIP_AXIS_Y with HALF_X (and REALDIM_X2) = real_dim along the ONE transformed axis, served by the one-pass kernels only.  A shift, a flip,
    complex input or REALDIM_X2 without HALF_X is a bad descriptor; a length no one-pass kernel takes is XRFTHIP_UNSUPPORTED_LENGTH -- and the API then
    transposes, with the reference's numbers either way."""
    import xrft_amd as xa
    from oracle import xrft_oracle as o
    from xrft_amd import engine

    def make(**kw):
        base = dict(ndim=2, batch=2, ny=96, nx=8, dtype=torch.float64, out_mode=_lib.OUT_POWER, detrend=0, flags=_lib.AXIS_Y | _lib.HALF_X, scale=1.0)
        base.update(kw)
        return engine.SpectralPlan(**base)

    p = make()
    assert (p.ny_out, p.nx_out) == (49, 8) and "[fastg y-only]" in p.describe()
    for bad in (dict(flags=_lib.AXIS_Y | _lib.HALF_X | _lib.SHIFT_Y), dict(flags=_lib.AXIS_Y | _lib.REALDIM_X2), dict(dtype=torch.complex128, out_mode=_lib.OUT_COMPLEX),
                dict(flags=_lib.AXIS_Y | _lib.HALF_X | _lib.FLIP_Y, out_mode=_lib.OUT_COMPLEX)):
        with pytest.raises(_lib.XrftHipError) as ei:
            make(**bad)
        assert ei.value.status == _lib.BAD_ARG, bad
    with pytest.raises(_lib.XrftHipError) as ei:
        make(ny=10007)  # (a prime beyond the tile: no one-pass kernel)
    assert ei.value.status == _lib.UNSUPPORTED_LENGTH
    rng = np.random.default_rng(3)
    for n, desc in ((96, False), (10007, False), (96, True)):
        v = rng.standard_normal((n, 2, 2))
        t = np.arange(n) * 0.5
        da = xa.DataArray(v, ("t", "y", "x"), {"t": t[::-1].copy() if desc else t})
        od = o.OArr(v, ("t", "y", "x"), {"t": t[::-1].copy() if desc else t})
        cases.check(xa.power_spectrum(da, dim="t", real_dim="t", detrend="constant"), o.power_spectrum(od, dim="t", real_dim="t", detrend="constant"), 1e-10)
        cases.check(xa.fft(da, dim="t", real_dim="t"), o.fft(od, dim="t", real_dim="t"), 1e-10)


def test_huge_slab_along_a_first_axis_takes_the_transposing_path():
    """a cube whose [n][inner] slab exceeds 2^31 elements cannot be indexed by the one-axis plans; plan creation is
    refused (XRFTHIP_BAD_ARG) and the API must not hand that error to the caller -- it never asks for such a plan."""
    import ctypes as C

    dll = _lib.load()
    for batch, ny, nx in ((1, 4096, 2048 * 2048), (1, 365, 1440 * 721 * 8), (2, 30, 1 << 27)):
        d = _lib.Desc(C.sizeof(_lib.Desc), 2, batch, ny, nx, _lib.F32, _lib.OUT_COMPLEX, 0, _lib.AXIS_Y, 1.0, 0, 0, 1)
        h = C.c_void_p(0)
        assert dll.xrfthip_plan_create(C.byref(h), C.byref(d)) == -1
    src = open(os.path.join(os.path.dirname(api.__file__), "api.py")).read()
    assert "ny * nx > (1 << 31) - 1" in src  # _execute_axis_y returns None (-> _arrange) before asking for the plan


def test_detrend_inner_scratch_is_bounded_and_extents_are_checked():
    """the partial sums of xrfthip_detrend_inner are sized by the chunks a (batch, inner) pair can use (they were 257 chunks'
    worth always: 6.4 GB for detrend(da, 'time') on a 1440 x 720 grid, 103 GB at 4096^2), extents beyond the 32-bit positions are refused
    with BAD_ARG (never truncated), and the API takes the transposing path for them."""
    import torch

    dll = _lib.load()
    for inner in (1440 * 720, 2048 * 2048, 4096 * 4096):
        nws = dll.xrfthip_detrend_inner_workspace_bytes(_lib.F32, 1, inner)
        assert 0 < nws <= inner * 24 * 4 + 256, (inner, nws)   # <= 3 chunks of three float64 sums per element + the coefficients
    assert dll.xrfthip_detrend_inner_workspace_bytes(_lib.F32, 1, 16) == ((16 * 24 * 257 + 255) // 256) * 256  # a short inner extent keeps its 256 chunks
    x = torch.zeros(16)
    assert dll.xrfthip_detrend_inner(_lib.F32, 1, 1, 1, 1000, (1 << 30) + 1, _lib.DETREND_LINEAR, x.data_ptr(), x.data_ptr(), x.data_ptr(), 1 << 62, None) == -1
    from xrft_amd import engine

    class _Fake:  # a tensor-like with the extents only
        dtype = torch.float32
        shape = (1000, (1 << 30) + 1)
        def is_contiguous(self): return True
    assert engine.detrend_inner(_Fake(), 0, 1, _lib.DETREND_LINEAR) is None
    # a long inner extent with few samples per element still runs where it lies and matches numpy
    rng = np.random.default_rng(5)
    v = rng.standard_normal((6, 40, 70)) + np.arange(6)[:, None, None] * 0.3
    got = api.detrend(api.DataArray(v, ("t", "y", "x"), {}), "t", "linear")
    import scipy.signal as sps
    assert np.abs(np.asarray(got.values) - sps.detrend(v, axis=0)).max() < 1e-12


def test_fused_radial_sums_compact_and_full_bin_codes():
    cases.run_fused_radial_code_forms(256)
    cases.run_fused_radial_code_forms(512)


@pytest.mark.parametrize("ny,nx,dtype", [(360, 240, "float64"), (240, 480, "float32")])
def test_fastm_radial_sums_gather_and_tables(ny, nx, dtype):
    cases.run_fastm_radial_code_forms(ny, nx, dtype)


def test_nan_poisons_its_own_slab_only():
    cases.run_nan_in_isotropic_spectra()


@pytest.mark.parametrize("shape,full,dtype", [((1, 900, 900), False, "float64"), ((1, 2000, 1500), False, "float32"), ((1, 1800, 900), False, "float64"), ((1, 900, 2000), False, "float64"),
                                               ((1, 3000, 900), False, "float32"), ((1, 1500, 3600), False, "float32")])
def test_fastm_round3_lengths(shape, full, dtype):
    """900, 1500, 1800, 2000 (both precisions; 2000 = 10 x 10 x 20, the radix-20 Good-Thomas butterfly), 3000 and 3600 (float32)."""
    cases.run_fastm_cases(shape, full, True, dtype)


@pytest.mark.parametrize("shape,full,dtype", [((1, 1080, 540), False, "float64"), ((1, 640, 320), True, "float32"), ((1, 1280, 640), False, "float64"), ((1, 2160, 1080), False, "float32"),
                                               ((1, 2160, 540), False, "float64"), ((1, 2560, 1280), False, "float32"), ((1, 2880, 1440), False, "float32"), ((1, 2160, 4320), False, "float32"),
                                               ((1, 2000, 1000), False, "float32"), ((1, 1800, 960), False, "float32"), ((1, 2160, 900), False, "float32"),
                                               ((1, 768, 384), False, "float64"), ((1, 1536, 768), False, "float32"), ((1, 1600, 1600), False, "float32"), ((1, 1920, 1080), False, "float64"),
                                               ((1, 1080, 1920), False, "float32"), ((1, 2400, 1200), False, "float32"), ((1, 3072, 1536), False, "float32"), ((1, 2160, 3840), False, "float32"), ((2, 192, 384), True, "float64")])
def test_fastm_grid_lengths(shape, full, dtype):
    """Gaussian grids (320 x 160 ... 2560 x 1280) and the 1/3 ... 1/12-degree lat/lon grids (1080 x 540 ... 4320 x 2160; 4320 = 15 x 16 x 18,
    the radix-18 Good-Thomas butterfly; 2560, 2880, 4320 in float32 only).  1800 / 2000 / 2160 rows in float32: pass 1 with four sequences per workgroup when
    the row length divides into 8-column blocks (1000, 960, 1080, 4320), with two otherwise (900)."""
    cases.run_fastm_cases(shape, full, True, dtype)


@pytest.mark.parametrize("shape,dtype", [((1, 2160, 8), "float64"), ((2, 4320, 8), "float32"), ((2, 540, 16), "float32"), ((1, 1280, 8), "float64"), ((1, 1920, 8), "float64"), ((2, 3840, 8), "float32"), ((2, 768, 16), "float32"), ((1, 1536, 8), "float32")])
def test_one_axis_grid_lengths(shape, dtype):
    cases.run_yonly_fast_cases(shape, dtype)


@pytest.mark.parametrize("shape,dtype", [((3, 2160), "float32"), ((2, 4320), "float32"), ((2, 1080), "float64"), ((3, 320), "float32"), ((3, 3840), "float32"), ((2, 1920), "float64"), ((3, 384), "float32"), ((2, 1600), "float32")])
def test_short_axis_grid_lengths(shape, dtype):
    cases.run_xonly_fast_cases(shape, dtype)


@pytest.mark.parametrize("shape,dtype", [((1, 2000, 8), "float64"), ((2, 3000, 8), "float32"), ((1, 1800, 4), "float64"), ((1, 3600, 8), "float32"), ((2, 900, 16), "float32"), ((1, 1500, 8), "float64")])
def test_one_axis_round3_lengths(shape, dtype):
    cases.run_yonly_fast_cases(shape, dtype)


@pytest.mark.parametrize("shape,dtype", [((3, 2000), "float32"), ((2, 3000), "float32"), ((2, 1800), "float64"), ((3, 3600), "float32"), ((2, 900), "float64"), ((3, 1500), "float32")])
def test_short_axis_round3_lengths(shape, dtype):
    cases.run_xonly_fast_cases(shape, dtype)


@pytest.mark.parametrize("shape,dtype,full", [((3, 50, 50), "float32", True), ((2, 50, 50), "float64", True), ((2, 27, 96), "float32", True), ((1, 100, 100), "float64", False),
                                              ((2, 45, 30), "float64", True), ((1, 96, 96), "float32", False), ((1, 120, 60), "float32", False),
                                              ((3, 45, 45), "float32", True), ((2, 50, 75), "float64", True), ((2, 27, 81), "float32", True), ((1, 125, 125), "float32", False)])
def test_small_slabs_of_any_smooth_shape_in_one_pass(shape, dtype, full):
    """fastg.h: lengths as data (run-time radices), both precisions; (2, 27, 96) is the odd-ny true-phase case the random sweep found; an odd nx
    (45 x 45, 75-sample rows, 125 x 125) runs its rows as complex sequences, the whole spectrum in the tile."""
    cases.run_fastg_cases(shape, dtype, full)


@pytest.mark.parametrize("shape,dtype", [((3, 96, 40), "float32"), ((2, 250, 36), "float64"), ((2, 45, 22), "float32"), ((1, 1250, 8), "float32"), ((2, 120, 50), "float64"),
                                         ((30, 48, 6), "float32"), ((2, 27, 130), "float64"), ((4, 150, 2), "float32"),
                                         # a prime factor with no butterfly: Bluestein inside the tile (365 = 5 x 73 days, 730, 77 = 7 x 11, 131)
                                         ((2, 365, 20), "float32"), ((1, 730, 10), "float64"), ((3, 77, 34), "float64"), ((2, 131, 18), "float32"), ((1, 262, 10), "float64"),
                                         # round 5: ONE prime 17 ... 127 with a smooth p - 1 (73: a year of days; 61: a leap year; 29: radix 7 in the inverse passes; 97 alone): the
                                         # prime-factor form with Rader's algorithm along the prime
                                         ((2, 366, 12), "float64"), ((2, 97, 10), "float32"), ((1, 1460, 6), "float32"), ((2, 58, 14), "float64")])
def test_one_axis_not_contiguous_any_smooth_length(shape, dtype):
    """fastg.h, fastgy_kernel: `dim="time"` calls on lengths outside the mixed-radix table."""
    cases.run_yonly_any_length_cases(shape, dtype)
    d = next(reversed(api._plan_cache.values())).describe()
    n = shape[1]
    assert ("Rader" in d) == (n in (365, 730, 366, 97, 1460, 58)) and ("Bluestein" in d) == (n in (131, 262)), d


@pytest.mark.parametrize("shape,dtype", [((37, 250), "float32"), ((5, 96), "float64"), ((3, 4, 125), "float32"), ((2, 750), "float64"), ((300, 50), "float32"), ((2, 2250), "float32"),
                                         ((7, 243), "float64"), ((1, 1250), "float32")])
def test_last_axis_any_smooth_length(shape, dtype):
    """fastg.h on groups of rows: 1-D spectra along the contiguous axis on lengths outside the tables."""
    cases.run_rows_any_length_cases(shape, dtype)


@pytest.mark.parametrize("shape,dtype", [((37, 365), "float32"), ((5, 730), "float64"), ((3, 4, 146), "float32"), ((2, 1460), "float64"), ((9, 97), "float64"), ((33, 58), "float32")])
def test_last_axis_with_one_awkward_prime(shape, dtype):
    """fastg.h, fastgy_kernel FORM 3: 1-D spectra along the contiguous axis on 365 / 730 / 1460 / 146 / 97 / 58-sample rows (Rader's algorithm along the prime)."""
    cases.run_rows_rader_cases(shape, dtype)


@pytest.mark.parametrize("shape,dtype", [((2, 360, 250), "float64"), ((1, 300, 512), "float32"), ((2, 243, 125), "float32"), ((3, 50, 50), "float64"),
                                         ((1, 1440, 720), "float64"), ((2, 360, 240), "float32")])
def test_inverse_transforms_on_the_one_pass_kernels(shape, dtype):
    """xrft.ifft over two axes as two one-pass stages, over one axis where it lies, small slabs in one pass (csrc/fastg.h)."""
    cases.run_inverse_one_pass_cases(shape, dtype)


@pytest.mark.parametrize("axis", [0, 1])
def test_ifft_with_a_permuted_frequency_coordinate_on_a_non_last_axis(axis):
    """xrft.ifft sorts by the coordinate first (xrft.py:598), so an arbitrarily permuted frequency axis is legal; on a first / middle
    axis the index map is an array, which the one-axis fast path must not compare with a string."""
    import warnings

    import xrft_amd as xa

    rng = np.random.default_rng(5)
    shape = (6, 10, 8)
    dims = ("t", "y", "x")
    d = dims[axis]
    v = rng.standard_normal(shape)
    c = {"t": np.arange(6) * 2.0, "y": np.arange(10) * 0.5 - 1.0, "x": np.arange(8) * 0.25}
    da, _od = cases.pair(v, dims, c)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        F = xa.fft(da, dim=[d])
        perm = rng.permutation(shape[axis])
        assert not np.array_equal(np.sort(perm), perm) and not np.array_equal(perm[::-1], np.arange(len(perm)))
        fd = "freq_" + d
        idx = [slice(None)] * 3
        idx[axis] = perm
        fc = {k: np.asarray(F[k].values) for k in F.dims}
        fc[fd] = fc[fd][perm]
        lagattr = dict(F[fd].attrs)
        Fp = xa.DataArray(np.ascontiguousarray(np.asarray(F.values)[tuple(idx)]), F.dims, {k: xa.Coordinate((k,), fc[k], lagattr if k == fd else {}, k) for k in F.dims})
        back = xa.ifft(Fp, dim=[fd])
        ref = xa.ifft(F, dim=[fd])  # the same spectrum in sorted order
        assert np.abs(np.asarray(back.values) - np.asarray(ref.values)).max() < 1e-12
        assert np.abs(np.asarray(back.values).real - v).max() < 1e-10
        assert np.array_equal(back[d].values, ref[d].values)


@pytest.mark.parametrize("shape,dtype", [((2, 77, 90), "float64"), ((2, 63, 55), "float32"), ((1, 46, 60), "float32"), ((1, 243, 50), "float64"), ((2, 180, 84), "float64"),
                                         ((1, 94, 60), "float32"), ((2, 146, 44), "float64"), ((1, 206, 40), "float32"), ((1, 206, 40), "float64")])
def test_two_pass_pipeline_with_the_lengths_as_data(shape, dtype, monkeypatch):
    """csrc/fastn.h on the emulator (small slabs kept off the one-pass kernel): run-time radices incl. 7 / 11, odd lengths, the chirp convolution (94 = 2 x 47),
    4 passes (243), a table length on one side (180); the Rader columns (46 = 2 x 23, 146 = 2 x 73, 206 = 2 x 103: the 17-point butterfly)."""
    monkeypatch.setenv("XRFTHIP_FASTG", "0")
    cases.run_fastn_cases(shape, dtype)
    d = " ".join(p.describe() for p in api._plan_cache.values())
    assert ("Rader" in d) == (shape[1] in (46, 146, 206)) and ("chirp" in d) == (shape[1] == 94), d


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_two_transform_axes_that_are_not_adjacent(dtype):
    """xrfthip_desc.mid: dim = ["t", "x"] of (t, y, x) where the axes lie."""
    cases.run_mid_layout_cases(dtype)


@pytest.mark.parametrize("n", [256, 1024, 2048, 4096, 8192, 16384, 65536, 262144])
def test_complex_rows_in_one_pass(n):
    """csrc/fasty_c2c.h (rows of 256 .. 4096 points: the row pass on the input's own rows) and csrc/fastr.h fastc_kernel (8192, 16384) on the emulator: fft / ifft / power
    spectrum of complex64 rows against the oracle."""
    cases.run_complex_rows_cases(n, nt=2)


@pytest.mark.parametrize("ny,nx,variant", [(256, 256, 0), (512, 256, 1), (256, 1024, 2), (1024, 512, 3), (2048, 256, 0), (256, 4096, 1), (4096, 256, 2)])
def test_complex_slabs_through_the_two_pass_pipeline(ny, nx, variant):
    """csrc/fasty_c2c.h on the emulator: fft (with and without a window) / ifft / power spectrum of complex64 slabs, every W2 geometry (ny = 256 .. 4096)."""
    cases.run_complex_two_pass_cases(ny, nx, nt=2 if ny * nx <= (1 << 19) else 1, variant=variant)


@pytest.mark.parametrize("ny,nx,variant", [(256, 512, 0), (512, 1024, 1), (1024, 4096, 2), (2048, 512, 3), (4096, 1024, 1)])
def test_half_spectra_back_to_real_fields_through_the_two_pass_pipeline(ny, nx, variant):
    """csrc/fasty_c2c.h on the emulator: irfftn (the Nyquist column's extra block in pass 1, the c2r row pass) and irfft along the contiguous axis."""
    cases.run_c2r_two_pass_cases(ny, nx, nt=2 if ny * nx <= (1 << 20) else 1, variant=variant)


def test_small_slabs_walked_by_a_resident_set():
    """csrc/fasts.h: a resident set of workgroups with the next slab's loads in flight beside the stores (the default for long batches of 256 x 256 slabs)."""
    cases.run_small_slab_walk_cases(shapes=((5, 256, 256), (5, 128, 256), (7, 64, 64)))


def test_long_rows_walked_by_a_resident_set(monkeypatch):
    """csrc/fastr.h fastr2_kernel with a resident set of workgroups walking the rows (forced small here: the emulator runs one workgroup at a time)."""
    monkeypatch.setenv("XRFTHIP_FASTR_GRID", "2")
    monkeypatch.setenv("XRFTHIP_FASTR_STAGGER", "769")
    api._plan_cache.clear()
    cases.run_fourstep_1d(16384, nt=5)
    api._plan_cache.clear()


def test_caller_buffers_are_held_to_the_plan():
    """engine.SpectralPlan.execute with a caller's `out` / `iso` (graph capture, composed passes): a buffer of the wrong size, dtype or layout is a
    ValueError before the library sees its pointer (found with scripts/run_emu_asan.sh: a short buffer was a heap overflow in the kernel's stores)."""
    import torch

    from xrft_amd import engine

    plan = engine.SpectralPlan(2, 2, 64, 64, torch.float32, out_mode=_lib.OUT_POWER, detrend=_lib.DETREND_LINEAR)
    x = torch.randn(2, 64, 64)
    good = torch.empty(2, 64, 64)
    out, _ = plan.execute(x, out=good)
    assert out is good
    for bad in (torch.empty(2 * 64 * 64 - 64), torch.empty(2, 64, 64, dtype=torch.float64), torch.empty(2, 64, 128)[:, :, ::2]):
        with pytest.raises(ValueError):
            plan.execute(x, out=bad)


def test_inverse_transform_over_two_axes_that_are_not_the_trailing_pair():
    """xrft.ifft of (y, x, t) / (t, y, x) spectra over [y, x] / [t, x]: one axis at a time where the axes lie, no transposed copy."""
    cases.run_inverse_non_trailing_pairs()
