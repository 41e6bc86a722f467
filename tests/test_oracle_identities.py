"""
The reference's own test identities (closed-form numpy/scipy expressions, xrft/tests/test_xrft.py and
test_detrend.py) restated against the CPU oracle with fixed seeds.  This is the second leg of the oracle's
pinning: the reference has no golden files; its tests ARE these identities.
Each test cites the reference test it restates.
"""
import warnings

import numpy as np
import numpy.testing as npt
import pytest
import scipy.signal as sps

from oracle import xrft_oracle as o

warnings.simplefilter("ignore")
RNG = np.random.default_rng(1234)


def _da1d(kind):
    Nx = 16
    x = np.linspace(0, 1.0, Nx)
    coords = None if kind == "nocoords" else {"x": x}
    return o.OArr(RNG.random(Nx), ("x",), coords)


@pytest.mark.parametrize("kind", ["numpy", "nocoords"])
def test_fft_1d(kind):
    """test_xrft.py:58-97"""
    da = _da1d(kind)
    Nx = da.shape[0]
    dx = float(da.coord("x")[1] - da.coord("x")[0])
    ft = o.dft(da, detrend="constant")
    assert ft.dims == ("freq_x",)
    fx = np.fft.fftshift(np.fft.fftfreq(Nx, dx))
    npt.assert_allclose(ft.coord("freq_x"), fx)
    assert ft.coord_attrs["freq_x"]["spacing"] == fx[1] - fx[0]
    data = da.values - da.values.mean()
    npt.assert_allclose(np.fft.fftshift(np.fft.fft(data)), ft.values, atol=1e-14)
    ft = o.dft(da)
    npt.assert_allclose(np.fft.fftshift(np.fft.fft(da.values)), ft.values)
    ft = o.dft(da, detrend="linear")
    npt.assert_allclose(np.fft.fftshift(np.fft.fft(sps.detrend(da.values))), ft.values, atol=1e-14)
    if kind == "numpy":
        x = da.coord("x").copy()
        x[-1] *= 2
        with pytest.raises(ValueError):
            o.fft(o.OArr(da.values, ("x",), {"x": x}))


def test_fft_1d_time_datetime():
    """test_xrft.py:99-113 (pandas branch; cftime is not installed)"""
    import pandas as pd

    time = pd.date_range("2000-01-01", "2001-01-01", inclusive="left").values
    Nt = len(time)
    da = o.OArr(RNG.random(Nt), ("time",), {"time": time})
    ft = o.dft(da, shift=False)
    npt.assert_allclose(ft.coord("freq_time"), np.fft.fftfreq(Nt, 86400.0))


def test_fft_2d():
    """test_xrft.py:115-138"""
    N = 16
    da = o.OArr(RNG.random((N, N)), ("x", "y"), {"x": np.arange(N), "y": np.arange(N)})
    ft = o.dft(da, shift=False)
    npt.assert_almost_equal(ft.values, np.fft.fftn(da.values))
    ft = o.dft(da, shift=False, window="hann", detrend="constant")
    window = sps.windows.hann(N, sym=False) * sps.windows.hann(N, sym=False)[:, None]
    npt.assert_almost_equal(ft.values, np.fft.fftn((da.values - da.values.mean()) * window))
    da = o.OArr(RNG.random((N, N)), ("x", "y"), {"x": np.arange(N, 0, -1), "y": np.arange(N, 0, -1)})
    assert (o.power_spectrum(da, shift=False, density=True).values >= 0.0).all()


def test_dim_and_batched():
    """test_xrft.py:140-178"""
    N = 16
    da = o.OArr(RNG.random((N, N)), ("x", "y"), {"x": np.arange(N), "y": np.arange(N)})
    npt.assert_array_equal(o.fft(da, dim="y", shift=False).values, o.fft(da, dim=["y"], shift=False).values)
    assert o.fft(da, dim="y").dims == ("x", "freq_y")
    da = o.OArr(RNG.random((N, N, N)), ("time", "x", "y"),
                {"time": np.arange(N), "x": np.arange(N), "y": np.arange(N)})
    npt.assert_almost_equal(o.fft(da, dim=["x", "y"], shift=False).values, np.fft.fftn(da.values, axes=[1, 2]))
    daft = o.fft(da, dim=["time"], shift=False, detrend="linear")
    npt.assert_almost_equal(daft.values, np.fft.fftn(sps.detrend(da.values, axis=0), axes=[0]))


def test_fft_real():
    """test_xrft.py:214-270"""
    da = _da1d("numpy")
    Nx = da.shape[0]
    dx = float(da.coord("x")[1] - da.coord("x")[0])
    ft = o.dft(da, real_dim="x", detrend="constant")
    assert ft.dims == ("freq_x",)
    npt.assert_allclose(ft.coord("freq_x"), np.fft.rfftfreq(Nx, dx))
    npt.assert_allclose(np.fft.rfft(da.values - da.values.mean()), ft.values, atol=1e-14)
    with pytest.raises(ValueError):
        o.fft(da, real_dim="y", detrend="constant")
    Nx, Ny = 16, 32
    da = o.OArr(RNG.random((Nx, Ny)), ("x", "y"), {"x": np.arange(Nx), "y": np.arange(Ny)})
    daft = o.dft(da, real_dim="x")
    npt.assert_almost_equal(daft.values, np.fft.rfftn(da.values.T).T)
    npt.assert_almost_equal(daft.values, o.fft(da, dim=["y"], real_dim="x", true_phase=False,
                                               true_amplitude=False).values)
    npt.assert_almost_equal(daft.coord("freq_x"), np.fft.rfftfreq(Nx, 1.0))
    npt.assert_almost_equal(daft.coord("freq_y"), np.fft.fftfreq(Ny, 1.0))


def test_power_spectrum():
    """test_xrft.py:388-495 (numpy branch; segment part needs dask chunks and is out of scope for v1)"""
    N = 16
    da = o.OArr(RNG.random(N), ("x",), {"x": np.arange(N)})
    _, p_scipy = sps.periodogram(da.values, window="rectangular", return_onesided=True)
    ps = o.power_spectrum(da, dim="x", real_dim="x", detrend="constant")
    npt.assert_almost_equal(ps.values, p_scipy)

    t = np.array(["2019-04-18", "2019-04-19"], dtype="datetime64")
    da = o.OArr(RNG.random((2, N, N)), ("time", "y", "x"), {"time": t, "y": np.arange(N), "x": np.arange(N)})
    ps = o.power_spectrum(da, dim=["y", "x"], window="hann", density=False, detrend="constant")
    daft = o.fft(da, dim=["y", "x"], detrend="constant", window="hann")
    npt.assert_almost_equal(ps.values, np.real(daft.values * np.conj(daft.values)))
    assert np.isfinite(ps.values).all()

    ps = o.power_spectrum(da, dim=["y"], real_dim="x", window="hann", density=False, detrend="constant")
    daft = o.fft(da, dim=["y"], real_dim="x", detrend="constant", window="hann")
    f = np.full(daft.shape[-1], 2.0)
    f[0], f[-1] = 1.0, 1.0
    npt.assert_almost_equal(ps.values, np.real(daft.values * np.conj(daft.values)) * f)

    ps = o.power_spectrum(da, dim=["y", "x"], window="hann", detrend="constant")
    daft = o.fft(da, dim=["y", "x"], window="hann", detrend="constant")
    test = np.real(daft.values * np.conj(daft.values)) / N ** 4
    dk = np.diff(np.fft.fftfreq(N, 1.0))[0]
    npt.assert_almost_equal(ps.values, test / dk ** 2)

    ps = o.power_spectrum(da, dim=["y", "x"], window="hann", density=False, detrend="linear")
    daft = o.fft(da, dim=["y", "x"], window="hann", detrend="linear")
    npt.assert_almost_equal(ps.values, np.real(daft.values * np.conj(daft.values)))
    with pytest.raises(ValueError):
        o.power_spectrum(da, dim=["y", "x"], window=None, window_correction=True)


def test_cross_spectrum():
    """test_xrft.py:497-560"""
    N = 16
    dim = ["x", "y"]
    t = np.array(["2019-04-18", "2019-04-19"], dtype="datetime64")
    c = {"time": t, "x": np.arange(N), "y": np.arange(N)}
    da = o.OArr(RNG.random((2, N, N)), ("time", "x", "y"), c)
    da2 = o.OArr(RNG.random((2, N, N)), ("time", "x", "y"), c)
    daft = o.fft(da, dim=dim, shift=True, detrend="constant", window="hann")
    daft2 = o.fft(da2, dim=dim, shift=True, detrend="constant", window="hann")
    cs = o.cross_spectrum(da, da2, dim=dim, window="hann", density=False, detrend="constant")
    npt.assert_almost_equal(cs.values, daft.values * np.conj(daft2.values))
    cs = o.cross_spectrum(da, da2, dim=dim, shift=True, window="hann", detrend="constant")
    dk = np.diff(np.fft.fftfreq(N, 1.0))[0]
    test = (daft.values * np.conj(daft2.values)) / N ** 4 / dk ** 2
    npt.assert_almost_equal(cs.values, test)
    cs = o.cross_spectrum(da, da2, dim=dim, shift=True, window="hann", detrend="constant", window_correction=True)
    w = sps.windows.hann(N, sym=False)
    npt.assert_almost_equal(cs.values, test / (np.outer(w, w) ** 2).mean())
    with pytest.raises(ValueError):
        o.cross_spectrum(da, da2, dim=dim, window=None, window_correction=True)


def test_spectrum_dim():
    """test_xrft.py:562-603"""
    N = 16
    c = {"time": np.arange(2), "y": np.arange(N), "x": np.arange(N)}
    da = o.OArr(RNG.random((2, N, N)), ("time", "y", "x"), c)
    da2 = o.OArr(RNG.random((2, N, N)), ("time", "y", "x"), c)
    ps = o.power_spectrum(da, dim="y", real_dim="x", window="hann", detrend="constant")
    assert ps.dims == ("time", "freq_y", "freq_x")
    cs = o.cross_spectrum(da, da2, dim="y", shift=True, window="hann", detrend="constant")
    assert cs.dims == ("time", "freq_y", "x")


def test_parseval():
    """test_xrft.py:693-842 (chunks_to_segments=False branch)"""
    N = 16
    c = {"x": np.arange(N), "y": np.arange(N)}
    da = o.OArr(RNG.random((N, N)), ("x", "y"), c)
    da2 = o.OArr(RNG.random((N, N)), ("x", "y"), c)
    ps = o.power_spectrum(da)
    npt.assert_almost_equal(ps.values.mean(), (da.values ** 2).mean(), decimal=5)
    ps = o.power_spectrum(da, window="hann", detrend="constant")
    w = sps.windows.hann(N, sym=False)
    window = np.outer(w, w)
    dp = da.values - da.values.mean()
    npt.assert_almost_equal(ps.values.mean(), ((dp * window) ** 2).mean(), decimal=5)
    cs = o.cross_spectrum(da, da2, window="hann", detrend="constant")
    dp2 = da2.values - da2.values.mean()
    npt.assert_almost_equal(cs.values.mean(), ((dp * window) * (dp2 * window)).mean(), decimal=5)
    d3 = o.OArr(RNG.random((N, N, N)), ("time", "y", "x"), {"time": np.arange(N), "y": np.arange(N), "x": np.arange(N)})
    ps = o.power_spectrum(d3, dim=["x", "y"], window="hann", detrend="linear")
    det = o.detrend(d3, ["x", "y"], "linear").transpose("time", "y", "x").values[0]
    npt.assert_almost_equal(ps.values[0].mean(), ((det * window) ** 2).mean(), decimal=5)

    Nx = 40
    dx = 0.37
    x = dx * (np.arange(-Nx // 2, -Nx // 2 + Nx) + 7)
    s = o.OArr(RNG.random(Nx) + 1j * RNG.random(Nx), ("x",), {"x": x})
    F = o.dft(s, dim="x", true_phase=True, true_amplitude=True)
    npt.assert_almost_equal((np.abs(s.values) ** 2).sum() * dx,
                            (np.abs(F.values) ** 2).sum() * F.coord_attrs["freq_x"]["spacing"])
    Nx, Ny, dx, dy = 40, 60, 0.37, 0.81
    s = o.OArr(RNG.random((Nx, Ny)) + 1j * RNG.random((Nx, Ny)), ("x", "y"),
               {"x": dx * (np.arange(-Nx // 2, -Nx // 2 + Nx) - 3), "y": dy * (np.arange(-Ny // 2, -Ny // 2 + Ny) + 11)})
    F = o.dft(s, dim=("x", "y"), true_phase=True, true_amplitude=True)
    npt.assert_almost_equal((np.abs(s.values) ** 2).sum() * dx * dy,
                            (np.abs(F.values) ** 2).sum() * F.coord_attrs["freq_x"]["spacing"]
                            * F.coord_attrs["freq_y"]["spacing"])


@pytest.mark.parametrize("truncate", [False, True])
def test_isotropize(truncate):
    """test_xrft.py:942-992 (N reduced 512 -> 128 to keep the CPU suite fast)"""
    N = 128
    rng = np.random.default_rng(7)
    theta = o.synthetic_field(N, 1.0, 10.0, -3.0, rng)
    for extra in (None, 3):
        if extra:
            v = theta[None] + np.ones((extra, 1, 1))
            da = o.OArr(v, ("d0", "y", "x"), {"y": np.arange(N), "x": np.arange(N)})
        else:
            da = o.OArr(theta, ("y", "x"), {"y": np.arange(N), "x": np.arange(N)})
        ps = o.power_spectrum(da, spacing_tol=1e-3, dim=["x", "y"])
        iso = o.isotropize(ps, ["freq_x", "freq_y"], nfactor=4, truncate=truncate)
        assert [d for d in iso.dims if d != "d0"] == ["freq_r"]
        npt.assert_allclose(iso.values.sum(), ps.values.sum(), atol=1e-3)


def test_isotropic_ps_slope():
    """test_xrft.py:995-1031"""
    N, s = 512, -3.0
    rng = np.random.default_rng(11)
    theta = o.synthetic_field(N, 1.0, 10.0, s, rng)
    v = theta[None] + np.ones((2, 1, 1))
    da = o.OArr(v, ("d0", "y", "x"), {"y": np.arange(N), "x": np.arange(N)})
    iso = o.isotropic_power_spectrum(da, dim=["y", "x"], detrend="constant", density=True)
    m = iso.values.mean(axis=0)
    assert np.isfinite(m).all()
    _, a, _ = o.fit_loglog(iso.coord("freq_r")[:-35], m[:-35])
    npt.assert_allclose(a, s, atol=0.06)
    seq = np.stack([o.isotropic_power_spectrum(o.OArr(v[i], ("y", "x"), {"y": np.arange(N), "x": np.arange(N)}),
                                               detrend="constant", density=True).values for i in range(2)])
    npt.assert_almost_equal(m, seq.mean(axis=0))


def test_isotropic_errors_and_cs():
    """test_xrft.py:1034-1111"""
    c = {"time": np.arange(2), "z": np.arange(5), "zz": ("z", np.arange(5)), "y": np.arange(16), "x": np.arange(32)}
    da = o.OArr(RNG.random((2, 5, 16, 32)), ("time", "z", "y", "x"), c)
    da2 = o.OArr(RNG.random((2, 5, 16, 32)), ("time", "z", "y", "x"), c)
    with pytest.raises(ValueError):
        o.isotropic_power_spectrum(da, dim=["z", "y", "x"])
    iso = o.isotropic_power_spectrum(da, dim=["y", "x"])
    assert np.isfinite(iso.values).all() and iso.shape == (2, 5, 4)
    with pytest.raises(ValueError):
        o.isotropic_cross_spectrum(da, da2, dim=["z", "y", "x"])
    ics = o.isotropic_cross_spectrum(da, da2, dim=["y", "x"], window="hann")
    assert np.isfinite(ics.values).all() and np.iscomplexobj(ics.values)


def test_spacing_tol_and_errors():
    """test_xrft.py:1114-1135, 1315-1379"""
    Nx = 16
    x = np.linspace(0, 1.0, Nx)
    x[-1] += 0.001
    da3 = o.OArr(RNG.random(Nx), ("x",), {"x": x})
    o.fft(da3, spacing_tol=1e-1)
    with pytest.raises(ValueError):
        o.fft(da3, spacing_tol=1e-4)
    with pytest.raises(TypeError):
        o.fft(da3, spacing_tol="string")
    N = 20
    with pytest.raises(ValueError):
        o.dft(o.OArr(RNG.random(N) + 0j, ("freq_x",), {"freq_x": np.zeros(N)}))
    c = {"time": np.arange(2), "x": np.arange(16), "y": np.arange(16), "x_nondim": ("x", np.arange(16))}
    da = o.OArr(RNG.random((2, 16, 16)), ("time", "x", "y"), c)
    with pytest.raises(ValueError):
        o.power_spectrum(da)
    o.power_spectrum(da, dim=["time", "y"])
    da = o.OArr(RNG.random((2, 5, 3)), ("time", "x", "y"), {"time": np.arange(2), "x": np.arange(5),
                                                          "y": np.array(["a", "b", "c"])})
    with pytest.raises(ValueError):
        o.power_spectrum(da)
    o.power_spectrum(da, dim=["time", "x"])


def test_true_phase():
    """test_xrft.py:1191-1207, 1210-1250, 1336-1347"""
    f0, T, dx = 2.0, 4.0, 0.02
    x = np.arange(-8 * T, 5 * T + dx, dx)
    y = np.cos(2 * np.pi * f0 * x)
    y[np.abs(x) >= (T / 2.0)] = 0.0
    s = o.OArr(y, ("x",), {"x": x})
    lag = x[len(x) // 2]
    f = np.fft.fftfreq(len(x), dx)
    expected = np.fft.fft(np.fft.ifftshift(y)) * np.exp(-1j * 2.0 * np.pi * f * lag)
    out = o.dft(s, dim="x", true_phase=True, true_amplitude=False, shift=False, prefix="freq_")
    npt.assert_allclose(out.values, expected, rtol=1e-5, atol=1e-8)
    npt.assert_allclose(out.coord("freq_x"), f)

    # theoretical matching (dx coarsened 1e-4 -> 1e-3 to keep the CPU suite fast)
    dx = 1e-3
    x = np.arange(-6 * T, 5 * T, dx)
    y = np.cos(2.0 * np.pi * f0 * x)
    y[np.abs(x) >= (T / 2.0)] = 0.0
    S = o.dft(o.OArr(y, ("x",), {"x": x}), dim="x", true_phase=True, true_amplitude=True)
    fx = S.coord("freq_x")
    TF = T / 2 * (np.sinc(T * (fx - f0)) + np.sinc(T * (fx + f0)))
    npt.assert_allclose(S.values, TF.astype(complex), rtol=1e-8, atol=1e-2)

    # real transform is half the total transform
    Nx, dx = 40, 0.3
    s = o.OArr(RNG.random(Nx), ("x",), {"x": dx * (np.arange(-Nx // 2, -Nx // 2 + Nx) + 5)})
    s1 = o.dft(s, dim="x", true_phase=True, shift=True)
    s2 = o.dft(s, real_dim="x", true_phase=True, shift=True)
    half = np.conj(s1.values[: Nx // 2 + 1])
    fh = -s1.coord("freq_x")[: Nx // 2 + 1]
    order = np.argsort(fh)
    npt.assert_allclose(half[order], s2.values, atol=1e-12)
    npt.assert_allclose(fh[order], s2.coord("freq_x"), atol=1e-12)

    # reversed coordinates do not change the true-phase transform
    N = 20
    xr_ = np.arange(N // 2, -N // 2, -1) + 2
    v = RNG.random(N) + 1j * RNG.random(N)
    a = o.dft(o.OArr(v, ("x",), {"x": xr_}), dim="x", true_phase=True)
    idx = np.argsort(xr_)
    b = o.dft(o.OArr(v[idx], ("x",), {"x": xr_[idx]}), dim="x", true_phase=True)
    npt.assert_allclose(a.values, b.values, atol=1e-12)


def test_detrend_removes_injected_trend():
    """test_detrend.py:40-119"""
    rng = np.random.default_rng(3)
    noise = rng.standard_normal((3, 20, 30))
    da0 = o.OArr(noise, ("t", "y", "x"))
    base = o.detrend(da0, ["y", "x"], "linear").transpose("t", "y", "x").values
    ii, jj = np.meshgrid(np.arange(20), np.arange(30), indexing="ij")
    trended = base + 0.3 * ii - 0.2 * jj + 4
    out = o.detrend(o.OArr(trended, ("t", "y", "x")), ["y", "x"], "linear").transpose("t", "y", "x").values
    npt.assert_allclose(out, base, atol=1e-9)
    c = o.detrend(da0, ["y", "x"], "constant").values
    npt.assert_allclose(c.mean(axis=(1, 2)), 0, atol=1e-12)
    l1 = o.detrend(da0, "x", "linear").values
    npt.assert_allclose(l1, sps.detrend(noise, axis=2))
    with pytest.raises(NotImplementedError):
        o.detrend(da0, ["y", "x"], "quadratic")


def test_ifft_round_trips():
    """test_xrft.py:1253-1312"""
    rng = np.random.default_rng(21)
    N = 20
    s = o.OArr(rng.random(N) + 1j * rng.random(N), ("x",), {"x": np.arange(0, N)})
    for sh in (True, False):
        npt.assert_allclose(o.ifft(o.fft(s, shift=sh), shift=True).values, s.values, atol=1e-14)
    N, dx = 40, 0.37
    x = dx * (np.arange(-N // 2, -N // 2 + N) + 7)
    s = o.OArr(rng.random(N) + 1j * rng.random(N), ("x",), {"x": x})
    F = o.dft(s, true_phase=True, true_amplitude=True)
    for lag in (float(x[N // 2]), None):
        kw = {} if lag is None else {"lag": lag}
        back = o.idft(F, shift=True, true_phase=True, true_amplitude=True, **kw)
        npt.assert_allclose(back.values, s.values, atol=1e-13)
        npt.assert_allclose(back.coord("x"), x, atol=1e-12)
    with pytest.raises(ValueError):
        o.idft(o.OArr(rng.random(20) + 0j, ("freq_x",), {"freq_x": np.arange(-10, 10) + 2}))


def test_cross_phase_and_segments():
    """test_xrft.py:606-634, 273-337"""
    N = 32
    x = np.linspace(0, 1, num=N, endpoint=False)
    f, po = 6, np.pi / 2
    a = o.OArr(np.cos(2 * np.pi * f * x), ("x",), {"x": x}, name="a")
    b = o.OArr(np.cos(2 * np.pi * f * x - po), ("x",), {"x": x}, name="b")
    cp = o.cross_phase(a, b, dim=["x"])
    npt.assert_almost_equal(cp.values[np.argmin(np.abs(cp.coord("freq_x") - f))], po)
    assert cp.name == "a_b_phase"
    rng = np.random.default_rng(22)
    da = o.OArr(rng.random((N, N, N)), ("time", "y", "x"), {"time": np.arange(N), "y": np.arange(N), "x": np.arange(N)})
    ft = o.fft(da.chunk({"time": 16}), dim=["time"], shift=False, chunks_to_segments=True)
    assert ft.dims == ("time_segment", "freq_time", "y", "x")
    npt.assert_almost_equal(ft.values, np.fft.fftn(da.values.reshape((2, 16, N, N)), axes=[1]), decimal=7)
    ft = o.fft(da.chunk({"y": 16, "x": 16}), dim=["y", "x"], shift=False, chunks_to_segments=True)
    assert ft.dims == ("time", "y_segment", "freq_y", "x_segment", "freq_x")
    npt.assert_almost_equal(ft.values, np.fft.fftn(da.values.reshape((N, 2, 16, 2, 16)), axes=[2, 4]), decimal=7)
    ps = o.power_spectrum(da.chunk({"y": 16, "x": 16}), dim=["y", "x"], shift=False, density=False, chunks_to_segments=True)
    npt.assert_almost_equal(ps.values, (ft.values * np.conj(ft.values)).real)
    with pytest.raises(ValueError):
        o.fft(da.chunk({"time": 20}), dim=["time"], detrend="linear", chunks_to_segments=True)
