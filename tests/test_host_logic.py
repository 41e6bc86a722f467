"""Host-side behaviour that must match the reference before any device work happens: argument validation and the
exception types of xrft/xrft.py (SURVEY.md section 5 "config / flags"), the labelled-array container, frequency
coordinates.  No library is needed: every error below is raised by the host analysis."""
import numpy as np
import pytest

import xrft_amd as xa
from xrft_amd import api


def _da(shape=(2, 16, 16), dims=("time", "x", "y"), **coords):
    rng = np.random.default_rng(0)
    c = {d: np.arange(n) for d, n in zip(dims, shape)}
    c.update(coords)
    return xa.DataArray(rng.random(shape), dims, c)


def test_container_basics():
    da = _da()
    assert da.dims == ("time", "x", "y") and da.sizes == {"time": 2, "x": 16, "y": 16}
    assert da.get_axis_num("y") == 2 and da["x"].values.tolist() == list(range(16))
    nocoord = xa.DataArray(np.zeros((3, 4)), ("a", "b"))
    assert nocoord["b"].values.tolist() == [0, 1, 2, 3]  # xarray semantics: a dimension without coordinate is arange
    t = da.transpose("y", "time", "x")
    assert t.shape == (16, 2, 16) and t.dims == ("y", "time", "x")
    assert da.isel(time=0).dims == ("x", "y") and da.isel(time=slice(0, 1)).shape == (1, 16, 16)
    assert da.mean("time").shape == (16, 16)
    with pytest.raises(ValueError):
        xa.DataArray(np.zeros((3, 4)), ("a", "b"), {"a": np.arange(5)})
    positional = xa.DataArray(np.zeros(4), dims=["x"], coords=[np.arange(4) * 0.5])  # xarray's coords=[x] form
    assert positional["x"].values[1] == 0.5


def test_real_dim_must_exist():
    """test_xrft.py:243-244"""
    with pytest.raises(ValueError):
        xa.fft(_da((16,), ("x",)), real_dim="y", detrend="constant")


def test_uneven_and_constant_coordinates():
    """test_xrft.py:94-97, 1114-1135, 1315-1333"""
    x = np.linspace(0, 1.0, 16)
    x[-1] += 0.001
    da = xa.DataArray(np.zeros(16), ("x",), {"x": x})
    with pytest.raises(ValueError):
        xa.fft(da, spacing_tol=1e-4)
    with pytest.raises(TypeError):
        xa.fft(da, spacing_tol="string")
    with pytest.raises(ValueError):
        xa.dft(xa.DataArray(np.zeros(20) + 0j, ("freq_x",), {"freq_x": np.zeros(20)}))


def test_bad_coords_and_kwargs():
    """test_xrft.py:204-210, 1350-1379"""
    da = _da(x_nondim=("x", np.arange(16)))
    with pytest.raises(ValueError):
        xa.power_spectrum(da)
    da = xa.DataArray(np.zeros((2, 5, 3)), ("time", "x", "y"), {"time": np.arange(2), "x": np.arange(5), "y": np.array(["a", "b", "c"])})
    with pytest.raises(ValueError):
        xa.power_spectrum(da)
    with pytest.raises(TypeError):
        xa.fft(None, dims=1)
    with pytest.raises(TypeError):
        xa.power_spectrum(_da(), dims=1)


def test_window_and_detrend_names():
    """xrft.py:48-75, detrend.py:46-50"""
    with pytest.raises(NotImplementedError):
        xa.fft(_da(), dim=["x", "y"], window="not_a_window")
    with pytest.raises(NotImplementedError):
        xa.fft(_da(), dim=["x", "y"], detrend="quadratic")
    with pytest.raises(NotImplementedError):
        xa.detrend(_da(), ["x", "y"], "quadratic")
    with pytest.raises(ValueError):
        xa.power_spectrum(_da(), dim=["x", "y"], window=None, window_correction=True)
    with pytest.raises(ValueError):
        xa.cross_spectrum(_da(), _da(), dim=["x", "y"], window=None, window_correction=True)


def test_isotropic_needs_two_dims_and_matching_fields():
    """test_xrft.py:1045-1046, 1073-1079, 1098-1099"""
    with pytest.raises(ValueError):
        xa.isotropic_power_spectrum(_da((2, 5, 16, 32), ("time", "z", "y", "x")), dim=["z", "y", "x"])
    a = _da((16, 16), ("y", "x"))
    b = _da((16, 16), ("lat", "lon"))
    with pytest.raises(ValueError):
        xa.isotropic_cross_spectrum(a, b)


def test_frequency_coordinates_match_numpy():
    """the host analysis alone (xrft.py:139-155, 178-192): names, values and the spacing attribute"""
    da = _da((4, 9, 16), ("time", "y", "x"), y=np.arange(9) * 0.5, x=np.arange(16) * 2.0 - 3)
    c = api._analyze(da, 1e-3, ["y", "x"], None, True, None, None, True, False, "freq_", None)
    np.testing.assert_array_equal(c.new_coords["freq_y"].values, np.fft.fftshift(np.fft.fftfreq(9, 0.5)))
    np.testing.assert_array_equal(c.new_coords["freq_x"].values, np.fft.fftshift(np.fft.fftfreq(16, 2.0)))
    assert c.new_coords["freq_x"].attrs["spacing"] == np.fft.fftfreq(16, 2.0)[1]
    assert c.lag_x == [da["y"].values[4], da["x"].values[8]]
    c = api._analyze(da, 1e-3, ["y"], "x", True, None, None, False, False, "freq_", None)
    assert c.shift is False and c.dim == ["y", "x"]
    np.testing.assert_array_equal(c.new_coords["freq_x"].values, np.fft.rfftfreq(16, 2.0))
    t = np.array(["2019-04-18", "2019-04-19", "2019-04-20", "2019-04-21"], dtype="datetime64")
    c = api._analyze(xa.DataArray(np.zeros(4), ("time",), {"time": t}), 1e-3, None, None, False, None, None, False, False, "freq_", None)
    np.testing.assert_allclose(c.new_coords["freq_time"].values, np.fft.fftfreq(4, 86400.0))
    c = api._analyze(xa.DataArray(np.zeros(8), ("freq_x",), {"freq_x": np.arange(8.0)}), 1e-3, None, None, True, None, None, True, False, "freq_", None)
    assert list(c.swap.values()) == ["x"]  # the prefix is stripped when already present (xrft.py:186)


def test_radial_bins_match_pandas_cut():
    """xrft.py:975-981, 921-923: codes and per-bin mean radius"""
    import pandas as pd

    k = np.fft.fftshift(np.fft.fftfreq(32, 1.0))
    l = np.fft.fftshift(np.fft.fftfreq(16, 1.0))
    codes, kr, nb, _ = api._radial_bins(k, l, 4)
    r = np.sqrt(k[:, None] ** 2 + l[None, :] ** 2)
    ref = pd.cut(r.ravel(), 4)
    assert nb == 4 and np.array_equal(codes.ravel(), ref.codes)
    for b in range(4):
        np.testing.assert_allclose(kr[b], r.ravel()[ref.codes == b].mean(), rtol=1e-14)


def test_xarray_chunks_become_metadata():
    """A dask-chunked xarray input keeps its chunk layout (chunks_to_segments and the multi-chunk refusal depend on it)."""
    xr = pytest.importorskip("xarray")
    pytest.importorskip("dask")
    import xrft_amd as xa
    from xrft_amd.labeled import DataArray

    x = xr.DataArray(np.arange(24.0).reshape(2, 12), dims=("t", "x"), coords={"x": np.arange(12.0)}).chunk({"x": 4})
    d = DataArray.from_xarray(x)
    assert d._chunks == {"t": (2,), "x": (4, 4, 4)}
    with pytest.raises(ValueError):
        xa.fft(x, dim=["x"])


def test_coordinates_are_owned_and_read_only_and_the_analysis_is_remembered_per_label_set():
    """Round 4 (host overhead per call): coordinate vectors are owned read-only copies (xarray's dimension coordinates are immutable
    indexes, too), so api._analyze may remember what it derived from a labelled array by the identity of its labels; new labels -- a
    replaced coordinate, other arguments -- are analysed (and validated) afresh."""
    x = np.arange(16) * 0.5
    da = xa.DataArray(np.zeros((2, 16)), ("t", "x"), {"t": np.arange(2), "x": x})
    x[3] = 99.0                                   # the caller's array is not the coordinate
    assert da["x"].values[3] == 1.5 and not da["x"].values.flags.writeable
    with pytest.raises(ValueError):
        da["x"].values[3] = 7.0
    c1 = api._analyze(da, 1e-3, ["x"], None, True, None, None, True, False, "freq_", None)
    c2 = api._analyze(da, 1e-3, ["x"], None, True, None, None, True, False, "freq_", None)
    assert c1.k[0] is c2.k[0] and c2.da is da and c1._x is c2._x   # the stored context, re-bound to the array
    assert da._memo[1] and all(v.da is None for v in da._memo[1].values())  # (no reference cycle through the array)
    c3 = api._analyze(da, 1e-3, ["x"], None, False, None, None, True, False, "freq_", None)
    assert c3.shift is False and not np.array_equal(c3.k[0], c1.k[0])
    # the same array object with a coordinate replaced by an unevenly spaced one: validated again, and refused
    bad = np.arange(16) * 0.5
    bad[5] += 0.2
    da.coords["x"] = xa.Coordinate(("x",), bad, None, "x")
    with pytest.raises(ValueError, match="not evenly spaced"):
        api._analyze(da, 1e-3, ["x"], None, True, None, None, True, False, "freq_", None)
    # a list-valued window argument cannot key the memo: analysed every time, same answer
    da2 = xa.DataArray(np.zeros((2, 16)), ("t", "x"), {"t": np.arange(2), "x": np.arange(16) * 0.5})
    f1 = api._flags_tables(api._analyze(da2, 1e-3, ["x"], None, True, None, None, True, False, "freq_", None), da2)
    f2 = api._flags_tables(api._analyze(da2, 1e-3, ["x"], None, True, None, None, True, False, "freq_", None), da2)
    assert f1[0] == f2[0] and f1[2]["x"] is f2[2]["x"] and not f1[2]["x"].flags.writeable  # the phase table: built once


def test_label_memo_is_keyed_by_tokens_not_by_object_identity():
    """what the API remembers about a labelled array is keyed by a per-assignment token of each coordinate (never reused, unlike id()),
    nothing is remembered once a coordinate array is writeable again, and a read-only VIEW of a writeable base is digested by content."""
    import xrft_amd as xa
    from xrft_amd import api

    da = xa.DataArray(np.zeros((4, 8)), ("t", "x"), {"t": np.arange(4), "x": np.arange(8) * 0.5})
    g0 = api._label_guard(da)
    assert g0 is not None and api._label_guard(da) == g0
    tok = da.coords["x"]._token
    da.coords["x"].values = np.arange(8) * 0.25  # a new coordinate vector: a new token, read-only again
    assert da.coords["x"]._token != tok and not da.coords["x"].values.flags.writeable
    assert api._label_guard(da) != g0
    da.coords["x"].values.setflags(write=True)    # re-opened for in-place edits: nothing may be remembered
    assert api._label_guard(da) is None
    base = np.arange(8.0)
    view = base[:]
    view.setflags(write=False)
    k1 = api._akey(view)
    base[3] = 99.0
    assert api._akey(view) != k1
