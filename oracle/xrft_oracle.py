"""
CPU ORACLE for the xrft spectral hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module.  The product package (``xrft_amd``) never imports it and has no CPU fallback.

What it is
----------
A numpy/scipy/pandas restatement of the reference's algorithm (``/root/reference/xrft/xrft.py`` and
``/root/reference/xrft/detrend.py``) on plain ndarrays + explicit dimension names and coordinate vectors.
The reference is pure Python on top of xarray/dask; neither is installed in this image (nor on the GPU
box), so the reference cannot be imported end-to-end.  Its arithmetic is delegated to third-party code that
IS installed here and that this oracle calls in the same order with the same arguments:

  numpy.fft.{fftn,rfftn,fftshift,ifftshift,fftfreq,rfftfreq}  (numpy 2.2.6, pocketfft)   xrft.py:143-153,398-447
  scipy.signal.windows.<name>(n, sym=False)                     (scipy 1.15.3)              xrft.py:83-101
  scipy.signal.detrend                                          (scipy 1.15.3)              detrend.py:64-71
  scipy.linalg.inv                                              (scipy 1.15.3)              detrend.py:110
  pandas.cut                                                    (pandas 2.3.3)              xrft.py:921
  numpy_groupies.aggregate  -- ABSENT from the image; restated with numpy.bincount           xrft.py:898-906
      (published semantics: per-group sum / mean over the last axis, ``fill_value`` for empty groups).

Pinning (see tests/test_oracle_golden.py, tests/golden/make_golden.py)
-------
* ``_freq``, ``_detrend_2d_ufunc`` are checked against golden vectors produced by running the REFERENCE'S OWN
  source for those helpers in this container (they are pure numpy/scipy; imported from /root/reference with
  throw-away stub modules for xarray/dask, generator script committed as tests/golden/make_golden.py).
* every closed-form identity the reference's tests hold for this path is restated
  against this oracle in tests/test_oracle_identities.py.
* NOT pinned (cannot be observed without xarray): the dtype of the reference's outputs for float32 input
  (xarray.apply_ufunc casting), warning texts, and ``dropna`` on a NaN *coordinate* in ``isotropize``
  (restated from xarray's documented behaviour: ``dropna`` only inspects data values).

Each function cites the reference lines it follows.
"""
from __future__ import annotations

import operator
import warnings
from functools import reduce

import numpy as np
import pandas as pd
import scipy.linalg as spl
import scipy.signal as sps

__all__ = [
    "OArr",
    "fft",
    "ifft",
    "dft",
    "idft",
    "cross_phase",
    "detrend",
    "power_spectrum",
    "cross_spectrum",
    "isotropize",
    "isotropic_power_spectrum",
    "isotropic_cross_spectrum",
    "fit_loglog",
]


# --------------------------------------------------------------------------------------------------
# minimal labelled array (stands in for xarray.DataArray inside the oracle only)
# --------------------------------------------------------------------------------------------------
class OArr:
    """values + dims + 1-D dimension coordinates (+ optional extra coords) + per-coordinate attrs.

    ``coords``: {name: 1-D array} for dimension coordinates (name == dim) or {name: (dims_tuple, array)}
    for non-dimension coordinates.  A dimension without an entry behaves like xarray: its coordinate is
    ``arange(n)`` (cf. ``test_xrft.py:34-45`` "nocoords").
    """

    def __init__(self, values, dims, coords=None, attrs=None, coord_attrs=None, name=None, chunks=None):
        self.chunks = chunks  # {dim: tuple of chunk lengths} -- stands in for dask chunking (metadata only)
        self.values = np.asarray(values)
        self.dims = tuple(dims)
        assert self.values.ndim == len(self.dims), (self.values.shape, self.dims)
        self.coords = {}
        for k, v in (coords or {}).items():
            if isinstance(v, tuple) and len(v) == 2 and (
                    isinstance(v[0], str) or (isinstance(v[0], (tuple, list)) and all(isinstance(x, str) for x in v[0]))):
                d = (v[0],) if isinstance(v[0], str) else tuple(v[0])
                self.coords[k] = (d, np.asarray(v[1]))
            else:
                self.coords[k] = ((k,), np.asarray(v))
        self.attrs = dict(attrs or {})
        self.coord_attrs = {k: dict(v) for k, v in (coord_attrs or {}).items()}
        self.name = name

    # -- helpers mirroring the xarray calls the reference makes
    @property
    def shape(self):
        return self.values.shape

    def get_axis_num(self, d):
        return self.dims.index(d)

    def coord(self, d):
        """``da[d]`` for a dimension name: the coordinate vector (arange if absent)."""
        if d in self.coords:
            return self.coords[d][1]
        return np.arange(self.values.shape[self.get_axis_num(d)])

    def transpose(self, *dims):
        perm = [self.dims.index(d) for d in dims]
        return OArr(self.values.transpose(perm), dims, self._coords_raw(), self.attrs, self.coord_attrs, self.name)

    def _coords_raw(self):
        return {k: (v[0], v[1]) for k, v in self.coords.items()}

    def replace(self, values=None, dims=None):
        return OArr(self.values if values is None else values, self.dims if dims is None else dims,
                    self._coords_raw(), self.attrs, self.coord_attrs, self.name, self.chunks)

    def chunk(self, spec):
        """``da.chunk({dim: n})``: record equal-ish chunks of length n along dim (metadata only, like dask)."""
        ch = dict(self.chunks or {})
        for d, n in spec.items():
            N = self.shape[self.get_axis_num(d)]
            n = int(n)
            ch[d] = tuple([n] * (N // n) + ([N % n] if N % n else []))
        r = self.replace()
        r.chunks = ch
        return r

    def __repr__(self):
        return f"OArr(shape={self.shape}, dims={self.dims}, coords={list(self.coords)})"


# --------------------------------------------------------------------------------------------------
# helpers  (reference: xrft/xrft.py)
# --------------------------------------------------------------------------------------------------
_WINDOW_NAMES = [  # xrft.py:48-72
    "hann", "hamming", "kaiser", "tukey", "parzen", "taylor", "boxcar", "barthann", "bartlett",
    "blackman", "blackmanharris", "bohman", "chebwin", "cosine", "dpss", "exponential", "flattop",
    "gaussian", "general_cosine", "general_gaussian", "general_hamming", "triang", "nuttall",
]


def _window_vectors(da, dims, window_type):
    """xrft.py:39-101: one ``scipy.signal.windows.<name>(n, sym=False)`` vector per dim."""
    if window_type is True:  # xrft.py:42-47
        window_type = "hann"
        warnings.warn("Please provide the name of window adhering to scipy.signal.windows.", FutureWarning)
    elif window_type not in _WINDOW_NAMES:
        raise NotImplementedError(f"Window type {window_type} not supported.")
    if dims is None:
        dims = list(da.dims)
    elif isinstance(dims, str):
        dims = [dims]
    win_func = getattr(sps.windows, window_type)
    return dims, [win_func(len(da.coord(d)), sym=False) for d in dims]


def _broadcast_1d(vec, da, d):
    shape = [1] * len(da.dims)
    shape[da.get_axis_num(d)] = -1
    return np.reshape(vec, shape)


def _apply_window(da, dims, window_type="hann"):
    """xrft.py:39-103.  Returns (window as broadcastable ndarray over da.dims, da * window)."""
    dims, vecs = _window_vectors(da, dims, window_type)
    # reduce(operator.mul, windows[::-1]) -- xrft.py:103 (float64 windows: float32 data is promoted)
    win = reduce(operator.mul, [_broadcast_1d(v, da, d) for d, v in zip(dims, vecs)][::-1])
    return win, da.replace(values=da.values * win)


def _freq(N, delta_x, real, shift):
    """xrft.py:139-155 (verbatim semantics)."""
    if real is None:
        fftfreq = [np.fft.fftfreq] * len(N)
    else:
        fftfreq = [np.fft.fftfreq] * (len(N) - 1)
        fftfreq.append(np.fft.rfftfreq)
    k = [f(Nx, dx) for (f, Nx, dx) in zip(fftfreq, N, delta_x)]
    if shift:
        k = [np.fft.fftshift(l) for l in k]
    return k


def _ifreq(N, delta_x, real, shift):
    """xrft.py:158-175."""
    if real is None:
        fftfreq = [np.fft.fftfreq] * len(N)
    else:
        irfftfreq = lambda Nx, dx: np.fft.fftfreq(2 * (Nx - 1), dx)
        fftfreq = [np.fft.fftfreq] * (len(N) - 1)
        fftfreq.append(irfftfreq)
    k = [f(Nx, dx) for (f, Nx, dx) in zip(fftfreq, N, delta_x)]
    if shift:
        k = [np.fft.fftshift(l) for l in k]
    return k


def _stack_chunks(da, dim, suffix="_segment"):
    """xrft.py:106-136: reshape every chunked transform dimension d into (d_segment, d)."""
    newdims, newshape, newcoords = [], [], {}
    for d in da.dims:
        if d in dim:
            ch = (da.chunks or {}).get(d) or (da.shape[da.get_axis_num(d)],)
            if np.diff(ch).sum() != 0:
                raise ValueError("Chunk lengths need to be the same.")
            n = len(da.coord(d))
            chunklen = ch[0]
            coord_rs = da.coord(d).reshape((int(n / chunklen), int(chunklen)))
            newdims += [d + suffix, d]
            newshape += [int(n / chunklen), int(chunklen)]
            newcoords[d + suffix] = np.arange(int(n / chunklen))
            newcoords[d] = coord_rs[0]
        else:
            newdims.append(d)
            newshape.append(da.shape[da.get_axis_num(d)])
            if d in da.coords:
                newcoords[d] = da.coords[d][1]
    return OArr(da.values.reshape(newshape), newdims, newcoords, da.attrs)


def _diff_coord(coord):
    """xrft.py:195-212 (cftime branch omitted: cftime is not installed in this image)."""
    v0 = coord[0]
    if getattr(v0, "calendar", None):
        raise NotImplementedError("cftime coordinates: cftime is not available in this image")
    if pd.api.types.is_datetime64_dtype(v0):
        diff = np.diff(coord).astype("timedelta64[ns]").astype("f8")
        return diff / 1e9
    return np.diff(coord)


def _lag_coord(coord):
    """xrft.py:215-234."""
    v0 = coord[0]
    if coord[-1] > coord[0]:
        coord_data = coord
    else:
        coord_data = np.flip(coord, axis=-1)
    lag = coord_data[len(coord) // 2]
    if pd.api.types.is_datetime64_dtype(v0):
        return lag.astype("timedelta64[s]").astype("f8")
    return lag


def _is_valid_fft_coord(coord):
    """xrft.py:269-274."""
    from pandas.api.types import is_datetime64_any_dtype, is_numeric_dtype

    return bool(is_numeric_dtype(coord) or is_datetime64_any_dtype(coord)
                or bool(getattr(coord[0].item() if hasattr(coord[0], "item") else coord[0], "calendar", False)))


def _get_coordinate_spacing(coord, spacing_tol, name="?"):
    """xrft.py:291-304."""
    diff = _diff_coord(coord)
    delta = np.abs(diff[0])
    if not np.allclose(diff, diff[0], rtol=spacing_tol):
        raise ValueError("Can't take Fourier transform because coodinate %s is not evenly spaced" % name)
    if delta == 0.0:
        raise ValueError("Can't take Fourier transform because spacing in coordinate %s is zero" % name)
    return delta


def move_to_end(lst, el):
    """xrft.py:287-288."""
    return [i for i in lst if i != el] + [el]


# --------------------------------------------------------------------------------------------------
# detrend  (reference: xrft/detrend.py)
# --------------------------------------------------------------------------------------------------
def _detrend_2d_ufunc(arr):
    """detrend.py:100-113 restated: plane fit a + b(i+1) + c(j+1) through the normal equations."""
    assert arr.ndim == 2
    N = arr.shape
    col0 = np.ones(N[0] * N[1])
    col1 = np.repeat(np.arange(N[0]), N[1]) + 1
    col2 = np.tile(np.arange(N[1]), N[0]) + 1
    G = np.stack([col0, col1, col2]).transpose()
    d_obs = np.reshape(arr, (N[0] * N[1], 1))
    m_est = np.dot(np.dot(spl.inv(np.dot(G.T, G)), G.T), d_obs)
    d_est = np.dot(G, m_est)
    return arr - np.reshape(d_est, N)


def _detrend_3d_ufunc(arr):
    """detrend.py:116-138 restated: hyperplane a0 + a1(i+1) + a2(j+1) + a3(k+1) by numpy.linalg.lstsq."""
    assert arr.ndim == 3
    N0, N1, N2 = arr.shape
    i = np.repeat(np.arange(N0), N1 * N2) + 1
    j = np.tile(np.repeat(np.arange(N1), N2), N0) + 1
    k = np.tile(np.arange(N2), N0 * N1) + 1
    G = np.stack([np.ones(N0 * N1 * N2), i, j, k], axis=1)
    m_est, _, _, _ = np.linalg.lstsq(G, arr.reshape(-1, 1), rcond=None)
    return arr - (G @ m_est).reshape(N0, N1, N2)


def detrend(da, dim, detrend_type="constant"):
    """detrend.py:11-97 (constant over any dims; linear 1-D, 2-D, 3-D)."""
    if dim is None:
        dim = list(da.dims)
    elif isinstance(dim, str):
        dim = [dim]
    if detrend_type not in ["constant", "linear", None]:
        raise NotImplementedError("%s is not a valid detrending option." % detrend_type)
    if detrend_type is None:
        return da
    axis_num = [da.get_axis_num(d) for d in dim]
    if detrend_type == "constant":  # detrend.py:54-55
        return da.replace(values=da.values - da.values.mean(axis=tuple(axis_num), keepdims=True))
    if len(dim) == 1:  # detrend.py:64-71
        out = sps.detrend(da.values, axis_num[0])
        return da.replace(values=out.astype(da.values.dtype, copy=False))
    if len(dim) == 2:  # detrend.py:72-81: apply_ufunc(vectorize=True) moves the core dims last, in `dim` order
        v = np.moveaxis(da.values, axis_num, [-2, -1])
        out = np.empty(v.shape, dtype=da.values.dtype)  # output_dtypes=[da.dtype]
        for idx in np.ndindex(*v.shape[:-2]):
            out[idx] = _detrend_2d_ufunc(v[idx])
        other = [d for d in da.dims if d not in dim]
        res = OArr(out, tuple(other) + tuple(dim), da._coords_raw(), da.attrs, da.coord_attrs, da.name)
        return res  # NB: core dims are last, as with apply_ufunc; fft() transposes back (xrft.py:427-428)
    if len(dim) == 3:  # detrend.py:82-91
        v = np.moveaxis(da.values, axis_num, [-3, -2, -1])
        out = np.empty(v.shape, dtype=da.values.dtype)
        for idx in np.ndindex(*v.shape[:-3]):
            out[idx] = _detrend_3d_ufunc(v[idx])
        other = [d for d in da.dims if d not in dim]
        return OArr(out, tuple(other) + tuple(dim), da._coords_raw(), da.attrs, da.coord_attrs, da.name)
    raise NotImplementedError("Only 1D, 2D, and 3D detrending are implemented so far.")


# --------------------------------------------------------------------------------------------------
# pad / unpad  (reference: xrft/padding.py)
# --------------------------------------------------------------------------------------------------
def _pad_coordinates_callback(vector, iaxis_pad_width, iaxis, kwargs):
    """padding.py:277-323."""
    spacing = kwargs["spacing"]
    n_start, n_end = iaxis_pad_width[:]
    vmin, vmax = vector[n_start], vector[-(n_end + 1)]
    vector[:n_start] = vmin - n_start * spacing + np.linspace(0, spacing * (n_start - 1), n_start)
    vector[len(vector) - n_end:] = vmax + spacing + np.linspace(0, spacing * (n_end - 1), n_end)
    return vector


def pad(da, pad_width=None, mode="constant", constant_values=0, **pad_width_kwargs):
    """padding.py:11-193 for the gather / constant modes: numpy.pad on the values (what xarray's .pad does), linear
    extrapolation of the coordinates, ``pad_width`` attribute on every padded coordinate."""
    pad_width = dict(pad_width or {}, **pad_width_kwargs)
    full = [(0, 0)] * da.values.ndim
    for d, w in pad_width.items():
        full[da.get_axis_num(d)] = (w, w) if isinstance(w, int) else tuple(w)
    kw = {"constant_values": constant_values} if mode == "constant" else {}
    values = np.pad(da.values, full, mode=mode, **kw)
    coords = dict(da._coords_raw())
    cattrs = {k: dict(v) for k, v in da.coord_attrs.items()}
    for d, w in pad_width.items():
        c = np.asarray(da.coord(d))
        spacing = np.diff(c)[0]
        coords[d] = np.pad(c, full[da.get_axis_num(d)], mode=_pad_coordinates_callback, spacing=spacing)
        cattrs.setdefault(d, {})["pad_width"] = w
    return OArr(values, da.dims, coords, da.attrs, cattrs, da.name)


def unpad(da, pad_width=None, **pad_width_kwargs):
    """padding.py:326-446."""
    if pad_width is None and not pad_width_kwargs:
        pad_width = {d: a["pad_width"] for d, a in da.coord_attrs.items() if "pad_width" in a}
        if not pad_width:
            raise ValueError("The passed array doesn't seem to be a padded one")
    else:
        pad_width = dict(pad_width or {}, **pad_width_kwargs)
    values = da.values
    coords = dict(da._coords_raw())
    cattrs = {k: dict(v) for k, v in da.coord_attrs.items()}
    for d, w in pad_width.items():
        w = (w, w) if isinstance(w, int) else tuple(w)
        ax = da.get_axis_num(d)
        sl = slice(w[0], values.shape[ax] - w[1])
        values = values[(slice(None),) * ax + (sl,)]
        coords[d] = np.asarray(da.coord(d))[sl]
        cattrs.get(d, {}).pop("pad_width", None)
    return OArr(values, da.dims, coords, da.attrs, cattrs, da.name)


# --------------------------------------------------------------------------------------------------
# fft / dft  (reference: xrft/xrft.py:237-250, 307-476)
# --------------------------------------------------------------------------------------------------
def fft(da, spacing_tol=1e-3, dim=None, real_dim=None, shift=True, detrend=None, window=None,
        true_phase=True, true_amplitude=True, chunks_to_segments=False, prefix="freq_", real=None):
    """xrft.py:307-476 (``chunks_to_segments`` uses the chunk metadata of OArr.chunk())."""
    _detrend_kind = detrend
    if dim is None:
        dim = list(da.dims)
    elif isinstance(dim, str):
        dim = [dim]
    else:
        dim = list(dim)

    if real is not None:  # xrft.py:376-378
        real_dim = real
        warnings.warn("`real` flag will be deprecated", FutureWarning)

    if real_dim is not None:  # xrft.py:380-386
        if real_dim not in da.dims:
            raise ValueError("The dimension along which real FT is taken must be one of the existing dimensions.")
        dim = move_to_end(dim, real_dim)

    if not np.all([_is_valid_fft_coord(da.coord(d)) for d in dim]):  # xrft.py:277-281
        raise ValueError("All transformed dimensions coordinates must be numerical or datetime.")

    if chunks_to_segments:  # xrft.py:390-391
        da = _stack_chunks(da, dim)
    elif da.chunks and any(len(da.chunks.get(d, (0,))) > 1 for d in dim):
        raise ValueError("dask.array.fft refuses to transform along an axis that has more than one chunk")

    rawdims = da.dims
    if real_dim is not None:  # xrft.py:395-396
        da = da.transpose(*move_to_end(list(da.dims), real_dim))

    if real_dim is None:  # xrft.py:400-404
        fft_fn = np.fft.fftn
    else:
        shift = False
        fft_fn = np.fft.rfftn

    axis_num = [da.get_axis_num(d) for d in dim]
    N = [da.shape[n] for n in axis_num]

    for d in dim:  # xrft.py:412-420
        bad = [c for c, (cd, _) in da.coords.items() if c != d and d in cd]
        if bad:
            raise ValueError(f"The input array contains coordinate variable(s) ({bad}) whose dims include "
                             f"the transform dimension(s) `{d}`.")

    delta_x = [_get_coordinate_spacing(da.coord(d), spacing_tol, d) for d in dim]  # xrft.py:422
    lag_x = [_lag_coord(da.coord(d)) for d in dim]  # xrft.py:423

    if _detrend_kind is not None:  # xrft.py:425-430
        orig = da.dims
        da = globals()["detrend"](da, dim, detrend_type=_detrend_kind)
        if _detrend_kind == "linear":
            da = da.transpose(*orig)

    if window is not None:  # xrft.py:432-433
        _, da = _apply_window(da, dim, window_type=window)

    if true_phase:  # xrft.py:435-442
        reversed_axis = [da.get_axis_num(d) for d in dim if da.coord(d)[-1] < da.coord(d)[0]]
        f = fft_fn(np.fft.ifftshift(np.flip(da.values, axis=reversed_axis), axes=axis_num), axes=axis_num)
    else:
        f = fft_fn(da.values, axes=axis_num)

    if shift:  # xrft.py:446-447
        f = np.fft.fftshift(f, axes=axis_num)

    k = _freq(N, delta_x, real_dim, shift)  # xrft.py:449

    # xrft.py:178-192 and 451-456: rename dims, attach freq coords (+spacing attr), drop transform-dim coords
    swap = {}
    new_coords = {}
    new_cattrs = {}
    for d, kk in zip(dim, k):
        new_name = prefix + d if d[: len(prefix)] != prefix else d[len(prefix):]
        swap[d] = new_name
        new_coords[new_name] = ((new_name,), kk)
        new_cattrs[new_name] = {"spacing": kk[1] - kk[0]}
    out_dims = tuple(swap.get(d, d) for d in da.dims)
    kept = {c: (tuple(swap.get(x, x) for x in cd), cv) for c, (cd, cv) in da.coords.items() if c not in dim}
    kept.update(new_coords)
    cattrs = {c: a for c, a in da.coord_attrs.items() if c not in dim}
    cattrs.update(new_cattrs)

    updated_dims = [out_dims[i] for i in axis_num]

    if true_phase:  # xrft.py:462-469
        for up_dim, lag, ax in zip(updated_dims, lag_x, axis_num):
            ph = np.exp(-1j * 2.0 * np.pi * kept[up_dim][1] * lag)
            shape = [1] * f.ndim
            shape[ax] = -1
            f = f * ph.reshape(shape)
            cattrs[up_dim]["direct_lag"] = lag

    if true_amplitude:  # xrft.py:471-472
        f = f * np.prod(delta_x)

    daft = OArr(f, out_dims, kept, None, cattrs, None)
    return daft.transpose(*[swap.get(d, d) for d in rawdims])  # xrft.py:474-476


def dft(da, dim=None, true_phase=False, true_amplitude=False, **kwargs):
    """xrft.py:237-250 (deprecated alias; note the different defaults)."""
    warnings.warn("This function has been renamed and will disappear in the future. Please use `fft` instead",
                  FutureWarning)
    return fft(da, dim=dim, true_phase=true_phase, true_amplitude=true_amplitude, **kwargs)


def ifft(daft, spacing_tol=1e-3, dim=None, real_dim=None, shift=True, true_phase=True, true_amplitude=True,
         chunks_to_segments=False, prefix="freq_", lag=None, real=None):
    """xrft.py:479-646."""
    if dim is None:
        dim = list(daft.dims)
    elif isinstance(dim, str):
        dim = [dim]
    else:
        dim = list(dim)
    if real is not None:
        real_dim = real
        warnings.warn("`real` flag will be deprecated", FutureWarning)
    if real_dim is not None:
        if real_dim not in daft.dims:
            raise ValueError("The dimension along which real IFT is taken must be one of the existing dimensions.")
        dim = move_to_end(dim, real_dim)
    if not np.all([_is_valid_fft_coord(daft.coord(d)) for d in dim]):
        raise ValueError("All transformed dimensions coordinates must be numerical or datetime.")
    if lag is None:  # xrft.py:557-560
        lag = [daft.coord_attrs.get(d, {}).get("direct_lag", 0.0) for d in dim]
        warnings.warn("Default ifft's behaviour (lag=None) changed!", FutureWarning)
    else:
        if isinstance(lag, float) or isinstance(lag, int):
            lag = [lag]
        if len(dim) != len(lag):
            raise ValueError("dim and lag must have the same length.")
        if not true_phase:
            warnings.warn("Setting lag with true_phase=False does not guarantee accurate ifft.", Warning)
        lag = [daft.coord_attrs.get(d, {}).get("direct_lag") if l is None else l for d, l in zip(dim, lag)]
    v = daft.values
    if true_phase:  # xrft.py:574-576
        for d, l in zip(dim, lag):
            v = v * _broadcast_1d(np.exp(1j * 2.0 * np.pi * daft.coord(d) * l), daft, d)
        daft = daft.replace(values=v)
    if chunks_to_segments:
        daft = _stack_chunks(daft, dim)
    rawdims = daft.dims
    if real_dim is not None:
        daft = daft.transpose(*move_to_end(list(daft.dims), real_dim))
    fft_fn = np.fft.ifftn if real_dim is None else np.fft.irfftn
    axis_num = [daft.get_axis_num(d) for d in dim]
    N = [daft.shape[n] for n in axis_num]
    # daft.sortby(dim): sort by the coordinates (handles fftshifted grids), xrft.py:598
    vals = daft.values
    coords = daft._coords_raw()
    for d in dim:
        order = np.argsort(daft.coord(d), kind="stable")
        vals = np.take(vals, order, axis=daft.get_axis_num(d))
        coords[d] = ((d,), daft.coord(d)[order])
    daft = OArr(vals, daft.dims, coords, daft.attrs, daft.coord_attrs, daft.name)
    delta_x = [_get_coordinate_spacing(daft.coord(d), spacing_tol, d) for d in dim]
    for d in dim:  # xrft.py:600-606
        l = _lag_coord(daft.coord(d)) if d is not real_dim else daft.coord(d)[0]
        if np.abs(l) > spacing_tol:
            raise ValueError("Inverse Fourier Transform can not be computed because coordinate %s is not centered "
                             "on zero frequency" % d)
    axis_shift = [daft.get_axis_num(d) for d in dim if d is not real_dim]
    f = np.fft.ifftshift(daft.values, axes=axis_shift)
    f = fft_fn(f, axes=axis_num)
    if not true_phase:
        f = np.fft.ifftshift(f, axes=axis_num)
    if shift:
        f = np.fft.fftshift(f, axes=axis_num)
    k = _ifreq(N, delta_x, real_dim, shift)
    swap, new_coords, cattrs = {}, {}, {}
    for d, kk in zip(dim, k):
        new_name = prefix + d if d[: len(prefix)] != prefix else d[len(prefix):]
        swap[d] = new_name
        new_coords[new_name] = kk
        cattrs[new_name] = {"spacing": kk[1] - kk[0]}
    out_dims = tuple(swap.get(d, d) for d in daft.dims)
    kept = {c: (tuple(swap.get(x, x) for x in cd), cv) for c, (cd, cv) in daft.coords.items() if c not in dim}
    for d, l in zip(dim, lag):  # xrft.py:634-639
        tfd = swap[d]
        kept[tfd] = ((tfd,), new_coords[tfd] + l)
    if true_amplitude:  # xrft.py:641-642
        f = f / np.prod([float(cattrs[up]["spacing"]) for up in swap.values()])
    other_attrs = {c: a for c, a in daft.coord_attrs.items() if c not in dim}
    other_attrs.update(cattrs)
    da = OArr(f, out_dims, kept, None, other_attrs, None)
    return da.transpose(*[swap.get(d, d) for d in rawdims])


def idft(daft, dim=None, true_phase=False, true_amplitude=False, **kwargs):
    """xrft.py:253-266."""
    warnings.warn("This function has been renamed and will disappear in the future. Please use `ifft` instead",
                  FutureWarning)
    return ifft(daft, dim=dim, true_phase=true_phase, true_amplitude=true_amplitude, **kwargs)


# --------------------------------------------------------------------------------------------------
# spectra  (reference: xrft/xrft.py:649-835)
# --------------------------------------------------------------------------------------------------
def _window_correction_factor(da, dim, scaling, window):
    """xrft.py:649-660."""
    if window is None:
        raise ValueError("window_correction can only be applied when windowing is turned on.")
    windows, _ = _apply_window(da, dim, window_type=window)
    # ``windows`` is the outer-product window over the transform dims only; mean over those dims
    if scaling == "density":
        return (windows ** 2).mean()
    elif scaling == "spectrum":
        return windows.mean() ** 2
    raise ValueError("Unknown {} scaling flag".format(scaling))


def _psd_scaling_factor(ps, dims, scaling):
    """xrft.py:663-670."""
    fs = np.prod([float(ps.coord_attrs[d]["spacing"]) for d in dims])
    if scaling == "density":
        return fs
    elif scaling == "spectrum":
        return fs ** 2
    raise ValueError("Unknown {} scaling flag".format(scaling))


def _psd_real_dim_scaling(da, ps, real_dim, updated_dims):
    """xrft.py:673-682; returns (axis, factor vector)."""
    real = next(d for d in updated_dims if d.endswith(real_dim))
    n = ps.shape[ps.get_axis_num(real)]
    f = np.full(n, 2.0)
    if len(da.coord(real_dim)) % 2 == 0:
        f[0], f[-1] = 1.0, 1.0
    else:
        f[0] = 1.0
    return real, f


def _spectrum_tail(da, sp, dim, real_dim, scaling, window_correction, window, updated_dims):
    """Common tail of power_spectrum / cross_spectrum: xrft.py:742-748 and 827-833."""
    v = sp.values
    if real_dim is not None:
        real, f = _psd_real_dim_scaling(da, sp, real_dim, updated_dims)
        v = v * _broadcast_1d(f, sp, real)
    if scaling != "false_density":
        if window_correction:
            v = v / _window_correction_factor(da, dim, scaling, window)
        v = v * _psd_scaling_factor(sp, updated_dims, scaling)
    return sp.replace(values=v)


def power_spectrum(da, dim=None, real_dim=None, scaling="density", window_correction=False, **kwargs):
    """xrft.py:685-750."""
    if "density" in kwargs:  # xrft.py:718-726
        density = kwargs.pop("density")
        warnings.warn("density flag will be deprecated", FutureWarning)
        scaling = "density" if density else "false_density"
    if "real" in kwargs:  # xrft.py:728-730 (.get, not .pop: `real` is forwarded to fft as well)
        real_dim = kwargs.get("real")
        warnings.warn("`real` flag will be deprecated", FutureWarning)
    kwargs.update({"true_amplitude": True, "true_phase": False})  # xrft.py:732-734
    daft = fft(da, dim=dim, real_dim=real_dim, **kwargs)
    updated_dims = [d for d in daft.dims if (d not in da.dims and "segment" not in d)]
    ps = daft.replace(values=np.abs(daft.values) ** 2)  # xrft.py:740
    return _spectrum_tail(da, ps, dim, real_dim, scaling, window_correction, kwargs.get("window"), updated_dims)


def cross_spectrum(da1, da2, dim=None, real_dim=None, scaling="density", window_correction=False,
                   true_phase=True, **kwargs):
    """xrft.py:753-835."""
    if "real" in kwargs:
        real_dim = kwargs.get("real")
        warnings.warn("`real` flag will be deprecated", FutureWarning)
    if "density" in kwargs:
        density = kwargs.pop("density")
        warnings.warn("density flag will be deprecated", FutureWarning)
        scaling = "density" if density else "false_density"
    kwargs.update({"true_amplitude": True})  # xrft.py:814
    daft1 = fft(da1, dim=dim, real_dim=real_dim, true_phase=true_phase, **kwargs)
    daft2 = fft(da2, dim=dim, real_dim=real_dim, true_phase=true_phase, **kwargs)
    if daft1.dims != daft2.dims:  # xrft.py:819-820
        raise ValueError("The two datasets have different dimensions")
    updated_dims = [d for d in daft1.dims if (d not in da1.dims and "segment" not in d)]
    cs = daft1.replace(values=daft1.values * np.conj(daft2.values))  # xrft.py:825
    return _spectrum_tail(da1, cs, dim, real_dim, scaling, window_correction, kwargs.get("window"), updated_dims)


def cross_phase(da1, da2, dim=None, true_phase=True, **kwargs):
    """xrft.py:838-874."""
    cs = cross_spectrum(da1, da2, dim=dim, true_phase=true_phase, **kwargs)
    cp = cs.replace(values=np.angle(cs.values))
    if da1.name and da2.name:
        cp.name = "{}_{}_phase".format(da1.name, da2.name)
    return cp


# --------------------------------------------------------------------------------------------------
# isotropic spectra  (reference: xrft/xrft.py:877-1187)
# --------------------------------------------------------------------------------------------------
def _aggregate(int_indices, array, func, size, fill_value, dtype):
    """numpy_groupies.aggregate(idx, a, func, size, fill_value, dtype, axis=-1) restated (package absent).

    ``sum``: per-group sum over the last axis.  ``mean``: per-group mean, ``fill_value`` for empty groups.
    Called from xrft.py:898-906.
    """
    array = np.asarray(array)
    lead = array.shape[:-1]
    flat = array.reshape(-1, array.shape[-1])
    counts = np.bincount(int_indices, minlength=size)
    out_dtype = dtype if dtype is not None else (array.dtype if func == "sum" else np.result_type(array.dtype, np.float64))
    out = np.empty((flat.shape[0], size), dtype=out_dtype)
    for r in range(flat.shape[0]):
        row = flat[r]
        if np.iscomplexobj(row):
            s = np.bincount(int_indices, weights=row.real, minlength=size) + 1j * np.bincount(
                int_indices, weights=row.imag, minlength=size)
        else:
            s = np.bincount(int_indices, weights=row, minlength=size)
        if func == "sum":
            res = s
        elif func == "mean":
            with np.errstate(invalid="ignore", divide="ignore"):
                res = np.where(counts > 0, s / np.maximum(counts, 1), fill_value)
        else:
            raise ValueError(func)
        out[r] = res
    return out.reshape(lead + (size,))


def _groupby_bins_agg(array_vals, group_vals, bins, func="sum", fill_value=0, dtype=None):
    """xrft.py:910-945.  ``array_vals[..., *group.shape]``, ``group_vals`` 2-D; returns (result, categories)."""
    binned = pd.cut(np.ravel(group_vals), bins)  # xrft.py:921
    indices = binned.codes.reshape(group_vals.shape)  # xrft.py:923
    num_bins = binned.categories.size
    # _binned_agg, xrft.py:895-897: mask = ~isnan(indices) (all True for integer codes)
    mask = np.logical_not(np.isnan(indices))
    int_indices = indices[mask].astype(int)
    res = _aggregate(int_indices, array_vals[..., mask], func, num_bins, fill_value, dtype)
    return res, binned.categories


def isotropize(ps, fftdim, nfactor=4, truncate=True, complx=False):
    """xrft.py:948-1010."""
    k = ps.coord(fftdim[1])
    l = ps.coord(fftdim[0])
    N = [k.size, l.size]
    nbins = int(min(N) / nfactor)
    freq_r = np.sqrt(k[:, None] ** 2 + l[None, :] ** 2)  # dims (fftdim[1], fftdim[0]); xrft.py:980
    kr, _ = _groupby_bins_agg(freq_r, freq_r, bins=nbins, func="mean")  # xrft.py:981

    if truncate:  # xrft.py:983-988
        kmax = l.max() if k.max() > l.max() else k.max()
        kr = np.where(kr <= kmax, kr, np.nan)
    else:
        warnings.warn("Isotropic wavenumber larger than the Nyquist wavenumber may result.", FutureWarning)

    # apply_ufunc moves the core dims (fftdim[1], fftdim[0]) last, in that order (xrft.py:925-930)
    ax = [ps.get_axis_num(fftdim[1]), ps.get_axis_num(fftdim[0])]
    v = np.moveaxis(ps.values, ax, [-2, -1])
    iso, _ = _groupby_bins_agg(v, freq_r, bins=nbins, func="sum", dtype=np.complex128 if complx else None)
    other = [d for d in ps.dims if d not in fftdim]
    coords = {c: cv for c, cv in ps._coords_raw().items() if not (set(cv[0]) & set(fftdim))}
    coords["freq_r"] = (("freq_r",), kr)
    out = OArr(iso, tuple(other) + ("freq_r",), coords, ps.attrs,
               {c: a for c, a in ps.coord_attrs.items() if c not in fftdim}, ps.name)
    if truncate:  # xrft.py:1007-1008: dropna inspects DATA values only (coordinate NaNs are kept)
        keep = ~np.isnan(out.values).reshape(-1, out.values.shape[-1]).any(axis=0)
        if not keep.all():
            coords["freq_r"] = (("freq_r",), kr[keep])
            out = OArr(out.values[..., keep], out.dims, coords, out.attrs, out.coord_attrs, out.name)
    return out


def isotropic_power_spectrum(da, spacing_tol=1e-3, dim=None, shift=True, detrend=None, scaling="density",
                             window=None, window_correction=False, nfactor=4, truncate=False, **kwargs):
    """xrft.py:1013-1095."""
    if "density" in kwargs:
        density = kwargs.pop("density")
        scaling = "density" if density else "false_density"
    if dim is None:
        dim = da.dims
    if len(dim) != 2:
        raise ValueError("The Fourier transform should be two dimensional")
    ps = power_spectrum(da, spacing_tol=spacing_tol, dim=dim, shift=shift, detrend=detrend, scaling=scaling,
                        window_correction=window_correction, window=window, **kwargs)
    fftdim = ["freq_" + d for d in dim]
    return isotropize(ps, fftdim, nfactor=nfactor, truncate=truncate)


def isotropic_cross_spectrum(da1, da2, spacing_tol=1e-3, dim=None, shift=True, detrend=None,
                             scaling="density", window=None, window_correction=False, nfactor=4,
                             truncate=False, **kwargs):
    """xrft.py:1098-1187."""
    if "density" in kwargs:
        density = kwargs.pop("density")
        scaling = "density" if density else "false_density"
    if dim is None:
        dim = da1.dims
        if dim != da2.dims:
            raise ValueError("The two datasets have different dimensions")
    if len(dim) != 2:
        raise ValueError("The Fourier transform should be two dimensional")
    cs = cross_spectrum(da1, da2, spacing_tol=spacing_tol, dim=dim, shift=shift, detrend=detrend,
                        scaling=scaling, window_correction=window_correction, window=window, **kwargs)
    fftdim = ["freq_" + d for d in dim]
    return isotropize(cs, fftdim, nfactor=nfactor, truncate=truncate, complx=True)


def fit_loglog(x, y):
    """xrft.py:1190-1214."""
    p = np.polyfit(np.log2(x), np.log2(y), 1)
    y_fit = 2 ** (np.log2(x) * p[0] + p[1])
    return y_fit, p[0], p[1]


# --------------------------------------------------------------------------------------------------
# synthetic red-noise field used by the reference's isotropic tests (test_xrft.py:845-914), seeded
# --------------------------------------------------------------------------------------------------
def synthetic_field(N, dL, amp, s, rng):
    """test_xrft.py:845-914 with an explicit ``numpy.random.Generator`` instead of the global RNG."""
    k = np.fft.fftshift(np.fft.fftfreq(N, dL))
    kk, ll = np.meshgrid(k, k)
    K = np.sqrt(kk ** 2 + ll ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        r_kl = np.ma.masked_invalid(np.sqrt(amp * 0.5 * (np.pi) ** (-1) * K ** (s - 1.0))).filled(0.0)
    phi = np.zeros((N, N))
    N_2 = int(N / 2)
    u = lambda *shape: 2.0 * np.pi * rng.random(shape if shape else None) - np.pi
    a = u(N_2 - 1, N_2 - 1)
    phi[N_2 + 1:, N_2 + 1:] = a
    phi[1:N_2, 1:N_2] = -a[::-1, ::-1]
    b = u(N_2 - 1, N_2 - 1)
    phi[N_2 + 1:, 1:N_2] = b
    phi[1:N_2, N_2 + 1:] = -b[::-1, ::-1]
    c = u(N_2)
    phi[N_2:, N_2] = c
    phi[1:N_2, N_2] = -c[1:][::-1]
    d = u(N_2 - 1)
    phi[N_2, N_2 + 1:] = d
    phi[N_2, 1:N_2] = -d[::-1]
    e = u(N_2)
    phi[N_2:, 0] = e
    phi[1:N_2, 0] = -e[1:][::-1]
    g = u(N_2)
    phi[0, N_2:] = g
    phi[0, 1:N_2] = -g[1:][::-1]
    F_theta = r_kl * np.exp(1j * phi)
    theta = np.fft.ifft2(np.fft.ifftshift(F_theta))
    return np.real(theta)
