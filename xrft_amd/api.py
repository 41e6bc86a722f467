"""The xrft call surface (same names, argument meaning and error behaviour as the reference's
``xrft/xrft.py`` and ``xrft/detrend.py``), executed by the MI355X engine in libxrft_hip.so.

Host side (this file): argument normalisation, coordinate validation, spacing / lag / frequency vectors,
window vectors, the radial bin map -- all tiny float64 numpy work that must match the reference bit for bit.
Device side (one fused plan per call): detrend, window, flip/ifftshift, FFT, fftshift, phase, scaling,
|F|^2 / F conj(G), Hermitian mirror, radial bin-sum.

Deviations from the reference, all documented in DESIGN.md:
  * float32 input is computed and returned in float32/complex64 (the reference promotes to float64 as soon as a
    float64 window or ``prod(dx)`` touches the data); isotropic results are returned in float64/complex128;
  * at most two transform dimensions (the reference also offers 3-D linear detrend / N-D fftn);
  * ``chunks_to_segments`` takes its segment length from ``DataArray.chunk({dim: n})`` metadata (no dask here).
"""
from __future__ import annotations

import collections
import math
import threading
import warnings
from collections import OrderedDict

import numpy as np
import pandas as pd
import scipy.signal as sps
import torch
from pandas.api.types import is_datetime64_any_dtype, is_numeric_dtype

from . import _lib, engine
from .labeled import Coordinate, DataArray, from_any, to_like

__all__ = ["clear_plan_cache", "fft", "ifft", "dft", "idft", "detrend", "power_spectrum", "cross_spectrum", "cross_phase", "isotropize",
           "isotropic_power_spectrum", "isotropic_cross_spectrum", "fit_loglog"]

_WINDOW_NAMES = [  # xrft.py:48-72
    "hann", "hamming", "kaiser", "tukey", "parzen", "taylor", "boxcar", "barthann", "bartlett", "blackman",
    "blackmanharris", "bohman", "chebwin", "cosine", "dpss", "exponential", "flattop", "gaussian",
    "general_cosine", "general_gaussian", "general_hamming", "triang", "nuttall",
]
_real_flag_warning = ("`real` flag will be deprecated in future version of xrft.fft and replaced by `real_dim` flag.")


# ------------------------------------------------------------------------------------------------------
# coordinate helpers (host, float64) -- xrft.py:139-155, 195-234, 269-304
# ------------------------------------------------------------------------------------------------------
# Host work that depends only on (length, spacing) or on a coordinate vector is memoised: the reference recomputes it on
# every call, here it would be most of the ~0.25 ms a call costs on the host (what small problems are made of).
_memo = {}
_memo_ids = {}  # id(memoised vector) -> its key: plans are keyed by it instead of by a hash of 32 KB of window samples

try:  # bytes -> 64-bit digest at ~10 GB/s (python's own bytes hash does 2 GB/s: 16 us per 4096-point coordinate)
    import xxhash

    def _digest(arr):
        return xxhash.xxh3_64_intdigest(np.ascontiguousarray(arr))
except Exception:  # pragma: no cover
    def _digest(arr):
        return hash(np.ascontiguousarray(arr).tobytes())


def _memoised(key, fn):
    with _plan_lock:
        hit = _memo.get(key)
    if hit is None:
        hit = fn()
        with _plan_lock:
            if len(_memo) > 512:
                _memo.clear()
                _memo_ids.clear()
            _memo[key] = hit
            if isinstance(hit, np.ndarray):
                _memo_ids[id(hit)] = key
    return hit


def _ro(a):
    a = np.asarray(a)
    a.setflags(write=False)
    return a


def _freq(N, delta_x, real, shift):
    key = ("freq", tuple(int(n) for n in N), tuple(float(d) for d in delta_x), real is not None, bool(shift))
    return list(_memoised(key, lambda: tuple(_ro(k) for k in _freq_uncached(N, delta_x, real, shift))))


def _freq_uncached(N, delta_x, real, shift):
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
        fftfreq = [np.fft.fftfreq] * (len(N) - 1)
        fftfreq.append(lambda Nx, dx: np.fft.fftfreq(2 * (Nx - 1), dx))
    k = [f(Nx, dx) for (f, Nx, dx) in zip(fftfreq, N, delta_x)]
    if shift:
        k = [np.fft.fftshift(l) for l in k]
    return k


def _stack_chunks(da, dim, suffix="_segment"):
    """xrft.py:106-136: reshape every chunked transform dimension d into (d_segment, d) -- a view, no copy."""
    newdims, newshape, newcoords = [], [], {}
    chunks = da._chunks or {}
    for d in da.dims:
        if d in dim:
            ch = chunks.get(d) or (da.sizes[d],)
            if np.diff(ch).sum() != 0:
                raise ValueError("Chunk lengths need to be the same.")
            n = da.sizes[d]
            chunklen = int(ch[0])
            coord_rs = np.asarray(da[d].values).reshape((int(n / chunklen), chunklen))
            newdims += [d + suffix, d]
            newshape += [int(n / chunklen), chunklen]
            newcoords[d + suffix] = np.arange(int(n / chunklen))
            newcoords[d] = coord_rs[0]
        else:
            newdims.append(d)
            newshape.append(da.sizes[d])
            if d in da.coords:
                newcoords[d] = da.coords[d]
    data = da.data
    if isinstance(data, torch.Tensor) and not data.is_contiguous():
        data = data.contiguous()
    return DataArray(data.reshape(newshape), newdims, newcoords, da.name, da.attrs)


def _diff_coord(coord):
    v0 = coord[0]
    if getattr(v0, "calendar", None):
        import cftime  # xrft.py:200-206; only reachable when cftime is installed

        decoded = cftime.date2num(coord, "seconds since 1800-01-01 00:00:00", v0.calendar)
        return np.diff(decoded)
    if pd.api.types.is_datetime64_dtype(v0):
        return np.diff(coord).astype("timedelta64[ns]").astype("f8") / 1e9
    return np.diff(coord)


def _lag_coord(coord):
    v0 = coord[0]
    coord_data = coord if coord[-1] > coord[0] else np.flip(coord, axis=-1)
    lag = coord_data[len(coord) // 2]
    if getattr(v0, "calendar", None):
        import cftime

        return cftime.date2num(lag, "seconds since 1800-01-01 00:00:00", v0.calendar)
    if pd.api.types.is_datetime64_dtype(v0):
        return lag.astype("timedelta64[s]").astype("f8")
    return lag


def _is_valid_fft_coord(coord):
    c0 = coord[0]
    return bool(is_numeric_dtype(coord) or is_datetime64_any_dtype(coord)
                or bool(getattr(c0.item() if hasattr(c0, "item") else c0, "calendar", False)))


def _get_coordinate_spacing(coord, spacing_tol, name):
    diff = _diff_coord(coord)
    delta = np.abs(diff[0])
    if not np.allclose(diff, diff[0], rtol=spacing_tol):
        raise ValueError("Can't take Fourier transform because coodinate %s is not evenly spaced" % name)
    if delta == 0.0:
        raise ValueError("Can't take Fourier transform because spacing in coordinate %s is zero" % name)
    return delta


def _move_to_end(lst, el):
    return [i for i in lst if i != el] + [el]


def _window_vector(window_type, n):
    if isinstance(window_type, str) and window_type in _WINDOW_NAMES:
        return _memoised(("window", window_type, int(n)), lambda: _ro(_window_vector_uncached(window_type, n)))
    return _window_vector_uncached(window_type, n)


def _window_vector_uncached(window_type, n):
    if window_type is True:  # xrft.py:42-47
        window_type = "hann"
        warnings.warn("Please provide the name of window adhering to scipy.signal.windows. The boolean option "
                      "will be deprecated in future releases.", FutureWarning)
    elif window_type not in _WINDOW_NAMES:
        raise NotImplementedError(f"Window type {window_type} not supported. Please adhere to "
                                  "scipy.signal.windows for naming convention.")
    return getattr(sps.windows, window_type)(n, sym=False)


# ------------------------------------------------------------------------------------------------------
# device plumbing
# ------------------------------------------------------------------------------------------------------
_TORCH_OK = (torch.float32, torch.float64, torch.complex64, torch.complex128)


def _to_device(data):
    dev = _lib.device()
    if isinstance(data, torch.Tensor):
        # the C library reads raw memory: materialise torch's lazy conjugate / negative views (x.conj(), x.mH)
        t = data.resolve_conj().resolve_neg()
    else:
        a = np.asarray(data)
        if a.dtype == np.float16:
            a = a.astype(np.float32)
        elif a.dtype.kind in "biu":
            a = a.astype(np.float64)  # numpy.fft promotes integers to float64
        elif a.dtype.kind not in "fc":
            raise TypeError(f"cannot transform data of dtype {a.dtype}")
        elif a.dtype.itemsize > 8 and a.dtype.kind == "f" or a.dtype.itemsize > 16:
            a = a.astype(np.complex128 if a.dtype.kind == "c" else np.float64)
        t = torch.from_numpy(np.ascontiguousarray(a))
    if t.dtype not in _TORCH_OK:
        t = t.to(torch.float64 if not t.is_complex() else torch.complex128)
    return t.to(dev)


_plan_cache: "OrderedDict[tuple, engine.SpectralPlan]" = OrderedDict()
_PLAN_CACHE_SIZE = 16
_plan_lock = threading.RLock()  # the functions are pure like the reference's: callable from several (dask-style) worker threads


_digest_ids = {}  # id(read-only array) -> (the array, its digest): tables that are built once per labelled array (phase factors,
#                   coordinate vectors) are digested once, not on every call (1 MB of phase factors per 65536-point axis: 100 us)


def _akey(a):
    if a is None:
        return None
    k = _memo_ids.get(id(a))  # a memoised window vector: identified by (name, n)
    if k is not None:
        return k
    if a.flags.writeable or not a.flags.owndata or a.base is not None:  # (a read-only VIEW of a writeable base can change under its id)
        return _digest(a)
    with _plan_lock:
        e = _digest_ids.get(id(a))
    if e is not None and e[0] is a:
        return e[1]
    d = _digest(a)
    with _plan_lock:
        if len(_digest_ids) > 256:
            _digest_ids.clear()
        _digest_ids[id(a)] = (a, d)
    return d


def _get_plan(binmap_key=None, **kw):
    # the bin map can be 16M entries: it is identified by the key of the (cached) host computation, not by its bytes
    key = (engine.bluestein_in_float64(),) + tuple((k, (binmap_key if k == "binmap" else _akey(v)) if isinstance(v, np.ndarray) else v)
                                                   for k, v in sorted(kw.items()))
    with _plan_lock:
        p = _plan_cache.get(key)
        if p is None:
            p = engine.SpectralPlan(**kw)
            _plan_cache[key] = p
            while len(_plan_cache) > _PLAN_CACHE_SIZE:
                _plan_cache.popitem(last=False)
        else:
            _plan_cache.move_to_end(key)
    return p


_TWO_STAGE = collections.OrderedDict()  # _ifft_two_stages: (shape, flags, tables) -> bool, a small LRU of DECISIONS (the stage plans live in _plan_cache only:
_TWO_STAGE_SIZE = 32                   # its LRU governs their lifetime and their device tables)


def clear_plan_cache():
    """Drop every cached plan (device tables) and the shared scratch buffers."""
    with _plan_lock:
        _plan_cache.clear()
        _BLUE_TABLES.clear()
        _TWO_STAGE.clear()
    engine.clear_workspaces()


class _Ctx:
    """Everything ``fft`` derives on the host before touching the device (xrft.py:370-433)."""


def _label_guard(da):
    """What the host analysis of a labelled array depends on: coordinate vectors are owned read-only arrays with a token that is
    issued once per assignment of ``Coordinate.values`` and never reused (labeled.Coordinate), so the same tokens mean the same
    labels.  None: do not remember anything (a coordinate array was made writeable again)."""
    toks = []
    for k, v in da.coords.items():
        vals = v.values
        if vals.flags.writeable:  # (someone re-opened the array for writing: its content is no longer pinned by the token)
            return None
        # ... plus a cheap fingerprint of the content (size, first, second and last sample: origin, spacing, extent): an array re-opened for
        # writing, changed and closed again between two calls keeps its token, and a deep copy shares it while owning another array
        n = vals.size
        fp = (n,) if n == 0 or vals.dtype.kind not in "fiuMm" else (n, vals.flat[0].item(), vals.flat[min(1, n - 1)].item(), vals.flat[n - 1].item())
        toks.append((k, v._token, fp))
    return (da.dims, da.shape, tuple(toks), None if not da._chunks else tuple(sorted(da._chunks.items())))


def _analyze(da, spacing_tol, dim, real_dim, shift, detrend, window, true_phase, chunks_to_segments, prefix, real):
    """The host side of a call: every check the reference makes, spacings, lags, frequency coordinates.  Remembered on the labelled
    array per argument set (validation, coordinate digests and frequency vectors of a 65536-point axis cost ~0.1 ms a call, more
    than the transform of a small cube): a later call with the same arguments on the same labels starts from the stored context."""
    mkey = None
    if not chunks_to_segments and real is None:
        mkey = (spacing_tol if isinstance(spacing_tol, (int, float)) else None, dim if (dim is None or isinstance(dim, str)) else tuple(dim),
                real_dim, shift, detrend, window, true_phase, prefix)
        try:
            hash(mkey)
        except TypeError:
            mkey = None
        if not isinstance(spacing_tol, (int, float)):
            mkey = None
    if mkey is not None:
        guard = _label_guard(da)
        if guard is None:
            mkey = None
    if mkey is not None:
        memo = da._memo
        if memo is not None and memo[0] == guard:
            hit = memo[1].get(mkey)
            if hit is not None:
                c = _Ctx()
                c.__dict__.update(hit.__dict__)
                c.da = da  # (kept out of the stored context: no reference cycle through the array and its device memory)
                return c
    c = _analyze_uncached(da, spacing_tol, dim, real_dim, shift, detrend, window, true_phase, chunks_to_segments, prefix, real)
    if mkey is not None and c.da is da:
        stored = _Ctx()
        stored.__dict__.update(c.__dict__)
        stored.da = None
        c._x = stored._x = {}  # shared by every copy of this context: what _execute derives from it (flags, phase tables)
        memo = da._memo
        if memo is None or memo[0] != guard or len(memo[1]) > 16:
            memo = da._memo = (guard, {})
        memo[1][mkey] = stored
    return c


def _analyze_uncached(da, spacing_tol, dim, real_dim, shift, detrend, window, true_phase, chunks_to_segments, prefix, real):
    c = _Ctx()
    c._x = None
    if dim is None:
        dim = list(da.dims)
    elif isinstance(dim, str):
        dim = [dim]
    else:
        dim = list(dim)
    if real is not None:  # xrft.py:376-378
        real_dim = real
        warnings.warn(_real_flag_warning, FutureWarning)
    if real_dim is not None:  # xrft.py:380-386
        if real_dim not in da.dims:
            raise ValueError("The dimension along which real FT is taken must be one of the existing dimensions.")
        dim = _move_to_end(dim, real_dim)
    for d in dim:
        da.get_axis_num(d)
    if not np.all([_is_valid_fft_coord(da[d].values) for d in dim]):  # xrft.py:277-281
        raise ValueError("All transformed dimensions coordinates must be numerical or datetime.")
    c.corr_N = [da.sizes[d] for d in dim]  # window_correction uses the UNsegmented lengths (xrft.py:747 -> :654)
    if chunks_to_segments:  # xrft.py:390-391
        da = _stack_chunks(da, dim)
    elif da._chunks and any(len(da._chunks.get(d, (0,))) > 1 for d in dim):
        raise ValueError("The transform dimension(s) are split into several chunks; dask.array.fft refuses that "
                         "(use chunks_to_segments=True or rechunk).")
    c.da = da
    if real_dim is not None:
        shift = False  # xrft.py:403
    c.rawdims = da.dims
    c.dim, c.real_dim, c.shift = dim, real_dim, bool(shift)
    c.N = [da.sizes[d] for d in dim]
    for d in dim:  # xrft.py:412-420
        bad = [cn for cn, cv in da.coords.items() if cn != d and d in cv.dims]
        if bad:
            raise ValueError(f"The input array contains coordinate variable(s) ({bad}) whose dims include the "
                             f"transform dimension(s) `{d}`. Please drop these coordinates (`.drop({bad}`) "
                             "before invoking xrft.")
    def _coord_info(d):  # spacing check + lag of one coordinate vector, memoised on its bytes
        cv = np.asarray(da[d].values)
        if cv.dtype.kind not in "fiu" or not isinstance(spacing_tol, (int, float)):  # (a bad spacing_tol must fail in numpy, as in the reference)
            return _get_coordinate_spacing(cv, spacing_tol, d), _lag_coord(cv)
        key = ("coord", cv.dtype.str, cv.size, _akey(cv), float(spacing_tol))
        return _memoised(key, lambda: (_get_coordinate_spacing(cv, spacing_tol, d), _lag_coord(cv)))

    info = [_coord_info(d) for d in dim]
    c.delta_x = [i[0] for i in info]  # xrft.py:422
    c.lag_x = [i[1] for i in info]  # xrft.py:423
    if detrend not in (None, "constant", "linear"):  # detrend.py:46-50
        raise NotImplementedError("%s is not a valid detrending option. Valid options are: 'constant','linear', "
                                  "or None." % detrend)
    if len(dim) > 2 or len(dim) == 0:
        raise NotImplementedError("xrft_amd transforms one or two dimensions per call (the reference's N-D fftn / "
                                  "3-D linear detrend are outside the MI355X hot path, SURVEY.md 8f).")
    c.detrend = {None: _lib.DETREND_NONE, "constant": _lib.DETREND_CONSTANT, "linear": _lib.DETREND_LINEAR}[detrend]
    c.windows = None if window is None else [_window_vector(window, n) for n in c.N]
    c.true_phase = bool(true_phase)
    c.reversed = [bool(da[d].values[-1] < da[d].values[0]) for d in dim] if true_phase else [False] * len(dim)
    # device axis roles: x = last listed dim when real (xrft.py:395-396), else the one stored last in memory
    if len(dim) == 1:
        c.ydim, c.xdim = None, dim[0]
    elif real_dim is not None:
        c.ydim, c.xdim = dim[0], dim[1]
    else:
        a0, a1 = da.get_axis_num(dim[0]), da.get_axis_num(dim[1])
        c.ydim, c.xdim = (dim[0], dim[1]) if a0 < a1 else (dim[1], dim[0])
    c.k = _freq(c.N, c.delta_x, real_dim, c.shift)  # xrft.py:449
    c.k_unshifted = _freq(c.N, c.delta_x, real_dim, False)
    c.prefix = prefix
    c.swap = OrderedDict()
    c.new_coords = {}
    for d, kk in zip(dim, c.k):  # xrft.py:178-192
        new_name = prefix + d if d[: len(prefix)] != prefix else d[len(prefix):]
        c.swap[d] = new_name
        c.new_coords[new_name] = Coordinate((new_name,), kk, {"spacing": kk[1] - kk[0]} if len(kk) > 1 else {}, new_name)
    return c


def _flags_tables(c, da, other_lag=None, other_reversed=None):
    """Engine flags, window vectors and phase tables per device axis (y, x).  ``other_reversed``: cross spectra -- the
    reference flips each field by its own coordinate (xrft.py:436-441): ``c`` is field 0 (FLIP0_*), the other field 1.
    Remembered on a stored context (the phase factors of a 65536-point axis are 0.8 ms of numpy per call)."""
    x = getattr(c, "_x", None)
    if x is None:
        return _flags_tables_uncached(c, other_lag, other_reversed)
    key = ("ft", None if other_lag is None else tuple(float(v) for v in other_lag), None if other_reversed is None else tuple(other_reversed))
    hit = x.get(key)
    if hit is None:
        flags, win, ph = _flags_tables_uncached(c, other_lag, other_reversed)
        for v in ph.values():
            if v is not None:
                v.setflags(write=False)
        hit = x[key] = (flags, win, ph)
    return hit[0], dict(hit[1]), dict(hit[2])


def _flags_tables_uncached(c, other_lag, other_reversed):
    flags = 0
    win = {"y": None, "x": None}
    ph = {"y": None, "x": None}
    for i, d in enumerate(c.dim):
        ax = "x" if d == c.xdim else "y"
        if c.shift:
            flags |= _lib.SHIFT_X if ax == "x" else _lib.SHIFT_Y
        if c.true_phase:
            flags |= _lib.ISHIFT_X if ax == "x" else _lib.ISHIFT_Y
            if other_reversed is None:
                if c.reversed[i]:
                    flags |= _lib.FLIP_X if ax == "x" else _lib.FLIP_Y
            else:
                if c.reversed[i]:
                    flags |= _lib.FLIP0_X if ax == "x" else _lib.FLIP0_Y
                if other_reversed[i]:
                    flags |= _lib.FLIP_X if ax == "x" else _lib.FLIP_Y
            # xrft.py:462-469 -- indexed by unshifted frequency; the real axis uses rfftfreq on its kept half
            n = c.N[i]
            f = np.fft.fftfreq(n, c.delta_x[i])
            if c.real_dim is not None and d == c.real_dim:
                f[: n // 2 + 1] = np.fft.rfftfreq(n, c.delta_x[i])
            p = np.exp(-1j * 2.0 * np.pi * f * c.lag_x[i])
            if other_lag is not None:  # cross spectrum: F1 phase * conj(F2 phase)
                p = p * np.conj(np.exp(-1j * 2.0 * np.pi * f * other_lag[i]))
            ph[ax] = p
        if c.windows is not None:
            win[ax] = c.windows[i]
    if c.real_dim is not None:
        flags |= _lib.HALF_X
    return flags, win, ph


def _arrange(c, da):
    """Device tensor with the transform axes last: shape (*other, [ny,] nx); returns (tensor, other_dims)."""
    t = _to_device(da.data)
    tdims = ([c.ydim] if c.ydim is not None else []) + [c.xdim]
    other = [d for d in da.dims if d not in tdims]
    order = other + tdims
    if tuple(order) != tuple(da.dims):
        t = t.permute([da.get_axis_num(d) for d in order])
    return t.contiguous(), other


def _label_output(c, da, out_t, other, extra_cattrs=None, drop_transform=False):
    """Wrap the engine output (other..., ky, kx) as a DataArray in the reference's dim order (xrft.py:451-476).
    ``other is None``: the output already has the input's dim order (in-place axis transform)."""
    tdims = ([c.ydim] if c.ydim is not None else []) + [c.xdim]
    final = [c.swap.get(d, d) for d in c.rawdims]
    if other is not None:
        cur = other + [c.swap[d] for d in tdims]
        out_t = out_t.reshape([da.sizes[d] for d in other] + list(out_t.shape[-len(tdims):]))
        if cur != final:
            out_t = out_t.permute([cur.index(d) for d in final])
    coords = {k: v._clone(k) for k, v in da.coords.items() if k not in c.dim}  # (own objects over the input's immutable values)
    for name, cv in c.new_coords.items():
        attrs = dict(cv.attrs)
        if extra_cattrs and name in extra_cattrs:
            attrs.update(extra_cattrs[name])
        coords[name] = Coordinate(cv.dims, cv.values, attrs, name)
    if any(d in c.dim for k, v in da.coords.items() if k not in c.dim for d in v.dims):
        return DataArray(out_t, final, coords, None, None)  # (a kept coordinate that spans a transformed dim: the validating constructor's error, as before)
    return DataArray._trusted(out_t, final, coords)


def _inplace_axis(c, da, iso):
    """Axis number k if the call is a single-axis transform along a middle or first axis that the engine can do where the
    axis lies (XRFTHIP_AXIS_Y: the array is (batch, n, inner) with no transposed copy, like the reference, xrft.py:395-409)."""
    if len(c.dim) != 1 or iso is not None:  # (real_dim along the axis: the half output of the one-pass kernels, ABI 0.1.4)
        return None
    k = da.get_axis_num(c.xdim)
    return k if k != len(da.dims) - 1 else None


def _execute_axis_y(c, da, mode, scale, k, da2=None, c2=None, extra_flags=0):
    """Single transform axis k < last: (batch, ny, nx) = (prod(shape[:k]), shape[k], prod(shape[k+1:])), y transformed in
    place.  Returns the output tensor in the ORIGINAL dim order, or None if the plan cannot be built (column too long
    for one LDS tile): the caller then takes the transposing path."""
    t = _to_device(da.data).contiguous()  # C-contiguous input: no copy
    shape = list(t.shape)
    ny = shape[k]
    nx = int(np.prod(shape[k + 1:], dtype=np.int64))
    batch = int(np.prod(shape[:k], dtype=np.int64))
    flags, win, ph = _flags_tables(c, da, None if c2 is None else c2.lag_x, None if c2 is None else c2.reversed)
    if mode == _lib.OUT_POWER:
        ph = {"y": None, "x": None}
    yflags = _lib.AXIS_Y | (flags & _lib.HALF_X) | extra_flags  # (HALF_X / REALDIM_X2 with AXIS_Y: along the transformed axis)
    if (yflags & _lib.HALF_X) and (t.is_complex() or (flags & (_lib.FLIP_X | _lib.FLIP0_X))):
        return None  # (the half output exists for real input in coordinate order only: the transposing path)
    for fx, fy in ((_lib.SHIFT_X, _lib.SHIFT_Y), (_lib.ISHIFT_X, _lib.ISHIFT_Y), (_lib.FLIP_X, _lib.FLIP_Y), (_lib.FLIP0_X, _lib.FLIP0_Y)):
        if flags & fx:
            yflags |= fy
    t2 = None
    if da2 is not None:
        t2 = _to_device(da2.data).contiguous()
        if tuple(da2.dims) != tuple(da.dims) or t2.shape != t.shape:
            raise ValueError("The two datasets have different dimensions")
        if t2.dtype != t.dtype:
            dt = torch.promote_types(t.dtype, t2.dtype)
            t, t2 = t.to(dt), t2.to(dt)
    kw = dict(ndim=2, batch=batch, ny=ny, nx=nx, dtype=t.dtype, out_mode=mode, detrend=c.detrend, flags=yflags,
              scale=float(scale), window_y=win["x"], window_x=None, phase_y=ph["x"], phase_x=None)
    if ny * nx > (1 << 31) - 1 or nx > (1 << 30) or ny > (1 << 30):
        return None  # the engine indexes a slab with 32 bits: a cube this large along a first / middle axis takes the transposing path
    try:
        plan = _get_plan(**kw)
    except _lib.XrftHipError as e:
        if e.status == _lib.UNSUPPORTED_LENGTH:
            return None
        raise
    out, _ = plan.execute(t.reshape(batch, ny, nx), None if t2 is None else t2.reshape(batch, ny, nx))
    shape[k] = plan.ny_out
    return out.reshape(shape)


def _swap_xy(flags):
    """The per-axis flag bits with the roles of x and y exchanged."""
    out = flags & ~(_lib.SHIFT_X | _lib.SHIFT_Y | _lib.ISHIFT_X | _lib.ISHIFT_Y | _lib.FLIP_X | _lib.FLIP_Y | _lib.HALF_X | _lib.HALF_Y)
    for fx, fy in ((_lib.SHIFT_X, _lib.SHIFT_Y), (_lib.ISHIFT_X, _lib.ISHIFT_Y), (_lib.FLIP_X, _lib.FLIP_Y), (_lib.HALF_X, _lib.HALF_Y)):
        if flags & fx:
            out |= fy
        if flags & fy:
            out |= fx
    return out


def _execute_inner(c, da, mode, scale, extra_flags=0, da2=None, c2=None):
    """Two transform axes that are not the trailing pair, wherever they lie -- dim = ["y", "x"] of a (y, x, time) array, dim = ["t", "x"] of a (t, y, x) array:
    the engine's layout [batch][n0][mid][n1][inner] (xrfthip_desc.inner, .mid: the products of the extents in front of, between and behind the two axes)
    transforms them where they lie, as the reference does (xrft.py:395-409) -- no transposed copy of the input or of the result.  Returns the result in
    the input's dim order, or None when the call is not of this kind (the caller then takes the transposing path)."""
    if len(c.dim) != 2 or mode not in (_lib.OUT_COMPLEX, _lib.OUT_POWER, _lib.OUT_CROSS):
        return None
    p, q = da.get_axis_num(c.ydim), da.get_axis_num(c.xdim)
    first, second = min(p, q), max(p, q)
    if second == first + 1 and second == len(da.dims) - 1:
        return None  # (the trailing pair: the fused two-axis plans)
    t = _to_device(da.data).contiguous()  # C-contiguous input: no copy
    shape = list(t.shape)
    inner = int(np.prod(shape[second + 1:], dtype=np.int64))
    mid = int(np.prod(shape[first + 1:second], dtype=np.int64))
    batch = int(np.prod(shape[:first], dtype=np.int64))
    # the extents the composite plan carries in 32 bits (create_inner_plan; the one-axis stages): known limits are checked HERE,
    # so that a BAD_ARG from the library means a bug in the descriptor and is raised, not hidden behind the transposing path
    if (inner < 2 and mid < 2) or mid * inner * shape[second] > (1 << 30) or shape[first] * shape[second] * inner * mid > (1 << 31) - 1:
        return None
    t2 = None
    if mode == _lib.OUT_CROSS:  # (round 6: the cross spectrum of two real fields on the fused passes -- both fields where they lie)
        if da2 is None or tuple(da2.dims) != tuple(da.dims):
            return None
        t2 = _to_device(da2.data).contiguous()
        if t2.shape != t.shape or t2.dtype != t.dtype or t.is_complex():
            return None
    flags, win, ph = _flags_tables(c, da, None if c2 is None else c2.lag_x, None if c2 is None else c2.reversed)
    flags |= extra_flags  # (REALDIM_X2: the kept half of the real axis counts twice in a power spectrum)
    if mode == _lib.OUT_POWER:
        ph = {"y": None, "x": None}
    if p > q:  # the array holds (x, y): the plan's first axis is the one that comes first in memory
        flags, win, ph = _swap_xy(flags), {"y": win["x"], "x": win["y"]}, {"y": ph["x"], "x": ph["y"]}
    kw = dict(ndim=2, batch=batch, ny=shape[first], nx=shape[second], inner=inner, mid=mid, dtype=t.dtype, out_mode=mode, detrend=c.detrend, flags=flags,
              scale=float(scale), window_y=win["y"], window_x=win["x"], phase_y=ph["y"], phase_x=ph["x"])
    try:
        plan = _get_plan(**kw)
    except _lib.XrftHipError as e:
        if e.status == _lib.UNSUPPORTED_LENGTH:  # a length the one-axis plans do not take: the transposing path
            return None
        raise
    out, _ = plan.execute(t, t2)
    shape[first], shape[second] = plan.ny_out, plan.nx_out  # (real_dim: n / 2 + 1 samples along that axis -- the second of the pair in memory, or the first: XRFTHIP_HALF_Y)
    return out.reshape(shape)


def _execute(c, da, mode, scale, da2=None, c2=None, iso=None, extra_flags=0):
    k = _inplace_axis(c, da, iso) if (extra_flags & ~_lib.REALDIM_X2) == 0 else None
    if k is not None:
        out = _execute_axis_y(c, da, mode, scale, k, da2, c2, extra_flags)
        if out is not None:
            return out, None, None  # other = None: the output has the input's dim order
    if (extra_flags & ~_lib.REALDIM_X2) == 0 and iso is None and (da2 is None or mode == _lib.OUT_CROSS):
        out = _execute_inner(c, da, mode, scale, extra_flags, da2, c2)
        if out is not None:
            return out, None, None
    if extra_flags == 0 and iso is None and da2 is not None and mode == _lib.OUT_PHASE:
        # the cross phase of two fields on non-trailing axes: the fused cross spectrum where the axes lie, then its angle (xrft.py:871-874) -- no transposed copy
        out = _execute_inner(c, da, _lib.OUT_CROSS, scale, 0, da2, c2)
        if out is not None:
            return engine.angle(out), None, None
    t, other = _arrange(c, da)
    ndim = len(c.dim)
    nx = da.sizes[c.xdim]
    ny = da.sizes[c.ydim] if c.ydim is not None else 1
    batch = t.numel() // max(ny * nx, 1)
    flags, win, ph = _flags_tables(c, da, None if c2 is None else c2.lag_x, None if c2 is None else c2.reversed)
    if mode == _lib.OUT_POWER:
        ph = {"y": None, "x": None}
    flags |= extra_flags
    t2 = None
    if da2 is not None:
        t2, other2 = _arrange(c2, da2)
        if t2.shape != t.shape or other2 != other:
            raise ValueError("The two datasets have different dimensions")
        if t2.dtype != t.dtype:
            dt = torch.promote_types(t.dtype, t2.dtype)
            t, t2 = t.to(dt), t2.to(dt)
    kw = dict(ndim=ndim, batch=batch, ny=ny, nx=nx, dtype=t.dtype, out_mode=mode, detrend=c.detrend, flags=flags,
              scale=float(scale), window_y=win["y"], window_x=win["x"], phase_y=ph["y"], phase_x=ph["x"])
    bkey = None
    if iso is not None:
        kw.update(binmap=iso["binmap"], nbins=iso["nbins"])
        bkey = iso.get("binmap_key")
    try:
        plan = _get_plan(binmap_key=bkey, **kw)
    except _lib.XrftHipError as e:
        if e.status != _lib.UNSUPPORTED_LENGTH:
            raise
        # numpy.fft takes any length; here a prime factor above 128 goes through Bluestein inside one LDS tile, which bounds
        # the length (2 n - 1 rounded up to 2^a 3^b 5^c complex samples must fit 160 KB)
        if ndim == 1 and iso is None and t2 is None and mode in (_lib.OUT_COMPLEX, _lib.OUT_POWER) and not (flags & (_lib.INVERSE | _lib.C2R_X | _lib.PHASE_IN)):
            return _bluestein_1d(t, nx, mode, c.detrend, flags, scale, win["x"], ph["x"]), None, other
        lens = {d: da.sizes[d] for d in c.dim}
        lim = 8800 if t.dtype in (torch.float32, torch.complex64) else 4400
        raise _UnsupportedLength(f"transform length(s) {lens} not supported on the device: a length with a prime factor above 128 must be "
                         f"<= ~{lim} samples for {str(t.dtype).replace('torch.', '')} data (Bluestein inside one LDS tile) unless it is the only "
                         f"transform axis; transform the axes one at a time, or pad / crop the axis (e.g. xrft_amd.pad) to a smooth length") from e
    out, iso_out = plan.execute(t, t2)
    return out, iso_out, other


# ------------------------------------------------------------------------------------------------------
# Bluestein's algorithm through global memory: one transform axis (the last) whose length has a prime factor above 128 and exceeds
# what the in-tile Bluestein of the C ABI holds (numpy.fft takes any length, xrft.py:398-447).
#   X[k] = conj(c[k]) sum_j (x[j] conj(c[j])) c[k - j],   c[j] = exp(i pi j^2 / n)
# = chirp multiply with zero padding to a 2^a 3^b 5^c length m >= 2n - 1, a forward plan, the product with FFT_m(c wrapped) / m,
# an inverse plan, chirp multiply with truncation -- three table-multiply launches (xrfthip_table_mul) around two existing plans;
# detrend, window, flips and shifts, true-phase factors, scaling and |F|^2 are the same device calls the other paths use.
# ------------------------------------------------------------------------------------------------------
def _wide(da):
    """float32 data of a call that is composed of several device passes around a Bluestein axis (no two-axis plan exists for a long
    prime length): the whole composition runs in float64 -- detrending pass, chirp convolution, the other axis -- between two precision
    changes, so that its small bins hold the 1e-3 every fused path holds (stored float32 intermediates cost them 1.3e-3)."""
    t = _to_device(da.data)
    if t.dtype not in (torch.float32, torch.complex64) or not engine.bluestein_in_float64():
        return da, False
    w = engine.convert(t.contiguous(), torch.float64 if t.dtype == torch.float32 else torch.complex128)
    return DataArray(w, da.dims, da.coords, da.name, da.attrs), True


def _narrow(res, was_wide):
    if not was_wide or not isinstance(res.data, torch.Tensor) or res.data.dtype not in (torch.float64, torch.complex128):
        return res
    d = res.data
    n = engine.convert(d.contiguous(), torch.float32 if d.dtype == torch.float64 else torch.complex64)
    return DataArray(n, res.dims, res.coords, res.name, res.attrs)


class _UnsupportedLength(ValueError):
    """A two-axis plan could not be built for these lengths: the callers transform the axes one at a time instead."""


_BLUE_TABLES = {}


def _blue_tables(n, cdt, dev, inverse=False):
    """(m, conj chirp as a HOST complex128 vector, FFT_m(chirp) / m on the device) of Bluestein's algorithm for length n, cached.
    The chirp's spectrum is taken by the engine's own length-m plan in complex128 (no numpy.fft in the product path)."""
    key = (n, str(cdt), str(dev), inverse)
    with _plan_lock:
        tb = _BLUE_TABLES.get(key)
    if tb is None:
        m = 2 * n - 1
        while True:
            q = m
            for p in (2, 3, 5):
                while q % p == 0:
                    q //= p
            if q == 1:
                break
            m += 1
        j = np.arange(n, dtype=np.int64)
        ang = np.pi * ((j * j) % (2 * n)).astype(np.float64) / n  # j^2 mod 2n: exact phases for long sequences
        chirp = np.exp((-1j if inverse else 1j) * ang)             # c[j] (conjugated for the inverse transform)
        b = np.zeros(m, dtype=np.complex128)
        b[:n] = chirp
        b[m - n + 1:] = chirp[1:][::-1]                            # c[-j] = c[j], wrapped
        plan = _get_plan(ndim=1, batch=1, ny=1, nx=m, dtype=torch.complex128, out_mode=_lib.OUT_COMPLEX, detrend=_lib.DETREND_NONE, flags=0,
                         scale=1.0 / m, window_y=None, window_x=None, phase_y=None, phase_x=None)
        bhat, _ = plan.execute(torch.from_numpy(b).to(dev).reshape(1, 1, m))
        tb = (m, np.conj(chirp), bhat.reshape(m).to(cdt))  # (one precision change of a table, once per length)
        with _plan_lock:
            if len(_BLUE_TABLES) > 8:
                _BLUE_TABLES.clear()
            _BLUE_TABLES[key] = tb
    return tb


def _bluestein_1d(t, n, mode, detrend_kind, flags, scale, win, ph, phase_in=None):
    """The 1-D plan's result for t[..., n] (real or complex) without a 1-D plan of length n.  With XRFTHIP_INVERSE in ``flags`` the
    unnormalised inverse transform (the same pipeline with conjugated chirps); ``phase_in`` multiplies the INPUT, indexed by source
    position (XRFTHIP_PHASE_IN, xrft.py:574-576).  float32 data run in float64 between two precision changes (engine.convert;
    engine.bluestein_in_float64): stored float32 intermediates between the five passes cost the small bins their 1e-3.  (Bluestein
    inside one tile of the C ABI's plans stays in float32: 1.2e-4 per bin.)"""
    if t.dtype in (torch.float32, torch.complex64) and engine.bluestein_in_float64():
        X = _bluestein_1d(engine.convert(t, torch.float64 if t.dtype == torch.float32 else torch.complex128), n, mode, detrend_kind, flags,
                          scale, win, ph, phase_in)
        return engine.convert(X, torch.float32 if X.dtype == torch.float64 else torch.complex64)
    shape = list(t.shape)
    x = t.reshape(-1, n).contiguous()
    real_in = not x.is_complex()
    cdt = torch.complex64 if x.dtype in (torch.float32, torch.complex64) else torch.complex128
    if detrend_kind != _lib.DETREND_NONE:  # per row, before the window (xrft.py:425-433)
        x = engine.detrend(x, 1, detrend_kind)
    idx = np.arange(n)
    if flags & _lib.FLIP_X:      # np.flip, then ifftshift (xrft.py:436-441)
        idx = idx[::-1]
    if flags & _lib.ISHIFT_X:
        idx = np.roll(idx, -(n // 2))
    if (flags & (_lib.FLIP_X | _lib.ISHIFT_X)):
        x = engine.gather_axis(x, 1, index=idx)
    m, cconj_chirp, bhat = _blue_tables(n, cdt, x.device, inverse=bool(flags & _lib.INVERSE))
    # the pointwise tables are composed on the host (length-n vectors, like the window vectors) and uploaded: no torch arithmetic
    tab_h = cconj_chirp
    if phase_in is not None:
        tab_h = tab_h * np.asarray(phase_in, dtype=np.complex128)[idx]
    if win is not None:  # the window rides on the first chirp multiply (it multiplies the samples where they lie AFTER flip / shift:
        tab_h = tab_h * np.asarray(win, dtype=np.float64)[idx]  # the reference windows first, then flips: window of the source sample)
    tab = torch.from_numpy(np.ascontiguousarray(tab_h.astype(np.complex64 if cdt == torch.complex64 else np.complex128))).to(x.device)
    a = engine.table_mul(x, tab.contiguous(), m)
    fwd = _get_plan(ndim=1, batch=a.shape[0], ny=1, nx=m, dtype=a.dtype, out_mode=_lib.OUT_COMPLEX, detrend=_lib.DETREND_NONE, flags=0, scale=1.0,
                    window_y=None, window_x=None, phase_y=None, phase_x=None)
    A, _ = fwd.execute(a.reshape(a.shape[0], 1, m))
    C = engine.table_mul(A.reshape(-1, m), bhat, m)
    inv = _get_plan(ndim=1, batch=a.shape[0], ny=1, nx=m, dtype=a.dtype, out_mode=_lib.OUT_COMPLEX, detrend=_lib.DETREND_NONE, flags=_lib.INVERSE, scale=1.0,
                    window_y=None, window_x=None, phase_y=None, phase_x=None)
    cc, _ = inv.execute(C.reshape(-1, 1, m))
    half = bool(flags & _lib.HALF_X)
    n_out = n // 2 + 1 if half else n
    fac = np.ones(n_out, dtype=np.complex128)
    if mode == _lib.OUT_COMPLEX:
        fac = fac * float(scale)
        if ph is not None:
            fac = fac * np.asarray(ph, dtype=np.complex128)[:n_out]
    tab2 = torch.from_numpy(np.ascontiguousarray((cconj_chirp[:n_out] * fac).astype(np.complex64 if cdt == torch.complex64 else np.complex128))).to(x.device)
    X = engine.table_mul(cc.reshape(-1, m), tab2, n_out)  # F[k] (x phase x scale), k < n_out
    if mode == _lib.OUT_POWER:
        if flags & _lib.REALDIM_X2:
            X = engine.spectrum_tail_axis(X, None, float(scale), 1, n % 2 == 0)
        else:
            X = engine.spectrum_tail(X, None, float(scale))
    elif flags & _lib.REALDIM_X2:
        d2 = np.full(n_out, 2.0); d2[0] = 1.0
        if n % 2 == 0:
            d2[-1] = 1.0
        X = engine.table_mul(X, torch.from_numpy(d2.astype(np.complex64 if cdt == torch.complex64 else np.complex128)).to(x.device), n_out)
    if flags & _lib.SHIFT_X:
        X = engine.gather_axis(X, 1, roll=n // 2)
    return X.reshape(shape[:-1] + [n_out])


# ------------------------------------------------------------------------------------------------------
# more than two transform axes: numpy's fftn is separable, so the reference's N-D transform
# (xrft.py:439-447) is the fused two-axis plan over the last two listed dims followed by one- or two-axis plans over
# the remaining ones; detrending needs the whole block and runs first as its own device pass; windows, shifts, phase
# factors and the dx amplitude factor are per-axis and travel with the stage that transforms the axis.
# ------------------------------------------------------------------------------------------------------
def _nd_dims(da, dim, real_dim, real):
    """The normalised dim list if it has more than two entries, else None."""
    if dim is None:
        dims = list(da.dims)
    elif isinstance(dim, str):
        return None
    else:
        dims = list(dim)
    if len(dims) <= 2:
        return None
    rd = real if real is not None else real_dim
    if rd is not None:
        if rd not in da.dims:
            raise ValueError("The dimension along which real FT is taken must be one of the existing dimensions.")
        dims = _move_to_end(dims, rd)
    for d in dims:
        da.get_axis_num(d)
    return dims


def _two_dims(da, dim, real_dim, real):
    """The normalised list of two transform dims (the real one last), for the one-axis-at-a-time fallback."""
    dims = list(da.dims) if dim is None else ([dim] if isinstance(dim, str) else list(dim))
    rd = real if real is not None else real_dim
    return _move_to_end(dims, rd) if rd is not None else dims


def _fft_nd(da, dims, spacing_tol, real_dim, shift, detrend_, window, true_phase, true_amplitude, chunks_to_segments,
            prefix, one_at_a_time=False):
    if chunks_to_segments:
        raise NotImplementedError("chunks_to_segments is implemented for one or two transform dimensions.")
    if detrend_ not in (None, "constant", "linear"):
        raise NotImplementedError("%s is not a valid detrending option. Valid options are: 'constant','linear', "
                                  "or None." % detrend_)
    cur = da
    if detrend_ is not None:
        cur = from_any(detrend(cur, dims, detrend_))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", FutureWarning)
        g = 1 if one_at_a_time else 2  # (one axis per stage: a length no two-axis plan takes goes through the one-axis paths, which take any)
        cur = fft(cur, spacing_tol=spacing_tol, dim=dims[-g:], real_dim=real_dim, shift=shift, detrend=None,
                  window=window, true_phase=true_phase, true_amplitude=true_amplitude, prefix=prefix)
        rest = dims[:-g]
        sh = False if real_dim is not None else shift  # xrft.py:403: a real transform switches every shift off
        while rest:
            grp, rest = rest[-g:], rest[:-g]
            cur = fft(cur, spacing_tol=spacing_tol, dim=grp, shift=sh, detrend=None, window=window,
                      true_phase=true_phase, true_amplitude=true_amplitude, prefix=prefix)
    return cur


def _spectrum_nd(da, da2, dims, real_dim, scaling, window_correction, true_phase, kwargs, one_at_a_time=False):
    """power_spectrum / cross_spectrum over more than two axes: N-D transform(s), then the elementwise tail."""
    if "density" in kwargs:
        scaling = "density" if kwargs.pop("density") else "false_density"
    if kwargs.get("real") is not None:
        real_dim = kwargs.get("real")
    for k in ("true_amplitude", "true_phase"):
        kwargs.pop(k, None)
    kw = dict(spacing_tol=1e-3, shift=True, detrend=None, window=None, chunks_to_segments=False, prefix="freq_")
    unknown = set(kwargs) - set(kw) - {"real"}
    if unknown:
        raise TypeError(f"fft() got an unexpected keyword argument {sorted(unknown)[0]!r}")
    kw.update({k: v for k, v in kwargs.items() if k != "real"})
    f1 = _fft_nd(da, dims, kw["spacing_tol"], real_dim, kw["shift"], kw["detrend"], kw["window"], true_phase, True,
                 kw["chunks_to_segments"], kw["prefix"], one_at_a_time)
    f2 = None
    if da2 is not None:
        f2 = _fft_nd(da2, dims, kw["spacing_tol"], real_dim, kw["shift"], kw["detrend"], kw["window"], true_phase, True,
                     kw["chunks_to_segments"], kw["prefix"], one_at_a_time)
        if tuple(f1.dims) != tuple(f2.dims):
            raise ValueError("The two datasets have different dimensions")
    pf = kw["prefix"]
    new = [pf + d if d[: len(pf)] != pf else d[len(pf):] for d in dims]  # xrft.py:186
    scale = 1.0
    if scaling != "false_density":
        if window_correction:
            if kw["window"] is None:
                raise ValueError("window_correction can only be applied when windowing is turned on.")
            vecs = [_window_vector(kw["window"], da.sizes[d]) for d in dims]
            if scaling == "density":
                scale /= float(np.prod([(v ** 2).mean() for v in vecs]))
            elif scaling == "spectrum":
                scale /= float(np.prod([v.mean() for v in vecs])) ** 2
            else:
                raise ValueError("Unknown {} scaling flag".format(scaling))
        fs = float(math.prod([float(f1[n].attrs["spacing"]) for n in new]))
        if scaling == "density":
            scale *= fs
        elif scaling == "spectrum":
            scale *= fs ** 2
        else:
            raise ValueError("Unknown {} scaling flag".format(scaling))
    a = _to_device(f1.data).contiguous()
    b = None
    if f2 is not None:
        b = _to_device(f2.data).contiguous()
        if b.dtype != a.dtype:
            dt = torch.promote_types(a.dtype, b.dtype)
            a, b = a.to(dt), b.to(dt)
    if real_dim is not None:  # xrft.py:673-682: the kept half of the real axis counts twice, except k = 0 and Nyquist
        out = engine.spectrum_tail_axis(a, b, scale, f1.get_axis_num(new[-1]), da.sizes[real_dim] % 2 == 0)
    else:
        out = engine.spectrum_tail(a, b, scale)
    return DataArray(out, f1.dims, f1.coords, None, None)


# ------------------------------------------------------------------------------------------------------
# public API
# ------------------------------------------------------------------------------------------------------
def fft(da, spacing_tol=1e-3, dim=None, real_dim=None, shift=True, detrend=None, window=None, true_phase=True,
        true_amplitude=True, chunks_to_segments=False, prefix="freq_", real=None):
    """Discrete Fourier transform of ``da`` along ``dim`` (reference: xrft/xrft.py:307-476; same arguments)."""
    src = da
    da = from_any(da)
    nd = _nd_dims(da, dim, real_dim, real)
    if nd is not None:
        return to_like(_fft_nd(da, nd, spacing_tol, real_dim if real is None else real, shift, detrend, window,
                               true_phase, true_amplitude, chunks_to_segments, prefix), src)
    c = _analyze(da, spacing_tol, dim, real_dim, shift, detrend, window, true_phase, chunks_to_segments, prefix, real)
    scale = math.prod(c.delta_x) if true_amplitude else 1.0  # xrft.py:471-472 (math.prod: the same left-to-right product as np.prod of a short list, a tenth of its call time)
    try:
        out, _, other = _execute(c, c.da, _lib.OUT_COMPLEX, scale)
    except _UnsupportedLength:
        if chunks_to_segments:
            raise
        daw, wide = _wide(da)
        return to_like(_narrow(_fft_nd(daw, _two_dims(da, dim, real_dim, real), spacing_tol, real_dim if real is None else real, shift, detrend, window,
                                       true_phase, true_amplitude, False, prefix, one_at_a_time=True), wide), src)
    da = c.da
    extra = None
    if c.true_phase:  # xrft.py:469
        extra = {c.swap[d]: {"direct_lag": lag} for d, lag in zip(c.dim, c.lag_x)}
    return to_like(_label_output(c, da, out, other, extra), src)


def dft(da, dim=None, true_phase=False, true_amplitude=False, **kwargs):
    """Deprecated alias of ``fft`` with numpy-like defaults (xrft/xrft.py:237-250)."""
    warnings.warn("This function has been renamed and will disappear in the future. Please use `fft` instead",
                  FutureWarning)
    return fft(da, dim=dim, true_phase=true_phase, true_amplitude=true_amplitude, **kwargs)


def _ifft_host(daft, dim, lag, real_dim, shift, true_phase, spacing_tol, prefix):
    """The host side of an inverse transform (xrft.py:574-576, 598-623): the input phase factors by SOURCE position, the index map that sorting + ifftshift amount to,
    spacing and centring checks, the lag coordinates of the result.  Pure in the labels and the arguments: ifft remembers it per labelled array."""
    phase = {d: (_ro(np.exp(1j * 2.0 * np.pi * np.asarray(daft[d].values, dtype=np.float64) * l)) if true_phase else None)
             for d, l in zip(dim, lag)}
    N = [daft.sizes[d] for d in dim]
    # sortby(dim) + ifftshift (xrft.py:598, 612-614) expressed as an index map of the engine: ascending coordinates ->
    # ISHIFT, descending -> FLIP + ISHIFT, the unshifted layout (fftshift(sort) == identity) -> nothing
    coords_sorted, maps = {}, {}
    for d, n in zip(dim, N):
        cv = np.asarray(daft[d].values)
        order = np.argsort(cv, kind="stable")
        coords_sorted[d] = cv[order]
        if d == real_dim:
            if not np.array_equal(order, np.arange(n)):
                raise ValueError("the real dimension's frequency coordinate must be ascending (rfftfreq order)")
            maps[d] = "none"
            continue
        want = order[(np.arange(n) + n // 2) % n]  # source index feeding unshifted position m
        if np.array_equal(want, np.arange(n)):
            maps[d] = "none"
        elif np.array_equal(order, np.arange(n)):
            maps[d] = "ishift"
        elif np.array_equal(order, np.arange(n)[::-1]):
            maps[d] = "flip_ishift"
        else:
            maps[d] = want  # arbitrary permutation: gather on the device first
    delta_x = [_get_coordinate_spacing(coords_sorted[d], spacing_tol, d) for d in dim]
    for d in dim:  # xrft.py:600-606
        l = _lag_coord(coords_sorted[d]) if d is not real_dim else coords_sorted[d][0]
        if np.abs(l) > spacing_tol:
            raise ValueError("Inverse Fourier Transform can not be computed because coordinate %s is not centered on "
                             "zero frequency" % d)
    k = _ifreq(N, delta_x, real_dim, shift)  # xrft.py:623
    swap, new_coords = OrderedDict(), {}
    for d, kk in zip(dim, k):
        new_name = prefix + d if d[: len(prefix)] != prefix else d[len(prefix):]
        swap[d] = new_name
        new_coords[new_name] = Coordinate((new_name,), kk, {"spacing": kk[1] - kk[0]} if len(kk) > 1 else {}, new_name)
    return phase, maps, N, swap, new_coords


def ifft(daft, spacing_tol=1e-3, dim=None, real_dim=None, shift=True, true_phase=True, true_amplitude=True,
         chunks_to_segments=False, prefix="freq_", lag=None, real=None):
    """Inverse discrete Fourier transform (reference: xrft/xrft.py:479-646; same arguments).  The input phase factor,
    the coordinate sort / ifftshift, the conjugate-FFT-conjugate inverse, the 1/N and 1/prod(dk) factors and (for
    ``real_dim``) the Hermitian extension of the half spectrum are fused into one device plan."""
    src = daft
    daft = from_any(daft)
    if dim is None:
        dim = list(daft.dims)
    elif isinstance(dim, str):
        dim = [dim]
    else:
        dim = list(dim)
    if real is not None:
        real_dim = real
        warnings.warn(_real_flag_warning, FutureWarning)
    if real_dim is not None:
        if real_dim not in daft.dims:
            raise ValueError("The dimension along which real IFT is taken must be one of the existing dimensions.")
        dim = _move_to_end(dim, real_dim)
    for d in dim:
        daft.get_axis_num(d)
    if not np.all([_is_valid_fft_coord(daft[d].values) for d in dim]):
        raise ValueError("All transformed dimensions coordinates must be numerical or datetime.")
    if lag is None:  # xrft.py:557-560
        lag = [daft[d].attrs.get("direct_lag", 0.0) for d in dim]
        warnings.warn("Default ifft's behaviour (lag=None) changed! Default value of lag was zero (centered output "
                      "coordinates) and is now set to transformed coordinate's attribute: 'direct_lag'.", FutureWarning)
    else:
        if isinstance(lag, float) or isinstance(lag, int):
            lag = [lag]
        if len(dim) != len(lag):
            raise ValueError("dim and lag must have the same length.")
        if not true_phase:
            warnings.warn("Setting lag with true_phase=False does not guarantee accurate ifft.", Warning)
        lag = [daft[d].attrs.get("direct_lag") if l is None else l for d, l in zip(dim, lag)]
    if len(dim) == 0:
        raise NotImplementedError("xrft_amd needs at least one transform dimension")
    if len(dim) > 2:
        # ifftn / irfftn over more than two axes (xrft.py:612-621) is separable: complex stages over the leading dims
        # first, the stage holding the real dimension (whose c2r step must come last) at the end
        if chunks_to_segments:
            raise NotImplementedError("chunks_to_segments is implemented for one or two transform dimensions.")
        cur = daft
        lag_of = dict(zip(dim, lag))
        rest = list(dim)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            while rest:
                last = len(rest) <= 2
                grp, rest = (rest, []) if last else (rest[:2], rest[2:])
                cur = ifft(cur, spacing_tol=spacing_tol, dim=grp, real_dim=real_dim if (last and real_dim in grp) else None,
                           shift=shift, true_phase=true_phase, true_amplitude=true_amplitude, prefix=prefix,
                           lag=[lag_of[d] for d in grp])
        return to_like(from_any(cur), src)
    if len(dim) == 2 and real_dim is None and not chunks_to_segments and sorted(daft.get_axis_num(d) for d in dim) != [len(daft.dims) - 2, len(daft.dims) - 1]:
        # two inverse transform axes that are not the trailing pair: ifftn is separable (xrft.py:612-621) and each axis has a plan that runs where the axis lies
        # (XRFTHIP_AXIS_Y | XRFTHIP_INVERSE; the contiguous axis: the row kernels) -- one axis at a time, no transposed copy of the spectrum or of the result
        lag_of = dict(zip(dim, lag))
        cur = daft
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for d in sorted(dim, key=daft.get_axis_num):
                cur = from_any(ifft(cur, spacing_tol=spacing_tol, dim=[d], shift=shift, true_phase=true_phase, true_amplitude=true_amplitude, prefix=prefix, lag=[lag_of[d]]))
        return to_like(cur, src)
    if chunks_to_segments:
        phase = {d: (np.exp(1j * 2.0 * np.pi * np.asarray(daft[d].values, dtype=np.float64) * l) if true_phase else None) for d, l in zip(dim, lag)}  # (before the reshape)
        daft = _stack_chunks(daft, dim)
        _ph, maps, N, swap, new_coords = _ifft_host(daft, dim, lag, real_dim, shift, False, spacing_tol, prefix)
    else:
        # everything the host derives from the labels (phase tables of a 2^20-point axis, sorting, spacing, centring, lag coordinates: ~15 ms of numpy per call)
        # is remembered on the labelled array per argument set, as the forward calls do (_analyze)
        mkey, guard = None, None
        try:
            mkey = ("ifft", tuple(dim), tuple(float(l) for l in lag), real_dim, bool(shift), bool(true_phase), float(spacing_tol), prefix)
            hash(mkey)
            guard = _label_guard(daft)
        except (TypeError, ValueError):
            mkey = None
        hit = None
        if mkey is not None and guard is not None:
            memo = daft._memo
            if memo is not None and memo[0] == guard:
                hit = memo[1].get(mkey)
        if hit is None:
            hit = _ifft_host(daft, dim, lag, real_dim, shift, true_phase, spacing_tol, prefix)
            if mkey is not None and guard is not None:
                memo = daft._memo
                if memo is None or memo[0] != guard or len(memo[1]) > 16:
                    memo = daft._memo = (guard, {})
                memo[1][mkey] = hit
        phase, maps, N, swap, new_coords = hit
    rawdims = daft.dims
    # device layout: x = real dim if given, else the later axis
    if len(dim) == 1:
        ydim, xdim = None, dim[0]
    elif real_dim is not None:
        ydim, xdim = dim[0], dim[1]
    else:
        a0, a1 = daft.get_axis_num(dim[0]), daft.get_axis_num(dim[1])
        ydim, xdim = (dim[0], dim[1]) if a0 < a1 else (dim[1], dim[0])
    t = _to_device(daft.data)
    if not t.is_complex():
        t = t.to(torch.complex64 if t.dtype == torch.float32 else torch.complex128)
    if len(dim) == 1 and real_dim is None and daft.get_axis_num(xdim) != len(daft.dims) - 1 and isinstance(maps[xdim], str) and maps[xdim] in ("none", "ishift"):
        # one inverse transform along a first / middle axis: where the axis lies (XRFTHIP_AXIS_Y), no transposed copies -- as the forward call
        out = _ifft_axis_y(t, daft.get_axis_num(xdim), maps[xdim], phase[xdim], true_phase, shift, true_amplitude, new_coords[swap[xdim]])
        if out is not None:
            coords = {kname: v for kname, v in daft.coords.items() if kname not in dim}
            cv = new_coords[swap[xdim]]
            coords[swap[xdim]] = Coordinate(cv.dims, cv.values + lag[0], cv.attrs, swap[xdim])
            return to_like(DataArray(out, [swap.get(d, d) for d in rawdims], coords, None, None), src)
    tdims = ([ydim] if ydim is not None else []) + [xdim]
    other = [d for d in daft.dims if d not in tdims]
    order_dims = other + tdims
    if tuple(order_dims) != tuple(daft.dims):
        t = t.permute([daft.get_axis_num(d) for d in order_dims])
    flags = _lib.INVERSE
    ph = {"y": None, "x": None}
    for d in dim:
        ax = "x" if d == xdim else "y"
        m = maps[d]
        p = phase[d]
        if isinstance(m, np.ndarray):  # gather to the unshifted layout on the device; the phase follows the data
            axis = t.dim() - 1 if ax == "x" else t.dim() - 2
            t = engine.gather_axis(t, axis, index=m)
            p = None if p is None else p[m]
        elif m == "ishift":
            flags |= _lib.ISHIFT_X if ax == "x" else _lib.ISHIFT_Y
        elif m == "flip_ishift":
            flags |= (_lib.ISHIFT_X | _lib.FLIP_X) if ax == "x" else (_lib.ISHIFT_Y | _lib.FLIP_Y)
        if p is not None:
            ph[ax] = p
            flags |= _lib.PHASE_IN
        # output: "if not true_phase: ifftshift" then "if shift: fftshift" (xrft.py:617-621) -- for even N the two cancel;
        # the engine's output shift is fftshift, so the remaining cases are: only fftshift, or only ifftshift (odd N differs)
    t = t.contiguous()
    out_shift = {}
    post_roll = []
    for d, n in zip(dim, N if real_dim is None else N[:-1] + [2 * (N[-1] - 1)]):
        ax = "x" if d == xdim else "y"
        if true_phase:
            if shift:
                flags |= _lib.SHIFT_X if ax == "x" else _lib.SHIFT_Y
        else:
            if shift and n % 2 == 0:
                pass  # ifftshift followed by fftshift is the identity for even n
            elif shift:
                pass  # odd n: roll(-(n//2)) then roll(+n//2) is also the identity
            else:
                post_roll.append((ax, -(n // 2)))  # only the ifftshift remains
    nx_in = daft.sizes[xdim]
    nx = 2 * (nx_in - 1) if real_dim is not None else nx_in
    ny = daft.sizes[ydim] if ydim is not None else 1
    if real_dim is not None:
        flags |= _lib.C2R_X
    batch = t.numel() // max(ny * nx_in, 1)
    nprod = float(nx) * float(ny)
    scale = 1.0 / nprod
    if true_amplitude:  # xrft.py:641-642
        scale = scale / math.prod([float(new_coords[swap[d]].attrs["spacing"]) for d in dim])
    try:
        out = None
        if len(dim) == 2 and not (flags & (_lib.FLIP_X | _lib.FLIP_Y)):
            out = _ifft_two_stages(t, batch, ny, nx, flags, float(scale), ph)
        if out is None:
            plan = _get_plan(ndim=len(dim), batch=batch, ny=ny, nx=nx, dtype=t.dtype, out_mode=_lib.OUT_COMPLEX,
                             detrend=_lib.DETREND_NONE, flags=flags, scale=float(scale), window_y=None, window_x=None,
                             phase_y=ph["y"], phase_x=ph["x"])
            out, _ = plan.execute(t)
    except _lib.XrftHipError as e:
        if e.status != _lib.UNSUPPORTED_LENGTH:
            raise
        if len(dim) == 1 and real_dim is not None and not (flags & (_lib.FLIP_X | _lib.ISHIFT_X)):
            # irfft of a long prime length: x[j] = Re sum_k Y[k] e^(+2 pi i jk / n) with Y = the stored half spectrum, its interior
            # samples doubled, zero beyond n/2 -- the zero padding and the factors ride on a table multiply, then the inverse Bluestein
            # pipeline; the real part is a device gather over the (re, im) pairs
            cdt = t.dtype
            two = np.full(nx_in, 2.0)
            two[0] = 1.0
            if nx % 2 == 0:
                two[-1] = 1.0
            fac = two.astype(np.complex128) if ph["x"] is None else two * np.asarray(ph["x"], dtype=np.complex128)
            y = engine.table_mul(t.reshape(-1, nx_in), torch.from_numpy(fac).to(cdt).to(t.device), nx)
            z = _bluestein_1d(y, nx, _lib.OUT_COMPLEX, _lib.DETREND_NONE, flags & ~(_lib.PHASE_IN | _lib.C2R_X), float(scale), None, None)
            out = engine.gather_axis(torch.view_as_real(z.contiguous()), 2, index=np.array([0])).reshape(list(t.shape[:-1]) + [nx])
        elif len(dim) == 2 and not chunks_to_segments:
            # two axes, one of them a length no two-axis plan takes: ifftn is separable (xrft.py:612-621) -- one axis at a time, the
            # real dimension (whose c2r step must come last) at the end; each stage has its own way round (Bluestein through global memory)
            lag_of = dict(zip(dim, lag))
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                cur = ifft(from_any(src), spacing_tol=spacing_tol, dim=[dim[0]], shift=shift, true_phase=true_phase, true_amplitude=true_amplitude,
                           prefix=prefix, lag=[lag_of[dim[0]]])
                cur = ifft(cur, spacing_tol=spacing_tol, dim=[dim[1]], real_dim=real_dim, shift=shift, true_phase=true_phase,
                           true_amplitude=true_amplitude, prefix=prefix, lag=[lag_of[dim[1]]])
            return to_like(from_any(cur), src)
        elif len(dim) != 1 or real_dim is not None:
            lim = 8800 if t.dtype == torch.complex64 else 4400
            raise ValueError(f"transform length(s) {dict(zip(dim, N))} not supported on the device: a length with a prime factor above 128 must "
                             f"be <= ~{lim} samples (Bluestein inside one LDS tile) unless it is the only transform axis") from e
        else:
            out = _bluestein_1d(t, nx, _lib.OUT_COMPLEX, _lib.DETREND_NONE, flags & ~_lib.PHASE_IN, float(scale), None, None, phase_in=ph["x"])
    out = out.reshape([daft.sizes[d] for d in other] + list(out.shape[-len(tdims):]))
    for ax, sh in post_roll:
        out = engine.gather_axis(out, out.dim() - 1 if ax == "x" else out.dim() - 2, roll=sh)
    cur = other + [swap[d] for d in tdims]
    final = [swap.get(d, d) for d in rawdims]
    if cur != final:
        out = out.permute([cur.index(d) for d in final])
    coords = {kname: v for kname, v in daft.coords.items() if kname not in dim}
    for d, l in zip(dim, lag):  # xrft.py:634-639 (keep_attrs: the spacing attribute survives the shift by lag)
        cv = new_coords[swap[d]]
        coords[swap[d]] = Coordinate(cv.dims, cv.values + l, cv.attrs, swap[d])
    return to_like(DataArray(out, final, coords, None, None), src)


def _ifft_two_stages(t, batch, ny, nx, flags, scale, ph):
    """A two-axis inverse transform of (batch, ny, nx) complex data -- or of (batch, ny, nx / 2 + 1) half spectra with XRFTHIP_C2R_X in ``flags`` -- as two one-axis
    passes: ifftn / irfftn are separable (xrft.py:612-621), y where it lies (XRFTHIP_AXIS_Y) on the stored columns, then x along the rows (the c2r step last).
    Only when BOTH stages run well on one-pass kernels (csrc/fastg.h) and the two-axis plan would not: the generic two-axis passes take 30-38 GFFT/s, the stages 100
    each.  Returns None otherwise (the caller builds the two-axis plan)."""
    c2r = bool(flags & _lib.C2R_X)
    nxs = nx // 2 + 1 if c2r else nx  # stored columns
    if t.dtype == torch.complex64 and ny in (256, 512, 1024, 2048, 4096) and (nx in (512, 1024, 2048, 4096) if c2r else nx in (256, 512, 1024, 2048, 4096)):
        return None  # (the two-pass pipeline on complex slabs, csrc/fasty_c2c.h: one plan, a tiled intermediate written in whole lines)
    if ny * nx > (1 << 31) - 1 or batch * ny * nx == 0:
        return None
    fy = _lib.AXIS_Y | _lib.INVERSE | (flags & (_lib.ISHIFT_Y | _lib.SHIFT_Y)) | (_lib.PHASE_IN if ph["y"] is not None else 0)
    fx = _lib.INVERSE | (flags & (_lib.ISHIFT_X | _lib.SHIFT_X | _lib.C2R_X)) | (_lib.PHASE_IN if ph["x"] is not None else 0)
    # the decision is taken once per shape / flag / phase-table set and remembered here (not as an attribute hung on a cached plan)
    dkey = (batch, ny, nx, str(t.dtype), int(flags), float(scale), _akey(ph["y"]), _akey(ph["x"]), engine.bluestein_in_float64())
    def stage_plans():
        py_ = _get_plan(ndim=2, batch=batch, ny=ny, nx=nxs, dtype=t.dtype, out_mode=_lib.OUT_COMPLEX, detrend=_lib.DETREND_NONE, flags=fy, scale=1.0 / float(ny),
                        window_y=None, window_x=None, phase_y=ph["y"], phase_x=None)
        px_ = _get_plan(ndim=1, batch=batch * ny, ny=1, nx=nx, dtype=t.dtype, out_mode=_lib.OUT_COMPLEX, detrend=_lib.DETREND_NONE, flags=fx, scale=scale * float(ny),
                        window_y=None, window_x=None, phase_y=None, phase_x=ph["x"])
        return py_, px_

    with _plan_lock:
        dec = _TWO_STAGE.get(dkey)
        if dec is not None:
            _TWO_STAGE.move_to_end(dkey)
    if dec is False:
        return None
    if dec is None:
        try:
            py, px = stage_plans()
            # ... and run well: four complex columns or more per workgroup along y (32-byte row segments at least), two rows or more per workgroup along x (one long
            # row per workgroup runs at half the rate: (16, 4096, 4096) 24 GFFT/s in two such stages against 30 on the two-axis plan).  Decided on the kernel
            # kinds the C ABI reports (xrfthip_plan_kernel_info), not on the text of describe()
            (ky, cy), (kx, rx) = py.kernel_info(), px.kernel_info()
            seg = cy * t.element_size()  # bytes of a row one column workgroup reads
            good = ((ky == _lib.K_FASTG_Y and seg >= 32) or (ky == _lib.K_FASTM_Y and seg >= 16)) and ((kx in (_lib.K_FASTG_ROWS, _lib.K_FASTM_X) and rx >= 2) or kx == _lib.K_FASTR)  # (K_FASTR: complex rows in registers, csrc/fastr.h fastc_kernel)
            if good and ny * nxs <= 20000:  # (a slab this small may run in ONE pass over both axes: only then is the two-axis plan built to ask)
                whole = _get_plan(ndim=2, batch=batch, ny=ny, nx=nx, dtype=t.dtype, out_mode=_lib.OUT_COMPLEX, detrend=_lib.DETREND_NONE, flags=flags, scale=scale,
                                  window_y=None, window_x=None, phase_y=ph["y"], phase_x=ph["x"])
                good = whole.kernel_info()[0] == _lib.K_GENERIC
        except _lib.XrftHipError as e:
            if e.status not in (_lib.UNSUPPORTED_LENGTH, _lib.BAD_ARG):
                raise
            good = False
        dec = bool(good)
        with _plan_lock:
            _TWO_STAGE[dkey] = dec
            while len(_TWO_STAGE) > _TWO_STAGE_SIZE:
                _TWO_STAGE.popitem(last=False)
        if dec is False:
            return None
    py, px = stage_plans()  # (through _plan_cache every call: evicted plans are rebuilt, none is kept alive from here)
    mid, _ = py.execute(t.reshape(batch, ny, nxs))
    out, _ = px.execute(mid.reshape(batch * ny, 1, nxs))
    return out.reshape(list(t.shape[:-1]) + [nx])


def _ifft_axis_y(t, k, imap, phase, true_phase, shift, true_amplitude, new_coord):
    """ifft along axis k < last of the C-contiguous tensor ``t`` where the axis lies: (batch, n, inner) with XRFTHIP_AXIS_Y | XRFTHIP_INVERSE; None when no such
    plan exists (the caller takes the transposing path).  The shifts as in ``ifft``: an fftshifted input is rotated on load, the output is fftshifted with the true
    phase and shift, rolled afterwards in the one remaining case (xrft.py:612-621)."""
    t = t.contiguous()
    shape = list(t.shape)
    n = shape[k]
    inner = int(np.prod(shape[k + 1:], dtype=np.int64))
    batch = int(np.prod(shape[:k], dtype=np.int64))
    if n * inner > (1 << 31) - 1 or inner > (1 << 30) or batch * n * inner == 0:
        return None
    flags = _lib.AXIS_Y | _lib.INVERSE
    if imap == "ishift":
        flags |= _lib.ISHIFT_Y
    if phase is not None:
        flags |= _lib.PHASE_IN
    roll = 0
    if true_phase:
        if shift:
            flags |= _lib.SHIFT_Y
    elif not shift:
        roll = -(n // 2)  # only the ifftshift of the output remains
    scale = 1.0 / float(n)
    if true_amplitude:
        scale = scale / float(new_coord.attrs["spacing"])
    try:
        plan = _get_plan(ndim=2, batch=batch, ny=n, nx=inner, dtype=t.dtype, out_mode=_lib.OUT_COMPLEX, detrend=_lib.DETREND_NONE, flags=flags,
                         scale=float(scale), window_y=None, window_x=None, phase_y=phase, phase_x=None)
    except _lib.XrftHipError as e:
        if e.status in (_lib.UNSUPPORTED_LENGTH, _lib.BAD_ARG):
            return None
        raise
    out, _ = plan.execute(t.reshape(batch, n, inner))
    out = out.reshape(shape)
    if roll:
        out = engine.gather_axis(out, k, roll=roll)
    return out


def idft(daft, dim=None, true_phase=False, true_amplitude=False, **kwargs):
    """Deprecated alias of ``ifft`` with numpy-like defaults (xrft/xrft.py:253-266)."""
    warnings.warn("This function has been renamed and will disappear in the future. Please use `ifft` instead",
                  FutureWarning)
    return ifft(daft, dim=dim, true_phase=true_phase, true_amplitude=true_amplitude, **kwargs)


def detrend(da, dim, detrend_type="constant"):
    """Remove the mean or the least-squares line / plane over ``dim`` (xrft/detrend.py:11-97)."""
    src = da
    da = from_any(da)
    if dim is None:
        dim = list(da.dims)
    elif isinstance(dim, str):
        dim = [dim]
    else:
        dim = list(dim)
    if detrend_type not in ["constant", "linear", None]:
        raise NotImplementedError("%s is not a valid detrending option. Valid options are: 'constant','linear', "
                                  "or None." % detrend_type)
    if detrend_type is None:
        return src
    if detrend_type == "linear" and len(dim) > 3:
        raise NotImplementedError("Only 1D, 2D, and 3D detrending are implemented so far.")
    axes = [da.get_axis_num(d) for d in dim]
    t = _to_device(da.data)
    dim = [d for _, d in sorted(zip(axes, dim))]  # memory order: the fit does not depend on the order of the axes
    other = [d for d in da.dims if d not in dim]
    order = other + dim
    kind = _lib.DETREND_CONSTANT if detrend_type == "constant" else _lib.DETREND_LINEAR
    ax_sorted = sorted(axes)
    if (len(dim) <= 2 and ax_sorted[-1] != len(da.dims) - 1 and ax_sorted == list(range(ax_sorted[0], ax_sorted[0] + len(dim)))
            and int(np.prod(t.shape[ax_sorted[-1] + 1:], dtype=np.int64)) > 1):
        # one or two adjacent axes that are not the trailing ones: detrended where they lie (xrfthip_detrend_inner), no transposed copies
        # -- unless the extents exceed what the library carries in 32 bits, or its scratch would outgrow the array (None: transposing path)
        out = engine.detrend_inner(t.contiguous(), ax_sorted[0], len(dim), kind)
        if out is not None:
            return to_like(DataArray(out, da.dims, da.coords, da.name, da.attrs), src)
    if tuple(order) != tuple(da.dims):
        t = t.permute([da.get_axis_num(d) for d in order])
    t = t.contiguous()
    if len(dim) > 3 or (len(dim) == 3 and detrend_type == "constant"):
        # the mean over any number of trailing axes is the mean of the flattened block
        nblk = int(np.prod([da.sizes[d] for d in dim]))
        out = engine.detrend(t.reshape(-1, nblk), 1, kind).reshape(t.shape)
    else:
        out = engine.detrend(t, len(dim), kind)
    if tuple(order) != tuple(da.dims):
        out = out.permute([order.index(d) for d in da.dims])
    return to_like(DataArray(out, da.dims, da.coords, da.name, da.attrs), src)


def _window_correction_factor(c, scaling, window):
    """xrft.py:649-660: mean(w^2) (density) or mean(w)^2 (spectrum) of the outer-product window."""
    if window is None:
        raise ValueError("window_correction can only be applied when windowing is turned on.")
    vecs = [_window_vector(window, n) for n in c.corr_N]
    w = vecs[0]
    for v in vecs[1:]:
        w = np.multiply.outer(w, v)
    if scaling == "density":
        return (w ** 2).mean()
    elif scaling == "spectrum":
        return w.mean() ** 2
    raise ValueError("Unknown {} scaling flag".format(scaling))


def _psd_scaling_factor(c, scaling):
    """xrft.py:663-670."""
    fs = math.prod([float(c.new_coords[c.swap[d]].attrs["spacing"]) for d in c.dim])
    if scaling == "density":
        return fs
    elif scaling == "spectrum":
        return fs ** 2
    raise ValueError("Unknown {} scaling flag".format(scaling))


def _spectrum_scale(c, amp, scaling, window_correction, window):
    """Everything that multiplies |F|^2 (or F1 conj F2): (prod dx)^2, / window factor, x prod(dk)^(1|2)."""
    scale = float(amp)
    if scaling != "false_density":  # xrft.py:745-748
        if window_correction:
            scale = scale / _window_correction_factor(c, scaling, window)
        scale = scale * _psd_scaling_factor(c, scaling)
    return scale


def _spectrum(da, da2, dim, real_dim, scaling, window_correction, true_phase, kwargs, iso=None):
    if "real" in kwargs:  # xrft.py:728-730 (`real` stays in kwargs and reaches fft as well)
        real_dim = kwargs.get("real")
        warnings.warn(_real_flag_warning, FutureWarning)
    if "density" in kwargs:  # xrft.py:718-726
        density = kwargs.pop("density")
        warnings.warn("density flag will be deprecated in future version of xrft and replaced by scaling flag. "
                      'density=True should be replaced by scaling="density" and density=False will not be '
                      "maintained.\nscaling flag is ignored !", FutureWarning)
        scaling = "density" if density else "false_density"
    kw = dict(spacing_tol=1e-3, shift=True, detrend=None, window=None, chunks_to_segments=False, prefix="freq_",
              real=None)
    for k in list(kwargs):
        if k in ("true_amplitude", "true_phase"):
            kwargs.pop(k)  # overridden by the reference (xrft.py:732-734, 814)
    unknown = set(kwargs) - set(kw)
    if unknown:
        raise TypeError(f"fft() got an unexpected keyword argument {sorted(unknown)[0]!r}")
    kw.update(kwargs)
    c = _analyze(da, kw["spacing_tol"], dim, real_dim, kw["shift"], kw["detrend"], kw["window"], true_phase,
                 kw["chunks_to_segments"], kw["prefix"], kw["real"])
    c2 = None
    amp = math.prod(c.delta_x) ** 2
    if da2 is not None:
        c2 = _analyze(da2, kw["spacing_tol"], dim, real_dim, kw["shift"], kw["detrend"], kw["window"], true_phase,
                      kw["chunks_to_segments"], kw["prefix"], kw["real"])
        if [c.swap.get(d, d) for d in c.rawdims] != [c2.swap.get(d, d) for d in c2.rawdims]:  # xrft.py:819-820
            raise ValueError("The two datasets have different dimensions")
        if c.N != c2.N or not np.allclose(c.delta_x, c2.delta_x, rtol=1e-12):
            raise ValueError("The two datasets have different frequency coordinates (size or spacing)")
        amp = math.prod(c.delta_x) * math.prod(c2.delta_x)
    scale = _spectrum_scale(c, amp, scaling, window_correction, kw["window"])
    flags = _lib.REALDIM_X2 if c.real_dim is not None else 0  # xrft.py:742-743
    mode = _lib.OUT_POWER if da2 is None else _lib.OUT_CROSS
    return c, c2, mode, scale, flags


def power_spectrum(da, dim=None, real_dim=None, scaling="density", window_correction=False, **kwargs):
    """Power spectrum |F(da')|^2 with density / spectrum scaling (xrft/xrft.py:685-750)."""
    src = da
    da = from_any(da)
    nd = _nd_dims(da, dim, real_dim, kwargs.get("real"))
    if nd is not None:
        return to_like(_spectrum_nd(da, None, nd, real_dim, scaling, window_correction, False, dict(kwargs)), src)
    c, _, mode, scale, flags = _spectrum(da, None, dim, real_dim, scaling, window_correction, False, dict(kwargs))
    try:
        out, _, other = _execute(c, c.da, mode, scale, extra_flags=flags)
    except _UnsupportedLength:
        daw, wide = _wide(da)
        return to_like(_narrow(_spectrum_nd(daw, None, _two_dims(da, dim, real_dim, kwargs.get("real")), real_dim, scaling, window_correction, False,
                                            dict(kwargs), one_at_a_time=True), wide), src)
    return to_like(_label_output(c, c.da, out, other), src)


def cross_spectrum(da1, da2, dim=None, real_dim=None, scaling="density", window_correction=False, true_phase=True,
                   **kwargs):
    """Cross spectrum F(da1') conj(F(da2')) (xrft/xrft.py:753-835)."""
    src = da1
    da1, da2 = from_any(da1), from_any(da2)
    nd = _nd_dims(da1, dim, real_dim, kwargs.get("real"))
    if nd is not None:
        return to_like(_spectrum_nd(da1, da2, nd, real_dim, scaling, window_correction, true_phase, dict(kwargs)), src)
    c, c2, mode, scale, flags = _spectrum(da1, da2, dim, real_dim, scaling, window_correction, true_phase, dict(kwargs))
    try:
        return to_like(_cross_result(c, c2, mode, scale, flags), src)
    except _UnsupportedLength:
        d1w, wide1 = _wide(da1)
        d2w, wide2 = _wide(da2)  # (narrowed back only when BOTH were widened: float32 x float64 stays the promoted complex128, whatever the order)
        return to_like(_narrow(_spectrum_nd(d1w, d2w, _two_dims(da1, dim, real_dim, kwargs.get("real")), real_dim, scaling, window_correction, true_phase,
                                            dict(kwargs), one_at_a_time=True), wide1 and wide2), src)


def _cross_result(c, c2, mode, scale, flags):
    out, _, other = _execute(c, c.da, mode, scale, da2=c2.da, c2=c2, extra_flags=flags)
    extra = None
    if c.true_phase:  # the product keeps daft1's coordinates, including their direct_lag attribute (xrft.py:469, 825)
        extra = {c.swap[d]: {"direct_lag": lag} for d, lag in zip(c.dim, c.lag_x)}
    return _label_output(c, c.da, out, other, extra)


def cross_phase(da1, da2, dim=None, true_phase=True, **kwargs):
    """Cross phase arg(F(da1') conj(F(da2'))) in [-pi, pi] (xrft/xrft.py:838-874); the angle is taken in the
    epilogue of the cross-spectrum kernel, the complex cross spectrum is never written."""
    src = da1
    da1, da2 = from_any(da1), from_any(da2)
    kw = dict(kwargs)
    real_dim = kw.pop("real_dim", None)
    scaling = kw.pop("scaling", "density")
    window_correction = kw.pop("window_correction", False)
    c, c2, mode, scale, flags = _spectrum(da1, da2, dim, real_dim, scaling, window_correction, true_phase, kw)
    try:
        cp = _cross_result(c, c2, _lib.OUT_PHASE, abs(scale) if scale != 0 else 1.0, flags & ~_lib.REALDIM_X2)
    except _UnsupportedLength:
        # a length no fused plan takes (numpy.fft takes any): the angle of the cross spectrum, which has its own way round
        # (one axis at a time, Bluestein through global memory) -- what the reference does literally (xrft.py:871-874)
        cs = from_any(cross_spectrum(da1, da2, dim=dim, real_dim=real_dim, scaling=scaling, window_correction=window_correction,
                                     true_phase=true_phase, **kw))
        cp = DataArray(engine.angle(_to_device(cs.data)), cs.dims, cs.coords, None, None)
    if da1.name and da2.name:
        cp.name = "{}_{}_phase".format(da1.name, da2.name)
    return to_like(cp, src)


# ------------------------------------------------------------------------------------------------------
# isotropic spectra (xrft.py:877-1187)
# ------------------------------------------------------------------------------------------------------
_bins_cache: "OrderedDict[tuple, tuple]" = OrderedDict()


def _radial_bins(k, l, nfactor, ref_order=None):
    """Bin codes and per-bin mean radius for the grid sqrt(k^2 + l^2), dims (k, l)  (xrft.py:975-981, 910-923).

    ``pd.cut`` on the float64 radii, exactly the reference's expression; the per-bin mean replaces
    ``numpy_groupies.aggregate(func="mean", fill_value=0)``.  The result depends only on the two frequency vectors
    and nfactor, and costs seconds of host time at 4096^2, so it is cached (the reference recomputes it per call).
    """
    key = (k.size, l.size, _digest(k), _digest(l), nfactor, ref_order)
    with _plan_lock:
        hit = _bins_cache.get(key)
        if hit is not None:
            _bins_cache.move_to_end(key)
            return hit
    res = _radial_bins_uncached(k, l, nfactor, ref_order) + (key,)
    with _plan_lock:
        _bins_cache[key] = res
        while len(_bins_cache) > 8:
            _bins_cache.popitem(last=False)
    return res


def _radial_bins_uncached(k, l, nfactor, ref_order=None):
    N = [k.size, l.size]
    nbins = int(min(N) / nfactor)
    freq_r = np.sqrt(k[:, None] ** 2 + l[None, :] ** 2)
    binned = pd.cut(np.ravel(freq_r), nbins)
    codes = binned.codes.reshape(freq_r.shape)
    nb = binned.categories.size
    cr, fr = codes, freq_r
    if ref_order is not None:
        # The bin-centre coordinate is a per-bin MEAN: its last bits depend on the order of the sum.  The reference sums
        # over the (fftdim[1], fftdim[0]) grid of the SHIFTED coordinates (xrft.py:980-981); visit the same cells in the
        # same order (the codes and radii themselves are order-independent).
        shift_k, shift_l, transpose = ref_order
        ik = np.fft.fftshift(np.arange(k.size)) if shift_k else np.arange(k.size)
        il = np.fft.fftshift(np.arange(l.size)) if shift_l else np.arange(l.size)
        cr, fr = codes[np.ix_(ik, il)], freq_r[np.ix_(ik, il)]
        if transpose:
            cr, fr = cr.T, fr.T
    valid = cr >= 0
    cnt = np.bincount(cr[valid], minlength=nb)
    s_ = np.bincount(cr[valid], weights=fr[valid], minlength=nb)
    kr = np.where(cnt > 0, s_ / np.maximum(cnt, 1), 0.0)
    return codes.astype(np.int32), kr, nb


def _finish_iso(kr, k, l, truncate, iso_vals):
    """truncate / dropna semantics of xrft.py:983-991, 1007-1010 (dropna only looks at data values)."""
    if truncate:
        kmax = l.max() if k.max() > l.max() else k.max()
        kr = np.where(kr <= kmax, kr, np.nan)
        keep = ~np.isnan(iso_vals).reshape(-1, iso_vals.shape[-1]).any(axis=0)
        if not keep.all():
            iso_vals, kr = iso_vals[..., keep], kr[keep]
    else:
        warnings.warn("Isotropic wavenumber larger than the Nyquist wavenumber may result.", FutureWarning)
    return kr, iso_vals


def isotropize(ps, fftdim, nfactor=4, truncate=True, complx=False):
    """Azimuthal (radial-bin) SUM of an existing 2-D spectrum (xrft/xrft.py:948-1010)."""
    src = ps
    ps = from_any(ps)
    k = np.asarray(ps[fftdim[1]].values, dtype=np.float64)
    l = np.asarray(ps[fftdim[0]].values, dtype=np.float64)
    codes, kr, nb, _ = _radial_bins(k, l, nfactor)  # dims (fftdim[1], fftdim[0])
    other = [d for d in ps.dims if d not in fftdim]
    order = other + [fftdim[1], fftdim[0]]
    t = _to_device(ps.data)
    if tuple(order) != tuple(ps.dims):
        t = t.permute([ps.get_axis_num(d) for d in order])
    t = t.contiguous()
    bm = torch.from_numpy(np.ascontiguousarray(codes)).to(t.device)
    iso = engine.isotropize(t, bm, nb)
    iso = iso.reshape([ps.sizes[d] for d in other] + [nb])
    vals = iso.cpu().numpy() if truncate else iso  # (truncate: dropna inspects the data values, xrft.py:1007-1008)
    kr, vals = _finish_iso(kr, k, l, truncate, vals)
    coords = {c: v for c, v in ps.coords.items() if not (set(v.dims) & set(fftdim))}
    coords["freq_r"] = Coordinate(("freq_r",), kr, None, "freq_r")
    return to_like(DataArray(vals, other + ["freq_r"], coords, ps.name, ps.attrs), src)


def _iso_spectrum(da, da2, spacing_tol, dim, shift, detrend_, scaling, window, window_correction, nfactor, truncate,
                  kwargs):
    if "density" in kwargs:  # xrft.py:1072-1074
        density = kwargs.pop("density")
        scaling = "density" if density else "false_density"
    if dim is None:
        dim = da.dims
        if da2 is not None and tuple(dim) != tuple(da2.dims):
            raise ValueError("The two datasets have different dimensions")
    if len(dim) != 2:
        raise ValueError("The Fourier transform should be two dimensional")
    dim = list(dim)
    kw = dict(kwargs, spacing_tol=spacing_tol, shift=shift, detrend=detrend_, window=window)
    true_phase = kw.pop("true_phase", True) if da2 is not None else False
    real_dim = kw.pop("real_dim", None)
    c, c2, mode, scale, flags = _spectrum(da, da2, dim, real_dim, scaling, window_correction, true_phase, kw)
    fftdim = ["freq_" + d for d in dim]  # xrft.py:1093 (hard-coded prefix, as in the reference)
    for f in fftdim:
        if f not in c.new_coords:
            raise KeyError(f)
    # bin map on the engine's (ky, kx) grid in UNSHIFTED index order; radii are order-independent
    ky = c.k_unshifted[c.dim.index(c.ydim)]
    kx = c.k_unshifted[c.dim.index(c.xdim)]
    kk = c.new_coords[fftdim[1]].values
    ll = c.new_coords[fftdim[0]].values
    # reference bins over (fftdim[1], fftdim[0]) of the shifted coordinates: the edges depend only on min/max of the same set
    # of radii; kr (a per-bin mean of the same multiset) is summed in the reference's cell order to match it bit for bit
    codes_yx, kr, nb, bkey = _radial_bins(ky, kx, nfactor, (bool(c.shift), bool(c.shift), fftdim[1] == c.swap[c.xdim]))
    iso_cfg = {"binmap": codes_yx, "nbins": nb, "binmap_key": bkey}
    da_in = da
    da = c.da
    try:
        out, iso, other = _execute(c, da, mode, scale, da2=None if c2 is None else c2.da, c2=c2, iso=iso_cfg,
                                   extra_flags=flags | _lib.ISO | _lib.NO_SPECTRUM_OUT)
    except _UnsupportedLength:
        # no two-axis plan for these lengths: what the reference does literally (xrft.py:1085-1095, 1177-1187) -- the full spectrum
        # (its axes transformed one at a time), then isotropize
        skw = dict(dim=dim, real_dim=real_dim, scaling=scaling, window_correction=window_correction, **kw)
        if da2 is None:
            full = power_spectrum(da_in, **skw)
        else:
            full = cross_spectrum(da_in, da2, true_phase=true_phase, **skw)
        return from_any(isotropize(full, fftdim, nfactor=nfactor, truncate=truncate, complx=da2 is not None))
    vals = iso.reshape([da.sizes[d] for d in other] + [nb])
    if truncate:  # dropna inspects the DATA values (xrft.py:1007-1008): the only case that needs them on the host
        vals = vals.cpu().numpy()
    kr, vals = _finish_iso(kr, kk, ll, truncate, vals)
    coords = {cn: cv for cn, cv in da.coords.items() if not (set(cv.dims) & set(dim))}
    coords["freq_r"] = Coordinate(("freq_r",), kr, None, "freq_r")
    return DataArray(vals, other + ["freq_r"], coords, None, None)


def isotropic_power_spectrum(da, spacing_tol=1e-3, dim=None, shift=True, detrend=None, scaling="density",
                             window=None, window_correction=False, nfactor=4, truncate=False, **kwargs):
    """Isotropic (radially binned) power spectrum (xrft/xrft.py:1013-1095); spectrum and bin-sum are fused on
    the device, the full 2-D spectrum is never written."""
    src = da
    da = from_any(da)
    return to_like(_iso_spectrum(da, None, spacing_tol, dim, shift, detrend, scaling, window, window_correction,
                                 nfactor, truncate, dict(kwargs)), src)


def isotropic_cross_spectrum(da1, da2, spacing_tol=1e-3, dim=None, shift=True, detrend=None, scaling="density",
                             window=None, window_correction=False, nfactor=4, truncate=False, **kwargs):
    """Isotropic cross spectrum (xrft/xrft.py:1098-1187)."""
    src = da1
    da1, da2 = from_any(da1), from_any(da2)
    return to_like(_iso_spectrum(da1, da2, spacing_tol, dim, shift, detrend, scaling, window, window_correction,
                                 nfactor, truncate, dict(kwargs)), src)


def fit_loglog(x, y):
    """Least-squares line in log2-log2 space (xrft/xrft.py:1190-1214)."""
    p = np.polyfit(np.log2(x), np.log2(y), 1)
    y_fit = 2 ** (np.log2(x) * p[0] + p[1])
    return y_fit, p[0], p[1]
