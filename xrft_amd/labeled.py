"""A minimal labelled array: the subset of ``xarray.DataArray`` the xrft API surface needs.

xarray is not installed in the build image nor on the GPU box, so the label logic the reference gets from
xarray (dimension names, 1-D coordinate vectors with attrs, kept / dropped coordinates) lives in this small
self-contained class.  When xarray *is* installed, ``xrft_amd`` functions also accept ``xarray.DataArray``
and hand the same type back (see ``from_any`` / ``to_like``).

``.data`` is whatever holds the samples (numpy array or torch tensor, possibly on the GPU);
``.values`` always returns a host numpy array.
"""
from __future__ import annotations

import itertools

import numpy as np

try:  # torch is the device-memory carrier; the container itself also works with plain numpy
    import torch
except Exception:  # pragma: no cover
    torch = None


_coord_tokens = itertools.count(1)


class Coordinate:
    """A named coordinate variable: ``dims`` (tuple of dim names), ``values`` (numpy), ``attrs``."""

    def __init__(self, dims, values, attrs=None, name=None):
        self.dims = (dims,) if isinstance(dims, str) else tuple(dims)
        # The values are an owned, read-only array (xarray keeps its dimension coordinates in immutable indexes, too): what the API
        # derives from a coordinate vector (spacing check, lag, digest) can then be remembered by the array's identity instead of
        # being re-derived from its bytes on every call.  An array that already is an owned read-only one is shared, not copied.
        # NOTE for callers: ``coord.values`` is read-only (assign a new array to ``coord.values`` to change a coordinate; in-place edits raise).
        self.values = values
        self.attrs = dict(attrs or {})
        self.name = name

    @property
    def values(self):
        return self._values

    @values.setter
    def values(self, values):
        a = values if isinstance(values, np.ndarray) else np.asarray(values)
        if a.flags.writeable or not a.flags.owndata or a.base is not None:
            a = np.array(a, copy=True)
            a.setflags(write=False)
        self._values = a
        # a token that is never reused (``id()`` of a freed object is): what the API remembers about this coordinate is keyed by it
        self._token = next(_coord_tokens)

    def _clone(self, name=None):
        """Another Coordinate object over the same (immutable) values, with the same token -- what they stand for is the same -- and its own attrs."""
        c = Coordinate.__new__(Coordinate)
        c.dims = self.dims
        c._values = self._values
        c._token = self._token
        c.attrs = dict(self.attrs)
        c.name = self.name if name is None else name
        return c

    # numpy interop so that ``npt.assert_allclose(ft["freq_x"], expected)`` works as with xarray
    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.values, dtype=dtype)

    def __len__(self):
        return len(self.values)

    def __getitem__(self, i):
        return self.values[i]

    def __getattr__(self, item):  # ``ft["freq_x"].spacing`` like xarray attribute access
        attrs = self.__dict__.get("attrs", {})
        if item in attrs:
            return attrs[item]
        raise AttributeError(item)

    @property
    def data(self):
        return self.values

    @property
    def size(self):
        return self.values.size

    def __repr__(self):
        return f"Coordinate({self.name!r}, dims={self.dims}, n={self.values.shape}, attrs={self.attrs})"


def _is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


class DataArray:
    def __init__(self, data, dims=None, coords=None, name=None, attrs=None):
        if not _is_torch(data):
            data = np.asarray(data)
        self.data = data
        nd = data.ndim
        if dims is None:
            dims = tuple(f"dim_{i}" for i in range(nd))
        self.dims = (dims,) if isinstance(dims, str) else tuple(dims)
        if len(self.dims) != nd:
            raise ValueError(f"dims {self.dims} do not match data of rank {nd}")
        self.name = name
        self.attrs = dict(attrs or {})
        self._chunks = None  # {dim: tuple of chunk lengths}: metadata only (see .chunk)
        self._memo = None    # what the API derived from this array's labels on earlier calls (xrft_amd/api.py:_analyze), with its guard
        self.coords = {}
        if coords is not None:
            if isinstance(coords, dict):
                items = coords.items()
            else:  # xarray's positional form: coords=[x, y] aligned with dims
                items = zip(self.dims, coords)
            for k, v in items:
                self.coords[k] = self._as_coord(k, v)
        for k, c in self.coords.items():
            for d, n in zip(c.dims, c.values.shape):
                if d not in self.dims:
                    raise ValueError(f"coordinate {k} has dimension {d} which is not a dimension of the array")
                if self.shape[self.dims.index(d)] != n:
                    raise ValueError(f"coordinate {k} length {n} conflicts with dimension {d}")

    @classmethod
    def _trusted(cls, data, dims, coords):
        """A result the library itself has just labelled: ``coords`` are Coordinate objects under their own names whose lengths match ``data`` by construction,
        non-transform coordinates are the input's own (immutable) objects -- no copies, no checks (xrft_amd/api.py:_label_output; ~10 us of a 40-us call)."""
        self = cls.__new__(cls)
        self.data = data
        self.dims = tuple(dims)
        self.name = None
        self.attrs = {}
        self._chunks = None
        self._memo = None
        self.coords = coords
        return self

    @staticmethod
    def _as_coord(name, v):
        if isinstance(v, Coordinate):
            return Coordinate(v.dims, v.values, v.attrs, name)
        if isinstance(v, tuple) and len(v) in (2, 3) and (
                isinstance(v[0], str) or (isinstance(v[0], (tuple, list)) and all(isinstance(x, str) for x in v[0]))):
            return Coordinate(v[0], np.asarray(v[1]), v[2] if len(v) == 3 else None, name)
        a = np.asarray(v)
        if a.ndim == 0:
            return Coordinate((), a, None, name)
        return Coordinate((name,), a, None, name)

    # ---------------------------------------------------------------- basic protocol
    @property
    def shape(self):
        return tuple(self.data.shape)

    @property
    def ndim(self):
        return len(self.dims)

    @property
    def sizes(self):
        return dict(zip(self.dims, self.shape))

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def values(self):
        d = self.data
        if _is_torch(d):
            d = d.detach()
            if d.is_conj():
                d = d.resolve_conj()
            return d.cpu().numpy()
        return d

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.values, dtype=dtype)

    def __len__(self):
        return self.shape[0]

    def get_axis_num(self, dim):
        if isinstance(dim, (list, tuple)):
            return [self.get_axis_num(d) for d in dim]
        if dim not in self.dims:
            raise ValueError(f"{dim!r} not found in array dimensions {self.dims}")
        return self.dims.index(dim)

    def __getitem__(self, key):
        """``da["x"]``: the coordinate of a dimension (``arange`` if it has none), or a named coordinate."""
        if isinstance(key, str):
            if key in self.coords:
                return self.coords[key]
            if key in self.dims:
                return Coordinate((key,), np.arange(self.sizes[key]), None, key)
            raise KeyError(key)
        raise TypeError("only coordinate lookup by name is supported; use .isel for positional selection")

    def __contains__(self, key):
        return key in self.coords or key in self.dims

    def __getattr__(self, item):  # ``da.x`` / ``ps.freq_x`` shorthand
        d = self.__dict__
        if "coords" in d and (item in d["coords"] or item in d.get("dims", ())):
            return self[item]
        raise AttributeError(item)

    # ---------------------------------------------------------------- the few transformations the API needs
    def _new(self, data, dims=None, coords=None, name=None):
        out = DataArray(data, self.dims if dims is None else dims, self.coords if coords is None else coords,
                        self.name if name is None else name, self.attrs)
        if self._chunks and (dims is None or set(out.dims) == set(self.dims)):
            out._chunks = dict(self._chunks)
        return out

    def transpose(self, *dims):
        if not dims:
            dims = self.dims[::-1]
        perm = [self.get_axis_num(d) for d in dims]
        data = self.data.permute(perm) if _is_torch(self.data) else self.data.transpose(perm)
        return self._new(data, dims)

    def isel(self, **idx):
        data = self.data
        dims = list(self.dims)
        coords = dict(self.coords)
        for d, i in idx.items():
            ax = dims.index(d)
            sl = [slice(None)] * len(dims)
            sl[ax] = i
            data = data[tuple(sl)]
            scalar = not isinstance(i, slice)
            new = {}
            for k, c in coords.items():
                if d in c.dims:
                    cax = c.dims.index(d)
                    csl = [slice(None)] * len(c.dims)
                    csl[cax] = i
                    v = c.values[tuple(csl)]
                    cd = tuple(x for x in c.dims if x != d) if scalar else c.dims
                    new[k] = Coordinate(cd, v, c.attrs, k)
                else:
                    new[k] = c
            coords = new
            if scalar:
                dims.pop(ax)
        return DataArray(data, dims, coords, self.name, self.attrs)

    def _reduce(self, dim, mean):
        dims = list(self.dims) if dim is None else ([dim] if isinstance(dim, str) else list(dim))
        keep = [d for d in self.dims if d not in dims]
        coords = {k: c for k, c in self.coords.items() if not (set(c.dims) & set(dims))}
        data = self.data
        if (_is_torch(data) and data.device.type != "cpu" and (data.is_floating_point() or data.is_complex())
                and all(self.sizes[d] > 0 for d in dims)):
            # floating-point device data stay on the device: one library kernel per reduced dim (xrfthip_reduce_axis: float64 accumulation in
            # index order, bit-reproducible) -- e.g. the batch mean of isotropic spectra, test_xrft.py:1011-1013
            from . import engine

            cur = list(self.dims)
            for d in dims:
                ax = cur.index(d)
                data = engine.reduce_axis(data, ax, 1.0 / data.shape[ax] if mean else 1.0)
                cur.pop(ax)
            return DataArray(data, keep, coords, self.name, self.attrs)
        v = self.values
        ax = tuple(self.get_axis_num(d) for d in dims)
        return DataArray(v.mean(axis=ax) if mean else v.sum(axis=ax), keep, coords, self.name, self.attrs)

    def mean(self, dim=None):
        """Mean over ``dim`` (name or list of names; all dims if None), as xarray's ``.mean(dim)``."""
        return self._reduce(dim, True)

    def sum(self, dim=None):
        return self._reduce(dim, False)

    # ---------------------------------------------------------------- chunk metadata (stands in for dask chunking)
    @property
    def chunks(self):
        """None, or per-axis tuples of chunk lengths like a dask-backed xarray.DataArray."""
        if not self._chunks:
            return None
        return tuple(self._chunks.get(d, (n,)) for d, n in zip(self.dims, self.shape))

    def chunk(self, chunks=None):
        """``da.chunk({dim: n})`` records equal chunks of length n along dim.  xrft uses dask chunks only to define
        Bartlett/Welch segments (``chunks_to_segments``) and to refuse transforms across chunk boundaries; the data
        stay where they are."""
        out = self._new(self.data)
        ch = dict(self._chunks or {})
        for d, n in (chunks or {}).items():
            N = self.sizes[d]
            n = N if n is None else int(n)
            ch[d] = tuple([n] * (N // n) + ([N % n] if N % n else []))
        out._chunks = ch or {d: (n,) for d, n in zip(self.dims, self.shape)}
        return out

    def copy(self):
        data = self.data.clone() if _is_torch(self.data) else self.data.copy()
        return self._new(data, coords={k: Coordinate(c.dims, c.values.copy(), c.attrs, k) for k, c in self.coords.items()})

    def __repr__(self):
        kind = "torch:" + str(self.data.device) if _is_torch(self.data) else "numpy"
        return f"<xrft_amd.DataArray {self.name or ''} {self.sizes} dtype={self.dtype} [{kind}] coords={list(self.coords)}>"

    # ---------------------------------------------------------------- xarray interop (only if xarray is importable)
    def to_xarray(self):
        import xarray as xr

        coords = {k: (c.dims, c.values, c.attrs) for k, c in self.coords.items()}
        return xr.DataArray(self.values, dims=self.dims, coords=coords, name=self.name, attrs=self.attrs)

    @classmethod
    def from_xarray(cls, xda):
        coords = {k: (tuple(v.dims), np.asarray(v.values), dict(v.attrs)) for k, v in xda.coords.items()}
        out = cls(np.asarray(xda.values), tuple(xda.dims), coords, xda.name, dict(xda.attrs))
        # a dask-chunked source keeps its chunk layout as metadata: chunks_to_segments (xrft.py:106-136) and the refusal to
        # transform across chunk boundaries (xrft.py:279) depend on it, the data themselves sit in one buffer
        if getattr(xda, "chunks", None) is not None:
            out._chunks = {d: tuple(int(n) for n in ch) for d, ch in zip(xda.dims, xda.chunks)}
        return out


def is_xarray(obj):
    t = type(obj)
    return t.__module__.split(".")[0] == "xarray" and t.__name__ == "DataArray"


def from_any(obj):
    """Accept xrft_amd.DataArray or (if installed) xarray.DataArray."""
    if isinstance(obj, DataArray):
        return obj
    if is_xarray(obj):
        return DataArray.from_xarray(obj)
    raise TypeError(f"expected a DataArray, got {type(obj).__name__}")


def to_like(result, template):
    """Return ``result`` as the array type the caller passed in."""
    if is_xarray(template):
        return result.to_xarray()
    return result
