"""pad / unpad of evenly spaced grids (reference: xrft/padding.py; same arguments, same coordinate attributes).

Padding is a memory operation around the spectral hot path: the data stay where they are -- a
device tensor is padded on the device with copies / gathers (``constant``, ``edge``, ``wrap``, ``reflect``,
``symmetric``); the statistical modes and ``linear_ramp`` and the ``odd`` reflections go through ``numpy.pad`` on the
host.  Coordinates are extended with the coordinate's own spacing (padding.py:277-323) and carry ``pad_width``.
"""
from __future__ import annotations

import numpy as np

from .labeled import Coordinate, DataArray, from_any, to_like

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

__all__ = ["pad", "unpad", "get_spacing"]

_GATHER_MODES = ("edge", "wrap", "reflect", "symmetric")


def _diff_coord(values):
    """xrft.py:195-212 (numeric and datetime64 coordinates; cftime is handled by api._diff_coord)."""
    from .api import _diff_coord as impl

    return impl(values)


def get_spacing(coord):
    """Spacing of an evenly spaced coordinate (xrft/utils.py:8-19)."""
    values = coord.values if hasattr(coord, "values") else np.asarray(coord)
    name = getattr(coord, "name", None)
    diff = _diff_coord(values)
    if not np.allclose(diff, diff[0]):
        raise ValueError(f"Found unevenly spaced coordinates '{name}'. These coordinates should be evenly spaced.")
    return diff[0]


def _either_dict_or_kwargs(pos, kw, func_name):
    if pos is None or pos == {}:
        return dict(kw)
    if not isinstance(pos, dict):
        raise ValueError(f"the first argument to .{func_name} must be a dictionary")
    if kw:
        raise ValueError(f"cannot specify both keyword and positional arguments to .{func_name}")
    return dict(pos)


def _pair(v):
    if isinstance(v, (int, np.integer)):
        return int(v), int(v)
    a, b = v
    return int(a), int(b)


def _check_bad_coords(da, padding_coordinates):
    """padding.py:196-226."""
    bad = []
    for coord in padding_coordinates:
        dim = da[coord].dims[0]
        bad += [c for c, cv in da.coords.items() if dim in cv.dims and c != coord]
    if bad:
        raise ValueError("Please, drop the following coordinates from the passed DataArray before trying to pad it: "
                         + "'" + "', '".join(bad) + "'" + ".")


def _pad_coordinate(values, pw, spacing):
    """padding.py:277-323: linear extrapolation with the coordinate's spacing on both sides."""
    n_start, n_end = pw
    values = np.asarray(values)
    out = np.pad(values, (n_start, n_end))
    vmin, vmax = values[0], values[-1]
    out[:n_start] = vmin - n_start * spacing + np.linspace(0, spacing * (n_start - 1), n_start)
    out[len(out) - n_end:] = vmax + spacing + np.linspace(0, spacing * (n_end - 1), n_end)
    return out


def _per_dim(arg, dims, default):
    """xarray's {dim: (before, after)} | ((before, after),) | (v,) | scalar forms -> {dim: (before, after)}."""
    if arg is None:
        return {d: default for d in dims}
    if isinstance(arg, dict):
        return {d: (tuple(np.broadcast_to(arg[d], (2,)).tolist()) if d in arg else default) for d in dims}
    a = np.asarray(arg)
    if a.ndim == 0:
        return {d: (a.item(), a.item()) for d in dims}
    a = np.broadcast_to(a, (len(dims), 2)) if a.shape != (len(dims), 2) else a
    return {d: (a[i][0].item(), a[i][1].item()) for i, d in enumerate(dims)}


def pad(da, pad_width=None, mode="constant", stat_length=None, constant_values=0, end_values=None, reflect_type=None,
        **pad_width_kwargs):
    """Pad an array and extrapolate its evenly spaced coordinates (xrft/padding.py:11-193)."""
    src = da
    da = from_any(da)
    pad_width = _either_dict_or_kwargs(pad_width, pad_width_kwargs, "pad")
    _check_bad_coords(da, pad_width.keys())
    for d in pad_width:
        da.get_axis_num(d)
    pw = {d: _pair(v) for d, v in pad_width.items()}
    dims_axis_order = [d for d in da.dims if d in pw]
    data = da.data
    on_device = torch is not None and isinstance(data, torch.Tensor)
    simple = mode == "constant" or (mode in _GATHER_MODES and reflect_type in (None, "even"))
    if simple:
        cv = _per_dim(constant_values, dims_axis_order, (0, 0)) if mode == "constant" else None
        for d in dims_axis_order:  # numpy.pad works axis by axis in axis order: later axes own the corners
            ax = da.get_axis_num(d)
            b, a = pw[d]
            n = data.shape[ax]
            if mode == "constant":
                shp_b, shp_a = list(data.shape), list(data.shape)
                shp_b[ax], shp_a[ax] = b, a
                if on_device:
                    parts = [torch.full(shp_b, cv[d][0], dtype=data.dtype, device=data.device), data,
                             torch.full(shp_a, cv[d][1], dtype=data.dtype, device=data.device)]
                    data = torch.cat(parts, dim=ax)
                else:
                    data = np.concatenate([np.full(shp_b, cv[d][0], dtype=data.dtype), data,
                                           np.full(shp_a, cv[d][1], dtype=data.dtype)], axis=ax)
            else:
                idx = np.pad(np.arange(n), (b, a), mode=mode)
                if on_device:
                    data = torch.index_select(data, ax, torch.from_numpy(idx).to(data.device))
                else:
                    data = np.take(data, idx, axis=ax)
    else:  # statistics, linear_ramp, odd reflections: numpy.pad on the host
        host = da.values
        kw = {}
        if mode in ("maximum", "mean", "median", "minimum") and stat_length is not None:
            sl = _per_dim(stat_length, dims_axis_order, None)
        else:
            sl = None
        full_pw, full_sl, full_ev = [], [], []
        ev = _per_dim(end_values, dims_axis_order, (0, 0)) if mode == "linear_ramp" else None
        for d in da.dims:
            full_pw.append(pw.get(d, (0, 0)))
            if sl is not None:
                full_sl.append(sl.get(d) or (host.shape[da.get_axis_num(d)],) * 2)
            if ev is not None:
                full_ev.append(ev.get(d, (0, 0)))
        if sl is not None:
            kw["stat_length"] = full_sl
        if ev is not None:
            kw["end_values"] = full_ev
        if mode in ("reflect", "symmetric") and reflect_type is not None:
            kw["reflect_type"] = reflect_type
        padded = np.pad(host, full_pw, mode=mode, **kw)
        data = torch.from_numpy(padded).to(data.device) if on_device else padded
    coords = {}
    for name, cv_ in da.coords.items():
        if name in pw:
            spacing = get_spacing(cv_)
            attrs = dict(cv_.attrs)
            attrs["pad_width"] = pad_width[name]
            coords[name] = Coordinate(cv_.dims, _pad_coordinate(cv_.values, pw[name], spacing), attrs, name)
        else:
            coords[name] = cv_
    return to_like(DataArray(data, da.dims, coords, da.name, da.attrs), src)


def _pad_width_to_slice(pad_width, size):
    """padding.py:425-446."""
    if isinstance(pad_width, (int, np.integer)):
        pad_width = (pad_width, pad_width)
    return slice(int(pad_width[0]), int(size - pad_width[1]))


def unpad(da, pad_width=None, **pad_width_kwargs):
    """Undo ``pad`` by slicing the array and its coordinates (xrft/padding.py:326-422)."""
    src = da
    da = from_any(da)
    if pad_width is None and not pad_width_kwargs:
        pad_width = {dim: c.attrs["pad_width"] for dim, c in da.coords.items() if "pad_width" in c.attrs}
        if not pad_width:
            raise ValueError("The passed array doesn't seem to be a padded one: the 'pad_width' attribute was missing "
                             "on every one of its coordinates. ")
    else:
        pad_width = _either_dict_or_kwargs(pad_width, pad_width_kwargs, "pad")
    slices = {dim: _pad_width_to_slice(pad_width[dim], da[dim].size) for dim in pad_width}
    out = da.isel(**slices)
    for dim in pad_width:
        if dim in out.coords and "pad_width" in out.coords[dim].attrs:
            c = out.coords[dim]
            out.coords[dim] = Coordinate(c.dims, c.values, {k: v for k, v in c.attrs.items() if k != "pad_width"}, dim)
    return to_like(out, src)
