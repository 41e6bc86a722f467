"""The array-backend module of the reference's only seam: ``_fft_module(da)`` (xrft/xrft.py:32-36) returns ``numpy.fft`` or
``dask.array.fft`` and the reference calls ``fftn / rfftn / ifftn / irfftn / fftshift / ifftshift`` on it with
``(array, axes=[...])`` (xrft.py:398-404, 439-447, 612-621).  This module has exactly those six names with numpy's semantics,
computing on the MI355X through the C ABI (one plan per call, or one per group of axes), so a third backend slots in there:

    def _fft_module(da):
        if isinstance(da.data, torch.Tensor) and da.data.is_cuda:
            import xrft_amd.fftmod as fft_module          # <- this module
        elif da.chunks: ...

Inputs: torch tensors on the device (returned as such) or anything ``numpy.asarray`` takes (uploaded; a device tensor
comes back -- ``.cpu().numpy()`` it if needed).  numpy promotion rules: float32 -> complex64, float64 -> complex128.
Axes that lie last in memory use the fused one- / two-axis plans; a single middle or first axis is transformed where it lies
(XRFTHIP_AXIS_Y); anything else is composed of those (the transform is separable), never through torch.fft.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, engine
from .api import _bluestein_1d, _get_plan, _to_device

__all__ = ["fftn", "ifftn", "rfftn", "irfftn", "fftshift", "ifftshift"]


def _axes(a, axes):
    if axes is None:
        axes = list(range(a.dim()))
    elif np.isscalar(axes):
        axes = [int(axes)]
    axes = [int(ax) % a.dim() for ax in axes]
    if len(set(axes)) != len(axes):
        raise ValueError("axes must be unique")
    return axes


def _as_complex(t):
    return t if t.is_complex() else t.to(torch.complex64 if t.dtype == torch.float32 else torch.complex128)


def _c2c_one(t, ax, inverse):
    """Complex transform of one axis of a contiguous tensor, in place in memory order (no transposed copy)."""
    shape = list(t.shape)
    n = shape[ax]
    scale = 1.0 / n if inverse else 1.0
    flags = _lib.INVERSE if inverse else 0
    if ax == t.dim() - 1:
        batch = t.numel() // max(n, 1)
        try:
            plan = _get_plan(ndim=1, batch=batch, ny=1, nx=n, dtype=t.dtype, out_mode=_lib.OUT_COMPLEX, detrend=_lib.DETREND_NONE,
                             flags=flags, scale=scale, window_y=None, window_x=None, phase_y=None, phase_x=None)
        except _lib.XrftHipError as e:
            if e.status != _lib.UNSUPPORTED_LENGTH:
                raise
            # numpy.fft takes any length: a prime factor too large for one LDS tile goes through Bluestein in global memory
            return _bluestein_1d(t, n, _lib.OUT_COMPLEX, _lib.DETREND_NONE, flags, scale, None, None).reshape(shape)
        out, _ = plan.execute(t.reshape(batch, 1, n))
        return out.reshape(shape)
    batch = int(np.prod(shape[:ax], dtype=np.int64))
    inner = int(np.prod(shape[ax + 1:], dtype=np.int64))
    plan = None
    if n * inner <= (1 << 31) - 1 and inner <= (1 << 30):  # (the engine indexes one [n][inner] slab with 32 bits)
        try:
            plan = _get_plan(ndim=2, batch=batch, ny=n, nx=inner, dtype=t.dtype, out_mode=_lib.OUT_COMPLEX, detrend=_lib.DETREND_NONE,
                             flags=flags | _lib.AXIS_Y, scale=scale, window_y=None, window_x=None, phase_y=None, phase_x=None)
        except _lib.XrftHipError as e:
            if e.status != _lib.UNSUPPORTED_LENGTH:
                raise
    if plan is None:
        # a column too long for one LDS tile, or a slab beyond 2^31 elements: transposed copy, 1-D plan (four-step inside), copy back
        tt = t.movedim(ax, -1).contiguous()
        return _c2c_one(tt, tt.dim() - 1, inverse).movedim(-1, ax).contiguous()
    out, _ = plan.execute(t.reshape(batch, n, inner))
    return out.reshape(shape)


def _c2c_last2(t, inverse):
    ny, nx = t.shape[-2], t.shape[-1]
    batch = t.numel() // max(ny * nx, 1)
    try:
        return _c2c_last2_plan(t, inverse, ny, nx, batch)
    except _lib.XrftHipError as e:
        if e.status != _lib.UNSUPPORTED_LENGTH:
            raise
        return _c2c_one(_c2c_one(t, t.dim() - 1, inverse), t.dim() - 2, inverse)  # separable: one axis at a time


def _c2c_last2_plan(t, inverse, ny, nx, batch):
    plan = _get_plan(ndim=2, batch=batch, ny=ny, nx=nx, dtype=t.dtype, out_mode=_lib.OUT_COMPLEX, detrend=_lib.DETREND_NONE,
                     flags=_lib.INVERSE if inverse else 0, scale=1.0 / (ny * nx) if inverse else 1.0, window_y=None, window_x=None,
                     phase_y=None, phase_x=None)
    out, _ = plan.execute(t.reshape(batch, ny, nx))
    return out.reshape(t.shape)


def _c2c(t, axes, inverse):
    t = _as_complex(t).contiguous()
    axes = sorted(axes)
    nd = t.dim()
    if len(axes) >= 2 and axes[-2:] == [nd - 2, nd - 1]:
        t = _c2c_last2(t, inverse)
        axes = axes[:-2]
    for ax in reversed(axes):
        t = _c2c_one(t, ax, inverse)
    return t


def fftn(a, s=None, axes=None, norm=None):
    """numpy.fft.fftn(a, axes=axes) on the device (xrft.py:444); real input is transformed as real (half the work) when the
    last listed axis is the last in memory."""
    if s is not None or norm not in (None, "backward"):
        raise NotImplementedError("fftn: s / norm are not used by xrft")
    t = _to_device(a)
    axes = _axes(t, axes)
    if not t.is_complex() and len(axes) <= 2 and sorted(axes) == list(range(t.dim() - len(axes), t.dim())):
        # real input over the trailing axes: the fused plan computes the half spectrum and mirrors it in its last pass
        t = t.contiguous()
        nx = t.shape[-1]
        ny = t.shape[-2] if len(axes) == 2 else 1
        batch = t.numel() // max(ny * nx, 1)
        try:
            plan = _get_plan(ndim=len(axes), batch=batch, ny=ny, nx=nx, dtype=t.dtype, out_mode=_lib.OUT_COMPLEX,
                             detrend=_lib.DETREND_NONE, flags=0, scale=1.0, window_y=None, window_x=None, phase_y=None, phase_x=None)
        except _lib.XrftHipError as e:
            if e.status != _lib.UNSUPPORTED_LENGTH:
                raise
            return _c2c(t, axes, False)  # a length no fused plan takes: one axis at a time, Bluestein through global memory
        out, _ = plan.execute(t.reshape(batch, ny, nx))
        return out.reshape(t.shape)
    return _c2c(t, axes, False)


def ifftn(a, s=None, axes=None, norm=None):
    """numpy.fft.ifftn(a, axes=axes) (xrft.py:614): conj(FFT(conj z)) / prod(N), in the engine's inverse plans."""
    if s is not None or norm not in (None, "backward"):
        raise NotImplementedError("ifftn: s / norm are not used by xrft")
    t = _to_device(a)
    return _c2c(t, _axes(t, axes), True)


def rfftn(a, s=None, axes=None, norm=None):
    """numpy.fft.rfftn(a, axes=axes) (xrft.py:400): real transform over the LAST listed axis (n//2 + 1 samples kept), complex
    transforms over the others."""
    if s is not None or norm not in (None, "backward"):
        raise NotImplementedError("rfftn: s / norm are not used by xrft")
    t = _to_device(a)
    if t.is_complex():
        raise TypeError("rfftn needs real input")
    axes = _axes(t, axes)
    last = axes[-1]
    moved = last != t.dim() - 1
    if moved:  # the half-spectrum axis must be the contiguous one
        t = t.movedim(last, -1)
        axes = [ax - 1 if ax > last else ax for ax in axes[:-1]] + [t.dim() - 1]
    t = t.contiguous()
    nx = t.shape[-1]
    rest = axes[:-1]
    plan2 = None
    if rest and sorted(rest)[-1] == t.dim() - 2:  # fused two-axis half-spectrum plan over the trailing pair
        ny = t.shape[-2]
        batch = t.numel() // max(ny * nx, 1)
        try:
            plan2 = _get_plan(ndim=2, batch=batch, ny=ny, nx=nx, dtype=t.dtype, out_mode=_lib.OUT_COMPLEX, detrend=_lib.DETREND_NONE,
                              flags=_lib.HALF_X, scale=1.0, window_y=None, window_x=None, phase_y=None, phase_x=None)
        except _lib.XrftHipError as e:
            if e.status != _lib.UNSUPPORTED_LENGTH:
                raise
    if plan2 is not None:
        out, _ = plan2.execute(t.reshape(batch, ny, nx))
        out = out.reshape(list(t.shape[:-1]) + [nx // 2 + 1])
        rest = [ax for ax in rest if ax != t.dim() - 2]
    else:
        batch = t.numel() // max(nx, 1)
        try:
            plan = _get_plan(ndim=1, batch=batch, ny=1, nx=nx, dtype=t.dtype, out_mode=_lib.OUT_COMPLEX, detrend=_lib.DETREND_NONE,
                             flags=_lib.HALF_X, scale=1.0, window_y=None, window_x=None, phase_y=None, phase_x=None)
            out, _ = plan.execute(t.reshape(batch, 1, nx))
        except _lib.XrftHipError as e:
            if e.status != _lib.UNSUPPORTED_LENGTH:
                raise
            out = _bluestein_1d(t, nx, _lib.OUT_COMPLEX, _lib.DETREND_NONE, _lib.HALF_X, 1.0, None, None)  # (any length, as numpy.fft)
        out = out.reshape(list(t.shape[:-1]) + [nx // 2 + 1])
    for ax in sorted(rest, reverse=True):
        out = _c2c_one(out, ax, False)
    if moved:
        out = out.movedim(-1, last).contiguous()
    return out


def irfftn(a, s=None, axes=None, norm=None):
    """numpy.fft.irfftn(a, axes=axes) with the default output length 2 (m - 1) along the last listed axis (xrft.py:612):
    inverse complex transforms over the other axes, then the Hermitian (c2r) plan."""
    if s is not None or norm not in (None, "backward"):
        raise NotImplementedError("irfftn: s / norm are not used by xrft")
    t = _as_complex(_to_device(a))
    axes = _axes(t, axes)
    last = axes[-1]
    moved = last != t.dim() - 1
    if moved:
        t = t.movedim(last, -1)
        axes = [ax - 1 if ax > last else ax for ax in axes[:-1]] + [t.dim() - 1]
    t = t.contiguous()
    for ax in sorted(axes[:-1], reverse=True):
        t = _c2c_one(t, ax, True)
    m = t.shape[-1]
    nx = 2 * (m - 1)
    batch = t.numel() // max(m, 1)
    plan = _get_plan(ndim=1, batch=batch, ny=1, nx=nx, dtype=t.dtype, out_mode=_lib.OUT_COMPLEX, detrend=_lib.DETREND_NONE,
                     flags=_lib.INVERSE | _lib.C2R_X, scale=1.0 / nx, window_y=None, window_x=None, phase_y=None, phase_x=None)
    out, _ = plan.execute(t.reshape(batch, 1, m))
    out = out.reshape(list(t.shape[:-1]) + [nx])
    if moved:
        out = out.movedim(-1, last).contiguous()
    return out


def _roll(a, axes, sign):
    t = _to_device(a)
    for ax in _axes(t, axes):
        n = t.shape[ax]
        t = engine.gather_axis(t, ax, roll=sign * (n // 2) if sign > 0 else -(n // 2))
    return t


def fftshift(x, axes=None):
    """numpy.fft.fftshift: roll by n // 2 along ``axes`` (xrft.py:446-447) -- a device copy kernel, one pass per axis."""
    return _roll(x, axes, +1)


def ifftshift(x, axes=None):
    """numpy.fft.ifftshift: roll by -(n // 2) along ``axes`` (xrft.py:440, 617)."""
    return _roll(x, axes, -1)
