"""Thin object wrapper over the C ABI: one ``SpectralPlan`` = one ``xrfthip_plan`` + its workspace.

torch is used only as the owner of device memory and the provider of the current HIP stream; every
arithmetic step runs inside libxrft_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np
import torch

from . import _lib

_DTYPES = {torch.float32: _lib.F32, torch.float64: _lib.F64, torch.complex64: _lib.C64, torch.complex128: _lib.C128}
_REAL_OF = {torch.float32: torch.float32, torch.float64: torch.float64, torch.complex64: torch.float32,
            torch.complex128: torch.float64}
_CPLX_OF = {torch.float32: torch.complex64, torch.float64: torch.complex128, torch.complex64: torch.complex64,
            torch.complex128: torch.complex128}


def _stream_handle(t):
    if t.is_cuda:
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return C.c_void_p(0)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class SpectralPlan:
    """detrend -> window -> (flip, ifftshift) -> FFT over the last ``ndim`` axes -> shift/phase/scale ->
    complex | power | cross (-> radial bin-sum), in one call on the current stream."""

    def __init__(self, ndim, batch, ny, nx, dtype, out_mode=_lib.OUT_COMPLEX, detrend=_lib.DETREND_NONE, flags=0,
                 scale=1.0, window_y=None, window_x=None, phase_y=None, phase_x=None, binmap=None, nbins=0,
                 slabs_per_group=0, inner=1, mid=1):
        self._dll = _lib.load()
        self._h = C.c_void_p(0)
        if dtype not in _DTYPES:
            raise TypeError(f"unsupported dtype {dtype}")
        self.ndim, self.batch, self.ny, self.nx = int(ndim), int(batch), int(ny), int(nx)
        self.dtype, self.out_mode, self.flags = dtype, int(out_mode), int(flags)
        half_y = (bool(flags & _lib.HALF_X) and bool(flags & _lib.AXIS_Y)) or bool(flags & _lib.HALF_Y)  # (ABI 0.1.4: real_dim along the ONE transformed axis of an AXIS_Y plan; 0.1.6: HALF_Y of the inner / mid layouts)
        self.nx_out = self.nx // 2 + 1 if ((flags & _lib.HALF_X) and not half_y) else self.nx
        self.ny_out = self.ny // 2 + 1 if half_y else self.ny
        self.nbins = int(nbins)
        self.inner = max(int(inner), 1)  # > 1: (batch, ny, nx, inner) arrays, the transform axes are not the trailing ones
        self.mid = max(int(mid), 1)      # > 1: (batch, ny, mid, nx, inner): independent elements between the two transform axes
        d = _lib.Desc(C.sizeof(_lib.Desc), self.ndim, self.batch, self.ny, self.nx, _DTYPES[dtype], self.out_mode,
                      int(detrend), self.flags, float(scale), int(slabs_per_group), 0, self.inner, self.mid)
        _lib.check(self._dll.xrfthip_plan_create(C.byref(self._h), C.byref(d)))
        for axis, w in ((0, window_y), (1, window_x)):
            if w is not None:
                w = np.ascontiguousarray(w, dtype=np.float64)
                _lib.check(self._dll.xrfthip_plan_set_window(self._h, axis, w.ctypes.data_as(C.c_void_p), w.size))
        for axis, p in ((0, phase_y), (1, phase_x)):
            if p is not None:
                p = np.ascontiguousarray(p, dtype=np.complex128)
                _lib.check(self._dll.xrfthip_plan_set_phase(self._h, axis, p.ctypes.data_as(C.c_void_p), p.size))
        if flags & _lib.ISO:
            bm = np.ascontiguousarray(binmap, dtype=np.int32)
            if bm.shape != (self.ny, self.nx_out):
                raise ValueError(f"bin map shape {bm.shape} != {(self.ny, self.nx_out)}")
            _lib.check(self._dll.xrfthip_plan_set_binmap(self._h, bm.ctypes.data_as(C.c_void_p), self.ny,
                                                         self.nx_out, self.nbins))

    def __del__(self):
        try:
            if self._h:
                self._dll.xrfthip_plan_destroy(self._h)
                self._h = C.c_void_p(0)
        except Exception:
            pass

    @property
    def workspace_bytes(self):
        return int(self._dll.xrfthip_workspace_bytes(self._h))

    def describe(self):
        buf = C.create_string_buffer(8192)
        self._dll.xrfthip_plan_describe(self._h, buf, len(buf))
        return buf.value.decode()

    def kernel_info(self):
        """(kernel kind, independent sequences per workgroup) of the plan: ``_lib.K_*`` -- what ``describe()`` prints, as numbers."""
        k, n = C.c_int32(0), C.c_int32(0)
        _lib.check(self._dll.xrfthip_plan_kernel_info(self._h, C.byref(k), C.byref(n)))
        return int(k.value), int(n.value)

    def uses_bluestein(self):
        """True when an axis of the plan runs Bluestein's algorithm inside the tile kernels (a prime factor with no butterfly)."""
        return bool(self._dll.xrfthip_plan_uses_bluestein(self._h))

    def set_profiling(self, enable=True):
        _lib.check(self._dll.xrfthip_plan_set_profiling(self._h, int(bool(enable))))

    def read_profile(self):
        """{label: (launches, total_ms)} from HIP events recorded around every launch since set_profiling(True)."""
        buf = C.create_string_buffer(1 << 16)
        n = self._dll.xrfthip_plan_profile_read(self._h, buf, len(buf))
        if n < 0:
            _lib.check(n)
        out = {}
        for line in buf.value.decode().splitlines():
            label, cnt, ms = line.rsplit(" ", 2)
            out[label] = (int(cnt), float(ms))
        return out

    def out_dtype(self):
        real = self.out_mode in (_lib.OUT_POWER, _lib.OUT_PHASE) or (self.flags & _lib.C2R_X)
        return _REAL_OF[self.dtype] if real else _CPLX_OF[self.dtype]

    def execute(self, in0, in1=None, out=None, iso=None):
        """``in0``/``in1``: contiguous tensors of shape (batch, ny, nx) (any leading shape that flattens to it).
        Returns (out, iso); either may be None depending on the flags."""
        dev = in0.device
        nx_in = self.nx // 2 + 1 if (self.flags & _lib.C2R_X) else self.nx
        if in0.dtype != self.dtype or not in0.is_contiguous() or in0.numel() != self.batch * self.ny * nx_in * self.inner * self.mid:
            raise ValueError("in0 does not match the plan (dtype / contiguity / size)")
        if self.out_mode in (_lib.OUT_CROSS, _lib.OUT_PHASE):
            if in1 is None or in1.dtype != self.dtype or not in1.is_contiguous() or in1.numel() != in0.numel():
                raise ValueError("in1 does not match the plan")
        want_out = not (self.flags & _lib.NO_SPECTRUM_OUT)
        if want_out and out is None:
            shape = (self.batch, self.ny_out, self.nx_out) + ((self.inner,) if self.inner > 1 else ())
            if self.mid > 1:
                shape = (self.batch, self.ny_out, self.mid, self.nx_out, self.inner)
            out = torch.empty(shape, dtype=self.out_dtype(), device=dev)
        elif want_out:  # a caller's buffer (graph capture, composed passes): held to the plan before the device sees its pointer
            need = self.batch * self.ny_out * self.nx_out * self.inner * self.mid
            if out.dtype != self.out_dtype() or not out.is_contiguous() or out.numel() != need or out.device != dev:
                raise ValueError(f"out does not match the plan (dtype {self.out_dtype()}, contiguous, {need} elements on {dev})")
        iso_dtype = torch.complex128 if self.out_mode == _lib.OUT_CROSS else torch.float64
        if self.flags & _lib.ISO and iso is None:
            iso = torch.empty((self.batch, self.nbins), device=dev, dtype=iso_dtype)
        elif self.flags & _lib.ISO:
            if iso.dtype != iso_dtype or not iso.is_contiguous() or iso.numel() != self.batch * self.nbins or iso.device != dev:
                raise ValueError(f"iso does not match the plan (dtype {iso_dtype}, contiguous, {self.batch * self.nbins} elements on {dev})")
        if self.batch == 0:  # nothing to transform: empty outputs, no device call
            return (out if want_out else None), iso
        stream = _stream_handle(in0)
        # The C plan is immutable after creation, so nothing plan-wide is locked: threads on different streams enqueue
        # concurrently.  Callers that share a stream share its scratch buffer: their enqueues (a short sequence of kernel
        # launches each) must not interleave, hence one lock per (device, stream).
        with _stream_lock(dev, stream):
            # grown (and the outgrown buffer retired) only HERE, under the stream's lock: no other thread is between its look-up and
            # its enqueue on this stream, so the event recorded at retirement really follows every use of the old buffer
            ws = _workspace(dev, stream, self.workspace_bytes)
            _lib.check(self._dll.xrfthip_exec(self._h, _ptr(in0), _ptr(in1), _ptr(out if want_out else None), _ptr(iso),
                                              _ptr(ws), ws.numel(), stream))
        return (out if want_out else None), iso


# One grow-only scratch buffer per (device, stream), shared by every plan: work on one stream is ordered, so plans never
# overlap in it; two streams never share scratch memory.  (A workspace per plan pinned several GB per cached plan.)
_WS = {}
_WS_LOCKS = {}
_WS_RETIRED = []  # (buffer, event) of outgrown buffers: kernels enqueued earlier may still be using them until the event has passed
_WS_LOCK = threading.Lock()


def _stream_lock(dev, stream):
    key = (str(dev), stream.value)
    with _WS_LOCK:
        lock = _WS_LOCKS.get(key)
        if lock is None:
            lock = _WS_LOCKS[key] = threading.Lock()
        return lock


def _workspace(dev, stream, nbytes):
    """The stream's scratch buffer, grown if needed.  Call with _stream_lock(dev, stream) held."""
    key = (str(dev), stream.value)
    with _WS_LOCK:
        # outgrown buffers go once the work that may still use them has finished (an event recorded on their stream when they retired)
        _WS_RETIRED[:] = [(b, ev) for b, ev in _WS_RETIRED if ev is not None and not ev.query()]
        ws = _WS.get(key)
        if ws is None or ws.numel() < nbytes:
            want = max(int(nbytes), 256)
            if ws is not None:
                want = max(want, int(1.5 * ws.numel()))  # geometric growth: a session whose plans grow step by step re-allocates O(log) times
                ev = None
                if dev.type == "cuda":
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(dev))
                    _WS_RETIRED.append((ws, ev))
            ws = _WS[key] = torch.empty(want, dtype=torch.uint8, device=dev)
        return ws


def clear_workspaces():
    """Release the shared scratch buffers (synchronises the devices first)."""
    with _WS_LOCK:
        if _WS or _WS_RETIRED:
            if _lib.device() == "cuda" and torch.cuda.is_available():
                torch.cuda.synchronize()
            _WS.clear()
            _WS_RETIRED.clear()


def detrend(x, ndim, kind):
    """Stand-alone detrend over the last ``ndim`` (1|2|3) axes of a contiguous tensor (xrft/detrend.py:11-138)."""
    dll = _lib.load()
    if x.dtype not in _DTYPES or not x.is_contiguous():
        raise ValueError("detrend needs a contiguous float/complex tensor")
    if ndim == 3:
        n0, n1, n2 = x.shape[-3:]
        batch = x.numel() // max(n0 * n1 * n2, 1)
        out = torch.empty_like(x)
        nws = int(dll.xrfthip_detrend_workspace_bytes(batch))
        ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
        _lib.check(dll.xrfthip_detrend3(_DTYPES[x.dtype], batch, n0, n1, n2, kind, _ptr(x), _ptr(out), _ptr(ws), nws,
                                        _stream_handle(x)))
        return out
    nx = x.shape[-1]
    ny = x.shape[-2] if ndim == 2 else 1
    batch = x.numel() // max(ny * nx, 1)
    out = torch.empty_like(x)
    nws = int(dll.xrfthip_detrend_workspace_bytes(batch))
    ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
    _lib.check(dll.xrfthip_detrend(_DTYPES[x.dtype], ndim, batch, ny, nx, kind, _ptr(x), _ptr(out), _ptr(ws), nws,
                                   _stream_handle(x)))
    return out


def detrend_inner(x, axis0, naxes, kind):
    """Stand-alone detrend over ``naxes`` (1|2) ADJACENT axes starting at ``axis0`` of a contiguous tensor, whatever follows them
    (xrft.detrend over axes that are not the trailing ones: no transposed copy; xrft/detrend.py:54-55, 64-71, 100-113)."""
    dll = _lib.load()
    if x.dtype not in _DTYPES or not x.is_contiguous():
        raise ValueError("detrend needs a contiguous float/complex tensor")
    batch = int(np.prod(x.shape[:axis0], dtype=np.int64))
    ny = x.shape[axis0] if naxes == 2 else 1
    nx = x.shape[axis0 + naxes - 1]
    inner = int(np.prod(x.shape[axis0 + naxes:], dtype=np.int64))
    if inner > (1 << 30) or nx > (1 << 31) - 1 or ny > (1 << 31) - 1:
        return None  # beyond the extents xrfthip_detrend_inner takes (it would say BAD_ARG): the caller transposes
    nws = int(dll.xrfthip_detrend_inner_workspace_bytes(_DTYPES[x.dtype], batch, inner))
    if nws > max(x.numel() * x.element_size(), 64 << 20):
        return None  # a few samples per element: the partial sums would outgrow the array
    out = torch.empty_like(x)
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=x.device)
    _lib.check(dll.xrfthip_detrend_inner(_DTYPES[x.dtype], naxes, batch, ny, nx, inner, kind, _ptr(x), _ptr(out), _ptr(ws), nws, _stream_handle(x)))
    return out


_BLUESTEIN_F64 = [os.environ.get("XRFT_AMD_BLUESTEIN", "float64").lower() not in ("float32", "f32", "fast")]


def bluestein_in_float64(enable=None):
    """float32 data on a transform length too long for Bluestein's algorithm inside one LDS tile (a prime factor above 128 and more than
    ~8800 samples: api._bluestein_1d, three table products around two long plans through global memory) run in float64 between two
    precision changes -- the default: with float32 intermediates stored between the passes, the composition leaves the bins 1e-3 of
    the spectrum's peak 1.3e-3 of relative error, above the 1e-3 every other path holds.  ``bluestein_in_float64(False)`` (or
    XRFT_AMD_BLUESTEIN=float32 in the environment) keeps float32 arithmetic there: the max-norm bound of 1e-3 still holds.
    Bluestein lengths INSIDE a tile (the ERA5 grid's 721 = 7 x 103 latitudes) stay in float32 either way: measured on the GPU they hold
    1.2e-4 per bin (profiles/r04_bluestein_f32.txt).  Returns the setting."""
    if enable is not None:
        _BLUESTEIN_F64[0] = bool(enable)
    return _BLUESTEIN_F64[0]


_WIDER = {torch.float32: torch.float64, torch.complex64: torch.complex128}
_NARROWER = {v: k for k, v in _WIDER.items()}


def convert(x, dtype, out=None):
    """``x`` in the other precision (float32 <-> float64, complex64 <-> complex128) by the library's own kernel (xrfthip_convert)."""
    dll = _lib.load()
    if _WIDER.get(x.dtype) is not dtype and _NARROWER.get(x.dtype) is not dtype:
        raise TypeError(f"convert: {x.dtype} -> {dtype}")
    x = x.contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=dtype, device=x.device)
    elif out.dtype is not dtype or out.numel() != x.numel() or not out.is_contiguous():
        raise ValueError("convert: out does not match")
    _lib.check(dll.xrfthip_convert(_DTYPES[x.dtype], _DTYPES[dtype], x.numel(), _ptr(x), _ptr(out), _stream_handle(x)))
    return out


def spectrum_tail(a, b, scale):
    """|a|^2 * scale (real) or a * conj(b) * scale (complex) of transformed fields (xrft.py:740, 825)."""
    dll = _lib.load()
    if not a.is_complex() or not a.is_contiguous() or (b is not None and (b.dtype != a.dtype or b.shape != a.shape or not b.is_contiguous())):
        raise ValueError("spectrum_tail needs contiguous complex tensors of one shape and dtype")
    real_dt = torch.float32 if a.dtype == torch.complex64 else torch.float64
    out = torch.empty(a.shape, dtype=a.dtype if b is not None else real_dt, device=a.device)
    _lib.check(dll.xrfthip_spectrum_tail(_DTYPES[a.dtype], a.numel(), _ptr(a), _ptr(b), _ptr(out), float(scale),
                                         _stream_handle(a)))
    return out


def angle(a):
    """arg(a) in [-pi, pi] of a contiguous complex tensor (numpy.angle; xrft.cross_phase on composed cross spectra)."""
    dll = _lib.load()
    if not a.is_complex():
        raise ValueError("angle needs a complex tensor")
    a = a.contiguous()
    out = torch.empty(a.shape, dtype=torch.float32 if a.dtype == torch.complex64 else torch.float64, device=a.device)
    _lib.check(dll.xrfthip_angle(_DTYPES[a.dtype], a.numel(), _ptr(a), _ptr(out), _stream_handle(a)))
    return out


def spectrum_tail_axis(a, b, scale, axis, last_is_one):
    """spectrum_tail with the real-dim factor [1, 2, ..., 2, (1)] along ``axis`` (xrft.py:673-682)."""
    dll = _lib.load()
    if not a.is_complex() or not a.is_contiguous() or (b is not None and (b.dtype != a.dtype or b.shape != a.shape or not b.is_contiguous())):
        raise ValueError("spectrum_tail_axis needs contiguous complex tensors of one shape and dtype")
    real_dt = torch.float32 if a.dtype == torch.complex64 else torch.float64
    out = torch.empty(a.shape, dtype=a.dtype if b is not None else real_dt, device=a.device)
    axis = axis % a.dim()
    outer = int(np.prod(a.shape[:axis], dtype=np.int64))
    inner = int(np.prod(a.shape[axis + 1:], dtype=np.int64))
    _lib.check(dll.xrfthip_spectrum_tail_axis(_DTYPES[a.dtype], outer, a.shape[axis], inner, int(bool(last_is_one)), _ptr(a), _ptr(b),
                                              _ptr(out), float(scale), _stream_handle(a)))
    return out


def gather_axis(x, axis, index=None, roll=0):
    """``x`` re-ordered along ``axis``: out[..., i, ...] = x[..., index[i], ...] (``index``: host int array) or numpy.roll by
    ``roll``.  A device copy kernel (xrfthip_gather_axis): no torch arithmetic."""
    dll = _lib.load()
    if not x.is_contiguous():
        x = x.contiguous()
    axis = axis % x.dim()
    n_in = x.shape[axis]
    outer = int(np.prod(x.shape[:axis], dtype=np.int64))
    inner = int(np.prod(x.shape[axis + 1:], dtype=np.int64))
    idx = None
    n_out = n_in
    if index is not None:
        idx = torch.from_numpy(np.ascontiguousarray(index, dtype=np.int64)).to(x.device)
        n_out = idx.numel()
    shape = list(x.shape)
    shape[axis] = n_out
    out = torch.empty(shape, dtype=x.dtype, device=x.device)
    _lib.check(dll.xrfthip_gather_axis(x.element_size(), outer, n_out, inner, n_in, _ptr(idx), int(roll), _ptr(x), _ptr(out),
                                       _stream_handle(x)))
    return out


def table_mul(x, table, n_out):
    """out[..., j] = (j < n ? x[..., j] : 0) * table[j], j < n_out, complex result: zero padding / truncation with a pointwise factor
    (the pointwise steps of Bluestein's algorithm through global memory)."""
    dll = _lib.load()
    x = x.contiguous()
    n_in = x.shape[-1]
    batch = x.numel() // max(n_in, 1)
    cdt = torch.complex64 if x.dtype in (torch.float32, torch.complex64) else torch.complex128
    if table.dtype != cdt or table.numel() < min(n_in, n_out) or not table.is_contiguous():
        raise ValueError("table_mul needs a contiguous complex table of the data's precision with min(n, n_out) entries")
    out = torch.empty(list(x.shape[:-1]) + [n_out], dtype=cdt, device=x.device)
    _lib.check(dll.xrfthip_table_mul(_DTYPES[x.dtype], batch, n_in, n_out, _ptr(x), _ptr(table), _ptr(out), _stream_handle(x)))
    return out


def device_type():
    """torch device type the bound library computes on ('cuda'; 'cpu' only for the emulated test build)."""
    return _lib.device()


def reduce_axis(x, axis, scale=1.0):
    """scale * sum of ``x`` over ``axis`` (a device kernel, float64 accumulation in index order: bit-reproducible); the mean over a
    batch dimension with scale = 1 / n (the reference's users average isotropic spectra over the batch, test_xrft.py:1011-1013)."""
    dll = _lib.load()
    if x.dtype not in _DTYPES:
        raise TypeError(f"reduce_axis: unsupported dtype {x.dtype}")
    x = x.contiguous()
    axis = axis % x.dim()
    outer = int(np.prod(x.shape[:axis], dtype=np.int64))
    inner = int(np.prod(x.shape[axis + 1:], dtype=np.int64))
    out = torch.empty(list(x.shape[:axis]) + list(x.shape[axis + 1:]), dtype=x.dtype, device=x.device)
    _lib.check(dll.xrfthip_reduce_axis(_DTYPES[x.dtype], outer, x.shape[axis], inner, _ptr(x), _ptr(out), float(scale), _stream_handle(x)))
    return out


def isotropize(x, binmap_dev, nbins):
    """Radial bin-sum of the last two axes of ``x`` with a device int32 bin map (xrft/xrft.py:993-1004)."""
    dll = _lib.load()
    ny, nx = x.shape[-2], x.shape[-1]
    batch = x.numel() // max(ny * nx, 1)
    cplx = x.is_complex()
    iso = torch.empty((batch, nbins), dtype=torch.complex128 if cplx else torch.float64, device=x.device)
    nws = int(dll.xrfthip_isotropize_workspace_bytes(_DTYPES[x.dtype], batch, ny, nx, nbins))
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=x.device)  # per-workgroup partial sums (added in a fixed order: bit-reproducible)
    _lib.check(dll.xrfthip_isotropize(_DTYPES[x.dtype], batch, ny, nx, _ptr(x), _ptr(binmap_dev), nbins, _ptr(iso), _ptr(ws), nws,
                                      _stream_handle(x)))
    return iso
