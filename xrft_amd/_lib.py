"""ctypes binding of libxrft_hip.so (C ABI in include/xrft_hip.h).

The library is the product: there is NO CPU fallback.  If the shared object is missing or cannot be loaded
the import of any compute entry point fails loudly (``XrftHipUnavailable``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libxrft_hip.so"

# ---- enums mirrored from include/xrft_hip.h
F32, F64, C64, C128 = 0, 1, 2, 3
OUT_COMPLEX, OUT_POWER, OUT_CROSS, OUT_PHASE = 0, 1, 2, 3
DETREND_NONE, DETREND_CONSTANT, DETREND_LINEAR = 0, 1, 2
HALF_X, SHIFT_Y, SHIFT_X, ISHIFT_Y, ISHIFT_X, FLIP_Y, FLIP_X = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20, 0x40
REALDIM_X2, ISO, NO_SPECTRUM_OUT = 0x80, 0x100, 0x200
INVERSE, C2R_X, PHASE_IN = 0x400, 0x800, 0x1000
BAD_ARG = -1  # xrfthip_status
UNSUPPORTED_LENGTH = -2
AXIS_Y = 0x2000  # transform y of [batch][ny][nx] in place (a middle or first axis of the array), no transposed copy
FLIP0_Y, FLIP0_X = 0x4000, 0x8000  # cross spectra: flip field 0 (FLIP_Y / FLIP_X then flip field 1)
HALF_Y = 0x10000  # inner / mid layouts (ABI 0.1.6): real_dim along the FIRST of the two transform axes (ny / 2 + 1 rows out)
# xrfthip_kernel_kind (xrfthip_plan_kernel_info)
K_GENERIC, K_FASTY, K_FASTM, K_FASTN, K_FASTM_Y, K_FASTM_X, K_FASTG_Y, K_FASTG_ROWS, K_FASTG, K_FASTS, K_FASTR, K_COMPOSITE = range(12)

EXPORTS = [
    "xrfthip_version", "xrfthip_strerror", "xrfthip_last_hip_error", "xrfthip_plan_create",
    "xrfthip_plan_destroy", "xrfthip_plan_set_window", "xrfthip_plan_set_phase", "xrfthip_plan_set_binmap",
    "xrfthip_plan_set_profiling", "xrfthip_plan_profile_read", "xrfthip_workspace_bytes", "xrfthip_plan_describe", "xrfthip_exec", "xrfthip_detrend_workspace_bytes",
    "xrfthip_detrend", "xrfthip_detrend3", "xrfthip_spectrum_tail", "xrfthip_spectrum_tail_axis", "xrfthip_gather_axis", "xrfthip_isotropize",
    "xrfthip_isotropize_workspace_bytes", "xrfthip_table_mul", "xrfthip_reduce_axis", "xrfthip_detrend_inner_workspace_bytes", "xrfthip_detrend_inner", "xrfthip_angle",
    "xrfthip_plan_uses_bluestein", "xrfthip_convert", "xrfthip_plan_kernel_info", "xrfthip_selftest_floor",
]


class XrftHipUnavailable(RuntimeError):
    pass


class XrftHipError(RuntimeError):
    def __init__(self, status, msg, hip_error=0):
        super().__init__(f"xrfthip status {status}: {msg}" + (f" (hipError_t {hip_error})" if hip_error else ""))
        self.status = status
        self.hip_error = hip_error


class Desc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("ndim", C.c_int32), ("batch", C.c_int64), ("ny", C.c_int64),
        ("nx", C.c_int64), ("dtype", C.c_int32), ("out_mode", C.c_int32), ("detrend", C.c_int32),
        ("flags", C.c_uint32), ("scale", C.c_double), ("slabs_per_group", C.c_int32), ("reserved", C.c_int32),
        ("inner", C.c_int64),  # > 1: arrays are [batch][ny][nx][inner], two adjacent transform axes with the independent elements innermost
        ("mid", C.c_int64),    # > 1: ... and `mid` independent elements between the two transform axes: [batch][ny][mid][nx][inner]
    ]


def _bind(dll):
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    dll.xrfthip_version.restype = C.c_int
    dll.xrfthip_strerror.restype = C.c_char_p
    dll.xrfthip_strerror.argtypes = [C.c_int]
    dll.xrfthip_last_hip_error.restype = C.c_int
    dll.xrfthip_plan_create.argtypes = [C.POINTER(vp), C.POINTER(Desc)]
    dll.xrfthip_plan_destroy.argtypes = [vp]
    dll.xrfthip_plan_set_window.argtypes = [vp, C.c_int, vp, i64]
    dll.xrfthip_plan_set_phase.argtypes = [vp, C.c_int, vp, i64]
    dll.xrfthip_plan_set_binmap.argtypes = [vp, vp, i64, i64, i32]
    dll.xrfthip_plan_set_profiling.argtypes = [vp, C.c_int]
    dll.xrfthip_plan_profile_read.argtypes = [vp, C.c_char_p, sz]
    dll.xrfthip_workspace_bytes.restype = sz
    dll.xrfthip_workspace_bytes.argtypes = [vp]
    dll.xrfthip_plan_describe.argtypes = [vp, C.c_char_p, sz]
    dll.xrfthip_exec.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp]
    dll.xrfthip_detrend_workspace_bytes.restype = sz
    dll.xrfthip_detrend_workspace_bytes.argtypes = [i64]
    dll.xrfthip_detrend.argtypes = [i32, i32, i64, i64, i64, i32, vp, vp, vp, sz, vp]
    dll.xrfthip_detrend3.argtypes = [i32, i64, i64, i64, i64, i32, vp, vp, vp, sz, vp]
    dll.xrfthip_spectrum_tail.argtypes = [i32, i64, vp, vp, vp, C.c_double, vp]
    dll.xrfthip_spectrum_tail_axis.argtypes = [i32, i64, i64, i64, i32, vp, vp, vp, C.c_double, vp]
    dll.xrfthip_gather_axis.argtypes = [i32, i64, i64, i64, i64, vp, i64, vp, vp, vp]
    dll.xrfthip_table_mul.argtypes = [i32, i64, i64, i64, vp, vp, vp, vp]
    dll.xrfthip_reduce_axis.argtypes = [i32, i64, i64, i64, vp, vp, C.c_double, vp]
    dll.xrfthip_angle.argtypes = [i32, i64, vp, vp, vp]
    dll.xrfthip_plan_uses_bluestein.argtypes = [vp]
    dll.xrfthip_plan_kernel_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    dll.xrfthip_convert.argtypes = [i32, i32, i64, vp, vp, vp]
    dll.xrfthip_detrend_inner_workspace_bytes.restype = sz
    dll.xrfthip_detrend_inner_workspace_bytes.argtypes = [i32, i64, i64]
    dll.xrfthip_detrend_inner.argtypes = [i32, i32, i64, i64, i64, i64, i32, vp, vp, vp, sz, vp]
    dll.xrfthip_isotropize_workspace_bytes.restype = sz
    dll.xrfthip_isotropize_workspace_bytes.argtypes = [i32, i64, i64, i64, i32]
    dll.xrfthip_isotropize.argtypes = [i32, i64, i64, i64, vp, vp, i32, vp, vp, sz, vp]
    dll.xrfthip_selftest_floor.argtypes = [vp, vp, vp, i64, i32, C.POINTER(C.c_double), vp]
    for name in EXPORTS:
        getattr(dll, name)  # AttributeError here = the .so does not export what the header declares
    return dll


_state = {"dll": None, "path": None, "device": "cuda"}


def lib_path():
    return os.path.join(_HERE, LIB_NAME)


def load(path=None):
    """Load (once) and return the bound library.  Raises XrftHipUnavailable if it is not there."""
    if _state["dll"] is not None and path is None:
        return _state["dll"]
    p = path or lib_path()
    if not os.path.exists(p):
        raise XrftHipUnavailable(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). xrft_amd has no CPU fallback.")
    try:
        dll = C.CDLL(p)
    except OSError as e:  # pragma: no cover
        raise XrftHipUnavailable(f"cannot load {p}: {e}") from e
    _state["dll"] = _bind(dll)
    _state["path"] = p
    return _state["dll"]


def _load_for_testing(path, device="cpu"):
    """TEST HOOK (tests/emu only): bind an emulated build of the same C ABI whose 'device' memory is host
    memory, so host logic and kernel index arithmetic can be exercised without a GPU.  Never called by the
    product; there is no environment variable or automatic switch that reaches this."""
    _state["dll"] = None
    load(path)
    _state["device"] = device
    return _state["dll"]


def device():
    """torch device the bound library computes on ('cuda' for the real library)."""
    return _state["device"]


def check(status):
    if status != 0:
        dll = load()
        raise XrftHipError(status, dll.xrfthip_strerror(status).decode(), dll.xrfthip_last_hip_error())
