"""xrft_amd -- the xrft spectral hot path (fft / dft / power_spectrum / cross_spectrum / isotropic_* / detrend)
on AMD MI355X (gfx950): hand-written HIP kernels in libxrft_hip.so behind the reference's call signatures.

    import xrft_amd as xrft
    ps = xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")

``da`` is an ``xrft_amd.DataArray`` (numpy or torch data) or, when xarray is installed, an ``xarray.DataArray``.
There is no CPU fallback: the compute entry points raise ``XrftHipUnavailable`` if the library is not built.
"""
from ._lib import XrftHipError, XrftHipUnavailable  # noqa: F401
from .labeled import Coordinate, DataArray  # noqa: F401
from .api import (clear_plan_cache, cross_phase, cross_spectrum, detrend, dft, fft, fit_loglog, idft, ifft,  # noqa: F401
                  isotropic_cross_spectrum, isotropic_power_spectrum, isotropize, power_spectrum)
from .engine import bluestein_in_float64  # noqa: F401
from .padding import get_spacing, pad, unpad  # noqa: F401

__version__ = "0.1.0"
__all__ = ["DataArray", "Coordinate", "fft", "ifft", "dft", "idft", "detrend", "power_spectrum", "cross_spectrum",
           "cross_phase", "isotropize",
           "isotropic_power_spectrum", "isotropic_cross_spectrum", "fit_loglog", "pad", "unpad", "get_spacing", "XrftHipError",
           "XrftHipUnavailable", "clear_plan_cache", "bluestein_in_float64"]
