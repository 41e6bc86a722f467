"""Shared cases for the backend module xrft_amd.fftmod (the object `_fft_module` can return, xrft/xrft.py:32-36): the six
functions the reference calls, with the reference's call shapes, against numpy.fft.  Imported by the emulator test (CPU) and
the GPU test."""
import numpy as np

CALLS = [
    # (function, input kind, shape, axes) -- the reference's call sites: xrft.py:400 (rfftn), :444 (fftn), :612 (irfftn), :614 (ifftn)
    ("fftn", "real", (3, 16, 24), [1, 2]),
    ("fftn", "complex", (3, 16, 24), [1, 2]),
    ("fftn", "real", (4, 30), [1]),
    ("fftn", "complex", (5, 12, 7), [0]),
    ("fftn", "real", (5, 12, 7), [1]),
    ("fftn", "complex", (4, 6, 8, 10), [1, 2, 3]),
    ("fftn", "real", (6, 8, 10), [0, 2]),
    ("rfftn", "real", (3, 16, 24), [1, 2]),
    ("rfftn", "real", (4, 30), [1]),
    ("rfftn", "real", (3, 9, 15), [1, 2]),
    ("rfftn", "real", (5, 12, 8), [0, 2]),
    ("rfftn", "real", (5, 12, 8), [2, 1]),
    ("ifftn", "complex", (3, 16, 24), [1, 2]),
    ("ifftn", "complex", (5, 12, 7), [0]),
    ("ifftn", "complex", (4, 6, 8, 10), [1, 2, 3]),
    ("irfftn", "complex", (3, 16, 13), [1, 2]),
    ("irfftn", "complex", (4, 16), [1]),
    ("irfftn", "complex", (5, 12, 5), [0, 2]),
    # lengths with a prime factor too large for one LDS tile (numpy.fft takes any length): Bluestein through global memory
    ("fftn", "complex", (2, 9001), [1]),
    ("fftn", "real", (2, 6, 9001), [1, 2]),
    ("ifftn", "complex", (2, 9001, 4), [1]),
    ("fftn", "complex", (9001, 3), [0]),
    ("rfftn", "real", (2, 9001), [1]),
    ("rfftn", "real", (2, 6, 9001), [1, 2]),
]


def run_all(fftmod, dtypes=("float64", "float32")):
    rng = np.random.default_rng(12)
    worst = 0.0
    for dt in dtypes:
        cdt = "complex128" if dt == "float64" else "complex64"
        tol = 1e-11 if dt == "float64" else 2e-5
        for fn, kind, shape, axes in CALLS:
            v = rng.standard_normal(shape)
            if kind == "complex":
                v = v + 1j * rng.standard_normal(shape)
            v = v.astype(cdt if kind == "complex" else dt)
            got = getattr(fftmod, fn)(v, axes=axes)
            want = getattr(np.fft, fn)(v, axes=axes)
            g = got.cpu().numpy()
            assert g.shape == want.shape, (fn, shape, axes, g.shape, want.shape)
            assert g.dtype == (np.dtype(dt) if fn == "irfftn" else np.dtype(cdt)), (fn, g.dtype)
            err = np.abs(g - want).max() / max(np.abs(want).max(), 1e-300)
            assert err < tol, (fn, kind, shape, axes, dt, err)
            worst = max(worst, err)
        for shape, axes in (((3, 16, 24), [1, 2]), ((4, 9), [1]), ((5, 7, 6), [0, 2]), ((5, 7, 6), None)):
            v = rng.standard_normal(shape).astype(dt)
            for fn in ("fftshift", "ifftshift"):
                assert np.array_equal(getattr(fftmod, fn)(v, axes=axes).cpu().numpy(), getattr(np.fft, fn)(v, axes=axes)), (fn, shape, axes)
            z = (v + 1j * v[::-1]).astype(cdt)
            assert np.array_equal(fftmod.fftshift(z, axes=axes).cpu().numpy(), np.fft.fftshift(z, axes=axes))
    return worst
