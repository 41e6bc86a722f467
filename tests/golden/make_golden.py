#!/usr/bin/env python3
"""
Generate the golden fixtures under tests/golden/.   Run ONLY in the build container:

    PYTHONDONTWRITEBYTECODE=1 python3 -B tests/golden/make_golden.py

Two families, kept in separate files so it is always clear what pinned what:

* ``ref_*.npz``   -- outputs of the REFERENCE'S OWN SOURCE (/root/reference/xrft) for the helpers that are
  pure numpy/scipy and therefore runnable here: ``_freq`` (xrft.py:139-155) and ``_detrend_2d_ufunc``
  (detrend.py:100-113).  xarray/dask are not installed, so the package is imported with throw-away stub
  modules (SURVEY.md header note 3); only functions that never touch a DataArray are called.
  These files pin the oracle (tests/test_oracle_golden.py).
* ``case_*.npz``  -- seeded inputs + the ORACLE's outputs for whole-path cases (power_spectrum, cross_spectrum,
  isotropic, dft).  They pin the HIP path against the oracle on the GPU box, where neither the reference
  nor this generator can run, and guard the oracle against accidental edits.

Fixtures are data only (inputs / expected outputs); no reference source text is stored.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True


def import_reference():
    """Import /root/reference/xrft with stub xarray/dask modules (read-only use, no bytecode written)."""
    for name in ("xarray", "xarray.core", "xarray.core.utils", "dask", "dask.array"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["xarray"].DataArray = object
    sys.modules["xarray.core.utils"].either_dict_or_kwargs = lambda *a, **k: None
    sys.modules["dask"].delayed = lambda *a, **k: None
    sys.modules["dask"].array = sys.modules["dask.array"]
    sys.path.insert(0, "/root/reference")
    import xrft.xrft as rx  # noqa
    import importlib
    rd = importlib.import_module("xrft.detrend")  # (the package attribute `detrend` is the function)
    rd = sys.modules["xrft.detrend"]
    return rx, rd


def gen_reference_files():
    rx, rd = import_reference()
    # ---- _freq
    out = {}
    cases = []
    i = 0
    for N in ([8], [9], [16], [4096], [65536], [16, 32], [9, 8], [720, 1440]):
        for dx0 in (1.0, 0.25, 2.0e4):
            for real in (None, "x"):
                for shift in (False, True):
                    dx = [dx0 * (1 + 0.5 * j) for j in range(len(N))]
                    k = rx._freq(N, dx, real, shift)
                    for j, kk in enumerate(k):
                        out[f"k_{i}_{j}"] = kk
                    cases.append((len(N),) + tuple(N) + (0,) * (2 - len(N)) + tuple(dx) + (0.0,) * (2 - len(N))
                                 + (0 if real is None else 1, int(shift)))
                    i += 1
    out["cases"] = np.array(cases, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "ref_freq.npz"), **out)
    # ---- _detrend_2d_ufunc
    rng = np.random.default_rng(20260927)
    out = {}
    for j, shape in enumerate([(32, 16), (16, 32), (64, 48), (15, 9)]):
        ny, nx = shape
        ii, jj = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
        arr = rng.standard_normal(shape) + 0.3 * ii - 0.7 * jj + 5.0
        out[f"in_{j}"] = arr
        out[f"out_{j}"] = rd._detrend_2d_ufunc(arr)
    np.savez_compressed(os.path.join(HERE, "ref_detrend2d.npz"), **out)
    print("reference-generated: ref_freq.npz, ref_detrend2d.npz")


def gen_reference_misc():
    """More helpers run from the reference's own source: _ifreq (xrft.py:158-175), fit_loglog (:1190-1210),
    _detrend_3d_ufunc (detrend.py:116-138), padding helpers (padding.py:277-323, 425-446)."""
    rx, rd = import_reference()
    import importlib
    sys.modules.setdefault("xrft.utils", types.ModuleType("xrft.utils")).get_spacing = lambda c: None
    rp = importlib.import_module("xrft.padding")
    out = {}
    cases = []
    i = 0
    for N in ([8], [9], [16, 9], [6, 5]):
        for real in (None, "x"):
            for shift in (False, True):
                dx = [0.5 * (1 + j) for j in range(len(N))]
                k = rx._ifreq(N, dx, real, shift)
                for j, kk in enumerate(k):
                    out[f"ik_{i}_{j}"] = kk
                cases.append((len(N),) + tuple(N) + (0,) * (2 - len(N)) + tuple(dx) + (0.0,) * (2 - len(N))
                             + (0 if real is None else 1, int(shift)))
                i += 1
    out["ifreq_cases"] = np.array(cases, dtype=np.float64)
    rng = np.random.default_rng(20260927 + 7)
    x = np.arange(1, 200, dtype=np.float64) * 0.01
    y = 3.5 * x ** -2.7 * np.exp(0.05 * rng.standard_normal(x.size))
    out["loglog_x"], out["loglog_y"] = x, y
    fit = rx.fit_loglog(x, y)  # (y_fit, a, b)
    out["loglog_yfit"], out["loglog_ab"] = np.asarray(fit[0], dtype=np.float64), np.array([fit[1], fit[2]], dtype=np.float64)
    for j, shape in enumerate([(6, 5, 4), (9, 16, 12), (3, 32, 8)]):
        ii, jj, kk = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
        arr = rng.standard_normal(shape) + 0.3 * ii - 0.7 * jj + 0.11 * kk + 5.0
        out[f"d3_in_{j}"] = arr
        out[f"d3_out_{j}"] = rd._detrend_3d_ufunc(arr)
    pcs = []
    for j, (c0, n, sp, pw) in enumerate([(0.0, 3, 1.0, (2, 2)), (-5.0, 7, 0.25, (1, 4)), (10.0, 5, -2.0, (3, 0)), (1e3, 16, 1e-3, (0, 5))]):
        coord = c0 + sp * np.arange(n)
        out[f"pad_in_{j}"] = coord
        out[f"pad_out_{j}"] = np.pad(coord, pad_width=pw, mode=rp._pad_coordinates_callback, spacing=sp)
        sl = rp._pad_width_to_slice(pw, n + sum(pw))
        pcs.append((sp, pw[0], pw[1], sl.start, sl.stop))
    out["pad_cases"] = np.array(pcs, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "ref_misc.npz"), **out)
    print("reference-generated: ref_misc.npz")


def gen_reference_bins():
    """Radial bin maps (ref_bins.npz): the reference's own expressions of xrft.py:975-981 and :921-923 evaluated with the
    installed pandas, pushed through the reference's ``_binned_agg`` (xrft.py:877-907) whose numpy_groupies.aggregate call is
    intercepted by a recorder: what is stored are the integer bin indices the reference hands to the aggregation, for the
    frequency grids the reference's ``_freq`` produces.  (numpy_groupies itself is not installed: the aggregate's
    arithmetic -- a per-bin sum / mean -- is not pinned here.)"""
    import pandas as pd

    rx, _ = import_reference()
    rec = {}
    ng = types.ModuleType("numpy_groupies")

    def aggregate(idx, a, func=None, size=None, fill_value=0, dtype=None, axis=-1):
        rec["idx"], rec["size"], rec["func"] = np.array(idx), int(size), func
        return np.zeros(a.shape[:-1] + (size,))

    ng.aggregate = aggregate
    sys.modules["numpy_groupies"] = ng
    out = {}
    cases = []
    for j, (N, dx, nfactor) in enumerate([((16, 32), (1.0, 1.0), 4), ((512, 512), (1.0, 1.0), 4), ((720, 1440), (0.25, 0.25), 4),
                                           ((256, 256), (2.0, 0.5), 2)]):
        ll, kk = rx._freq(list(N), list(dx), None, True)  # freq_y (fftdim[0]), freq_x (fftdim[1]), shifted
        nbins = int(min(kk.size, ll.size) / nfactor)  # xrft.py:977-979
        freq_r = np.sqrt(kk[:, None] ** 2 + ll[None, :] ** 2)  # xrft.py:980 (xarray broadcasting of k (dim fftdim[1]) and l)
        binned = pd.cut(np.ravel(freq_r), nbins)  # xrft.py:921
        indices = binned.codes.reshape(freq_r.shape)  # xrft.py:923
        rx._binned_agg(freq_r, indices, binned.categories.size, func="mean", fill_value=0, dtype=None)
        assert rec["size"] == binned.categories.size and rec["idx"].size == freq_r.size
        out[f"k_{j}"], out[f"l_{j}"] = kk, ll
        out[f"codes_{j}"] = rec["idx"].reshape(freq_r.shape).astype(np.int16)
        out[f"left_{j}"] = np.array([iv.left for iv in binned.categories])
        out[f"right_{j}"] = np.array([iv.right for iv in binned.categories])
        cases.append((N[0], N[1], dx[0], dx[1], nfactor, nbins))
    out["cases"] = np.array(cases, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "ref_bins.npz"), **out)
    del sys.modules["numpy_groupies"]
    print("reference-generated: ref_bins.npz")


def gen_oracle_cases():
    from oracle import xrft_oracle as o
    import warnings

    warnings.simplefilter("ignore")
    rng = np.random.default_rng(20260927 + 1)

    # (3) power_spectrum on the C1 shape family, reduced to (2, 64, 48) float64 to keep the file small
    nt, ny, nx = 2, 64, 48
    ii, jj = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    data = rng.standard_normal((nt, ny, nx)) + 0.01 * ii - 0.02 * jj + 3.0
    out = {"data": data, "dy": 0.5, "dx": 2.0}
    da = o.OArr(data, ("time", "y", "x"), {"time": np.arange(nt), "y": np.arange(ny) * 0.5, "x": np.arange(nx) * 2.0})
    n = 0
    combos = []
    for det in (None, "constant", "linear"):
        for win in (None, "hann"):
            for scaling in ("density", "spectrum"):
                for wc in (False, True):
                    if wc and win is None:
                        continue
                    ps = o.power_spectrum(da, dim=["y", "x"], detrend=det, window=win, scaling=scaling,
                                          window_correction=wc)
                    out[f"ps_{n}"] = ps.values
                    combos.append(f"{det}|{win}|{scaling}|{int(wc)}")
                    n += 1
    out["combos"] = np.array(combos)
    out["freq_y"] = ps.coord("freq_y")
    out["freq_x"] = ps.coord("freq_x")
    np.savez_compressed(os.path.join(HERE, "case_ps2d_f64.npz"), **out)

    # real_dim variant + float32 input
    data32 = data.astype(np.float32)
    da32 = o.OArr(data32, da.dims, da._coords_raw())
    ps = o.power_spectrum(da32, dim=["y"], real_dim="x", detrend="linear", window="hann")
    np.savez_compressed(os.path.join(HERE, "case_ps2d_f32_real.npz"), data=data32, ps=ps.values,
                        freq_y=ps.coord("freq_y"), freq_x=ps.coord("freq_x"))

    # (4) cross_spectrum on (2,16,16) with different coordinate origins (true-phase net factor)
    a = rng.standard_normal((2, 16, 16))
    b = rng.standard_normal((2, 16, 16))
    c1 = {"t": np.arange(2), "y": np.arange(16) * 0.5 + 3.0, "x": np.arange(16) * 0.25 - 1.0}
    c2 = {"t": np.arange(2), "y": np.arange(16) * 0.5 + 4.5, "x": np.arange(16) * 0.25 + 2.0}
    cs = o.cross_spectrum(o.OArr(a, ("t", "y", "x"), c1), o.OArr(b, ("t", "y", "x"), c2), dim=["y", "x"],
                          window="hann", detrend="constant")
    cs2 = o.cross_spectrum(o.OArr(a, ("t", "y", "x"), c1), o.OArr(b, ("t", "y", "x"), c2), dim=["y", "x"],
                           true_phase=False, scaling="spectrum")
    np.savez_compressed(os.path.join(HERE, "case_cs2d.npz"), a=a, b=b, y1=c1["y"], x1=c1["x"], y2=c2["y"],
                        x2=c2["x"], cs=cs.values, cs_nophase_spectrum=cs2.values)

    # (5) isotropic power / cross spectrum: (16,32) random and a (64,64) red-noise field (slope -3)
    r = rng.standard_normal((3, 16, 32))
    dar = o.OArr(r, ("t", "y", "x"), {"t": np.arange(3), "y": np.arange(16), "x": np.arange(32)})
    iso = o.isotropic_power_spectrum(dar, dim=["y", "x"], detrend="constant", window="hann")
    r2 = rng.standard_normal((3, 16, 32))
    dar2 = o.OArr(r2, dar.dims, dar._coords_raw())
    ics = o.isotropic_cross_spectrum(dar, dar2, dim=["y", "x"], window="hann")
    theta = np.stack([o.synthetic_field(64, 1.0, 10.0, -3.0, rng) for _ in range(2)])
    dth = o.OArr(theta, ("d0", "y", "x"), {"d0": np.arange(2), "y": np.arange(64), "x": np.arange(64)})
    iso2 = o.isotropic_power_spectrum(dth, dim=["y", "x"], detrend="constant", truncate=True)
    np.savez_compressed(os.path.join(HERE, "case_iso.npz"), r=r, r2=r2, iso=iso.values, iso_kr=iso.coord("freq_r"),
                        ics=ics.values, ics_kr=ics.coord("freq_r"), theta=theta, iso2=iso2.values,
                        iso2_kr=iso2.coord("freq_r"))

    # (6) 1-D dft: (4, 4096) float32 (the 65536-point row is exercised by seeded tests, not stored)
    x = rng.standard_normal((4, 4096)).astype(np.float32)
    dax = o.OArr(x, ("t", "x"), {"t": np.arange(4), "x": np.arange(4096) * 0.5})
    ft = o.dft(dax, dim="x")
    ft2 = o.fft(dax, dim="x", detrend="linear", window="hann")
    np.savez_compressed(os.path.join(HERE, "case_dft1d_f32.npz"), x=x, ft=ft.values.astype(np.complex64),
                        freq_x=ft.coord("freq_x"), ft_lin_hann=ft2.values)
    print("oracle-generated: case_ps2d_f64.npz, case_ps2d_f32_real.npz, case_cs2d.npz, case_iso.npz, case_dft1d_f32.npz")


if __name__ == "__main__" and "--bins-only" in sys.argv:
    gen_reference_bins()
    sys.exit(0)
if __name__ == "__main__" and "--misc-only" in sys.argv:
    gen_reference_misc()
    sys.exit(0)

if __name__ == "__main__":
    if os.path.isdir("/root/reference/xrft"):
        gen_reference_files()
    else:
        print("no /root/reference: skipping reference-generated files")
    if os.path.isdir("/root/reference/xrft"):
        gen_reference_misc()
        gen_reference_bins()
    gen_oracle_cases()
