#!/usr/bin/env python3
"""Throughput + roofline of the BASELINE.json configurations other than the bench line (parity-test cases) on one MI355X.
GFFT/s = points of one field / wall time; roofline = SURVEY.md 8(d) algorithmic bytes per point / wall / 8 TB/s.
Run on the GPU box: python scripts/bench_configs.py > gpurun_out/r02/bench_configs.txt"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xrft_amd as xrft
from xrft_amd import api
warnings.simplefilter("ignore")
dev = "cuda"

def timeit(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    if t * reps < 0.02:  # a sub-millisecond call: five of them measure the final synchronize; take 20 ms worth
        reps = int(0.02 / t) + 1
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / reps
    return t

def cube(shape, dtype):
    return torch.randn(shape, dtype=dtype, device=dev)

rows = []
def add(name, pts, bpp, t):
    path = next(reversed(api._plan_cache.values())).describe().strip().split("\n")[1].strip().split("]")[0] + "]"
    rows.append((name, pts / t / 1e9, t, bpp, bpp * pts / t / 8e12, path))

# C1: PS (4,256,256) f64 (the reference's CPU-runnable case; one call is host-time bound here)
x = cube((4, 256, 256), torch.float64); c = {"t": np.arange(4), "y": np.arange(256.), "x": np.arange(256.)}
da = xrft.DataArray(x, ("t", "y", "x"), c)
add("C1 PS (4,256,256) f64 linear+hann", x.numel(), 16, timeit(lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")))
# C2: dft 1-D (1024, 65536) f32
x = cube((1024, 65536), torch.float32); da = xrft.DataArray(x, ("t", "x"), {"x": np.arange(65536) * 0.5})
add("C2 dft 1-D (1024,65536) f32", x.numel(), 12, timeit(lambda: xrft.dft(da, dim="x")))
add("   power_spectrum 1-D (1024,65536) f32", x.numel(), 8, timeit(lambda: xrft.power_spectrum(da, dim="x")))
add("   power_spectrum 1-D linear+hann (window: slab-shaped table)", x.numel(), 8, timeit(lambda: xrft.power_spectrum(da, dim="x", detrend="linear", window="hann")))
del x, da
for shape in ((2048, 32768), (4096, 16384), (8192, 8192)):  # (the shorter register-resident rows: a resident set + start stagger on long batches, profiles/r06_rows_stagger.txt)
    x = cube(shape, torch.float32); da = xrft.DataArray(x, ("t", "x"), {"x": np.arange(shape[1]) * 0.5})
    add(f"   dft 1-D {shape} f32", x.numel(), 12, timeit(lambda: xrft.dft(da, dim="x")))
    add(f"   fft 1-D {shape} f32 (true phase)", x.numel(), 12, timeit(lambda: xrft.fft(da, dim="x")))
    del x, da
# C3 variants
x = cube((32, 4096, 4096), torch.float32); c = {"y": np.arange(4096.), "x": np.arange(4096.)}
da = xrft.DataArray(x, ("t", "y", "x"), c)
add("C3 PS (32,4096,4096) f32 linear+hann", x.numel(), 8, timeit(lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")))
add("   isotropic PS (32,4096,4096) f32", x.numel(), 4, timeit(lambda: xrft.isotropic_power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")))
add("   PS real_dim=x (32,4096,4096) f32 (half output)", x.numel(), 6, timeit(lambda: xrft.power_spectrum(da, dim=["y"], real_dim="x", detrend="linear", window="hann")))
da8 = xrft.DataArray(x[:16].contiguous(), ("t", "y", "x"), c)
add("   fft complex out (16,4096,4096) f32", da8.data.numel(), 12, timeit(lambda: xrft.fft(da8, dim=["y", "x"], detrend="linear", window="hann")))
del x, da, da8
# C4: cross + isotropic on two (64,2048,2048) f32 (one GPU's share of nt = 512 over 8 GPUs)
a = cube((64, 2048, 2048), torch.float32); b = cube((64, 2048, 2048), torch.float32); c = {"y": np.arange(2048.), "x": np.arange(2048.)}
d1 = xrft.DataArray(a, ("t", "y", "x"), c); d2 = xrft.DataArray(b, ("t", "y", "x"), c)
add("C4 cross_spectrum 2x(64,2048,2048) f32 hann", a.numel(), 16, timeit(lambda: xrft.cross_spectrum(d1, d2, dim=["y", "x"], window="hann")))
add("C4 isotropic_cross_spectrum", a.numel(), 8, timeit(lambda: xrft.isotropic_cross_spectrum(d1, d2, dim=["y", "x"], window="hann")))
add("C4 isotropic_power_spectrum", a.numel(), 4, timeit(lambda: xrft.isotropic_power_spectrum(d1, dim=["y", "x"], window="hann")))
add("   PS (64,2048,2048) f32 linear+hann", a.numel(), 8, timeit(lambda: xrft.power_spectrum(d1, dim=["y", "x"], detrend="linear", window="hann")))
del a, b, d1, d2
for n, nt in ((1024, 256), (512, 1024), (256, 4096)):
    x = cube((nt, n, n), torch.float32); c = {"y": np.arange(float(n)), "x": np.arange(float(n))}
    da = xrft.DataArray(x, ("t", "y", "x"), c)
    add(f"   PS ({nt},{n},{n}) f32 linear+hann", x.numel(), 8, timeit(lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")))
    del x, da
# C5: PS (64,1440,720) f64
x = cube((64, 1440, 720), torch.float64); da = xrft.DataArray(x, ("t", "lat", "lon"), {"lat": np.arange(1440) * .25, "lon": np.arange(720) * .25})
add("C5 PS (64,1440,720) f64 linear+hann", x.numel(), 16, timeit(lambda: xrft.power_spectrum(da, dim=["lat", "lon"], detrend="linear", window="hann")))
add("   same, detrend=None", x.numel(), 16, timeit(lambda: xrft.power_spectrum(da, dim=["lat", "lon"], window="hann")))
da32 = xrft.DataArray(x.float(), da.dims, da.coords)
add("   same shape, float32, linear+hann", x.numel(), 8, timeit(lambda: xrft.power_spectrum(da32, dim=["lat", "lon"], detrend="linear", window="hann")))
add("   isotropic PS (64,1440,720) f64 linear+hann", x.numel(), 8, timeit(lambda: xrft.isotropic_power_spectrum(da, dim=["lat", "lon"], detrend="linear", window="hann")))
add("   dft complex out (64,1440,720) f64 linear+hann", x.numel(), 24, timeit(lambda: xrft.dft(da, dim=["lat", "lon"], detrend="linear", window="hann")))
y64 = cube((64, 1440, 720), torch.float64); db = xrft.DataArray(y64, da.dims, da.coords)
add("   cross_spectrum 2x(64,1440,720) f64 linear+hann", x.numel(), 32, timeit(lambda: xrft.cross_spectrum(da, db, dim=["lat", "lon"], detrend="linear", window="hann")))
del y64, db
# a middle axis in place (XRFTHIP_AXIS_Y)
x = cube((64, 1024, 2048), torch.float32); da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(1024.), "x": np.arange(2048.)})
add("fft along the MIDDLE axis (64,1024,2048) f32, no copies", x.numel(), 12, timeit(lambda: xrft.fft(da, dim=["y"])))
add("   power_spectrum along it, linear+hann", x.numel(), 8, timeit(lambda: xrft.power_spectrum(da, dim=["y"], detrend="linear", window="hann")))
x = cube((64, 1000, 2048), torch.float64); da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(1000.), "x": np.arange(2048.)})
add("   the same, (64,1000,2048) f64", x.numel(), 16, timeit(lambda: xrft.power_spectrum(da, dim=["y"], detrend="linear", window="hann")))
x = cube((16, 4096, 2048), torch.float32); da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(4096.), "x": np.arange(2048.)})
add("   power_spectrum along a 4096-point middle axis (16,4096,2048) f32", x.numel(), 8, timeit(lambda: xrft.power_spectrum(da, dim=["y"], detrend="linear", window="hann")))
# short contiguous axis (ndim = 1)
x = cube((131072, 1024), torch.float32); da = xrft.DataArray(x, ("t", "x"), {"x": np.arange(1024.)})
add("power_spectrum 1-D (131072,1024) f32 linear+hann", x.numel(), 8, timeit(lambda: xrft.power_spectrum(da, dim=["x"], detrend="linear", window="hann")))
add("   fft 1-D (131072,1024) f32", x.numel(), 12, timeit(lambda: xrft.fft(da, dim=["x"])))
del x, da
x = cube((32768, 4096), torch.float32); da = xrft.DataArray(x, ("t", "x"), {"x": np.arange(4096.)})
add("   power_spectrum 1-D (32768,4096) f32 linear+hann (round 2: 89-92, generic passes)", x.numel(), 8, timeit(lambda: xrft.power_spectrum(da, dim=["x"], detrend="linear", window="hann")))
add("   fft 1-D (32768,4096) f32", x.numel(), 12, timeit(lambda: xrft.fft(da, dim=["x"])))
x = cube((64, 1000, 1000), torch.float32); da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(1000.), "x": np.arange(1000.)})
add("PS (64,1000,1000) f32 linear+hann", x.numel(), 8, timeit(lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")))
# lengths that joined the mixed-radix table in round 3 (the generic tile kernels before: (64,2000,2000) f32 95, (16,3000,3000) f64 34 GFFT/s)
for shape, dt in (((64, 1280, 2560), torch.float32), ((16, 2160, 4320), torch.float32), ((32, 1080, 2160), torch.float64), ((32, 1440, 2880), torch.float32), ((64, 640, 1280), torch.float64), ((64, 2000, 2000), torch.float32), ((16, 3000, 3000), torch.float32), ((16, 3000, 3000), torch.float64), ((16, 1800, 3600), torch.float32), ((32, 1800, 900), torch.float64), ((32, 2000, 2000), torch.float64)):
    x = cube(shape, dt); da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(float(shape[1])), "x": np.arange(float(shape[2]))})
    add(f"PS {shape} {'f32' if dt == torch.float32 else 'f64'} linear+hann", x.numel(), 8 if dt == torch.float32 else 16, timeit(lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")))
    del x, da
# two adjacent axes with the batch INNERMOST (xrfthip_desc.inner): no transposed copies
x = cube((1024, 1024, 64), torch.float32); da = xrft.DataArray(x, ("y", "x", "t"), {"y": np.arange(1024.), "x": np.arange(1024.)})
add("PS over (y, x) of a (1024,1024,64) f32 (y, x, t) array, linear+hann", x.numel(), 8, timeit(lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")))
add("   fft (complex) of the same, no detrend", x.numel(), 12, timeit(lambda: xrft.fft(da, dim=["y", "x"])))
del x, da
# a length with one awkward prime: the ERA5 grid (721 = 7 x 103 latitudes) -- the prime-factor form with Rader's algorithm in the column tile (round 5)
x = cube((64, 721, 1440), torch.float32); da = xrft.DataArray(x, ("t", "lat", "lon"), {"lat": np.arange(721) * .25, "lon": np.arange(1440) * .25})
add("PS (64,721,1440) f32 linear+hann (ERA5 grid, 721 = 7 x 103: Rader columns)", x.numel(), 8, timeit(lambda: xrft.power_spectrum(da, dim=["lat", "lon"], detrend="linear", window="hann")))
del x, da
# round 4: lengths as data (csrc/fastg.h) -- small slabs of any smooth shape, one transform axis on any smooth length
for shape, dt in (((14400, 50, 50), torch.float32), ((14400, 50, 50), torch.float64), ((8192, 96, 96), torch.float32), ((14400, 45, 45), torch.float32), ((2048, 150, 150), torch.float32)):
    x = cube(shape, dt); da = xrft.DataArray(x, ("t", "y", "x"), {"y": np.arange(float(shape[1])), "x": np.arange(float(shape[2]))})
    tag, bpp = ("f32" if dt == torch.float32 else "f64"), (8 if dt == torch.float32 else 16)
    add(f"PS {shape} {tag} linear+hann (small boxes, one pass)", x.numel(), bpp, timeit(lambda: xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")))
    if shape[1] == 50:
        add(f"   isotropic PS {shape} {tag}", x.numel(), bpp // 2, timeit(lambda: xrft.isotropic_power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")))
        db = xrft.DataArray(torch.roll(x, 3, dims=2) * 0.5, ("t", "y", "x"), {"y": np.arange(float(shape[1])), "x": np.arange(float(shape[2]))})
        add(f"   cross_spectrum 2x{shape} {tag}", x.numel(), 2 * bpp, timeit(lambda: xrft.cross_spectrum(da, db, dim=["y", "x"], detrend="linear", window="hann")))
        del db
    del x, da
for shape, dt in (((96, 512, 512), torch.float32), ((250, 512, 512), torch.float32), ((250, 256, 512), torch.float64), ((365, 512, 512), torch.float32), ((1460, 128, 256), torch.float64)):
    x = cube(shape, dt); da = xrft.DataArray(x, ("time", "y", "x"), {"time": np.arange(float(shape[0]))})
    tag, bpp = ("f32" if dt == torch.float32 else "f64"), (8 if dt == torch.float32 else 16)
    add(f"power_spectrum along time of (time, y, x) = {shape} {tag}, linear+hann", x.numel(), bpp, timeit(lambda: xrft.power_spectrum(da, dim="time", detrend="linear", window="hann")))
    del x, da
for shape, dt in (((131072, 250), torch.float32), ((65536, 96), torch.float32), ((65536, 50), torch.float64)):
    x = cube(shape, dt); da = xrft.DataArray(x, ("t", "x"), {"x": np.arange(float(shape[1]))})
    tag, bpp = ("f32" if dt == torch.float32 else "f64"), (8 if dt == torch.float32 else 16)
    add(f"power_spectrum 1-D {shape} {tag} linear+hann (a length outside the tables)", x.numel(), bpp, timeit(lambda: xrft.power_spectrum(da, dim="x", detrend="linear", window="hann")))
    del x, da
# round 5: the calendar's primes along the contiguous axis, small grids with an awkward prime, short columns, non-adjacent axes
for shape, dt in (((131072, 365), torch.float32), ((32768, 1460), torch.float64)):
    x = cube(shape, dt); da = xrft.DataArray(x, ("s", "time"), {"time": np.arange(float(shape[1]))})
    tag, bpp = ("f32" if dt == torch.float32 else "f64"), (8 if dt == torch.float32 else 16)
    add(f"power_spectrum 1-D {shape} {tag} linear+hann (365 = 5 x 73: Rader along the contiguous axis)", x.numel(), bpp, timeit(lambda: xrft.power_spectrum(da, dim="time", detrend="linear", window="hann")))
    del x, da
for shape, dt in (((4096, 73, 144), torch.float32), ((16384, 37, 72), torch.float32), ((512, 100, 2000), torch.float32), ((64, 721, 1440), torch.float64)):
    x = cube(shape, dt); da = xrft.DataArray(x, ("t", "lat", "lon"), {"lat": np.arange(float(shape[1])), "lon": np.arange(float(shape[2]))})
    tag, bpp = ("f32" if dt == torch.float32 else "f64"), (8 if dt == torch.float32 else 16)
    add(f"PS {shape} {tag} linear+hann", x.numel(), bpp, timeit(lambda: xrft.power_spectrum(da, dim=["lat", "lon"], detrend="linear", window="hann")))
    del x, da
for shape, dt in (((1460, 73, 144), torch.float32), ((1024, 64, 512), torch.float32), ((720, 91, 360), torch.float64)):
    x = cube(shape, dt); da = xrft.DataArray(x, ("time", "lat", "lon"), {"time": np.arange(float(shape[0])), "lon": np.arange(float(shape[2])) * 2.5})
    tag, bpp = ("f32" if dt == torch.float32 else "f64"), (8 if dt == torch.float32 else 16)
    add(f"PS over (time, lon) of (time, lat, lon) = {shape} {tag}, linear+hann (non-adjacent axes, fused)", x.numel(), bpp, timeit(lambda: xrft.power_spectrum(da, dim=["time", "lon"], detrend="linear", window="hann")))
    del x, da
# round 6: inverse transforms and transforms of complex data (csrc/fasty_c2c.h, csrc/fastr.h fastc_kernel); GFFT/s counts the points of the full (real or complex) field
def spec(shape, real_x=False):
    """an fftshifted spectrum with centred frequency coordinates (what xrft.fft returns), complex64; real_x: the stored half of the last axis (rfftfreq)"""
    dims = ("t", "freq_y", "freq_x")[-len(shape):]
    z = torch.randn(shape, dtype=torch.complex64, device=dev)
    c = {d: np.fft.fftshift(np.fft.fftfreq(n, 1.0)) for d, n in zip(dims, shape) if d != "t"}
    if real_x:
        c["freq_x"] = np.fft.rfftfreq(2 * (shape[-1] - 1), 1.0)
    return xrft.DataArray(z, dims, c)
for shape in ((16384, 4096), (131072, 1024), (4096, 16384), (1024, 65536), (64, 1048576)):  # (65536 points and beyond: the four-step form on fasty_c2c.h)
    F = spec(shape)
    add(f"ifft 1-D {shape} complex64", F.data.numel(), 16, timeit(lambda: xrft.ifft(F, dim="freq_x")))
    del F
for shape in ((16, 4096, 4096), (64, 2048, 2048), (256, 1024, 1024)):
    F = spec(shape)
    add(f"ifft 2-D {shape} complex64", F.data.numel(), 16, timeit(lambda: xrft.ifft(F, dim=["freq_y", "freq_x"])))
    del F
for shape in ((16, 4096, 2049), (64, 2048, 1025)):
    F = spec(shape, real_x=True)
    nfull = shape[0] * shape[1] * 2 * (shape[2] - 1)
    add(f"ifft real_dim (irfftn) {shape} complex64 -> real", nfull, 8, timeit(lambda: xrft.ifft(F, dim=["freq_y"], real_dim="freq_x")))
    del F
def spec128(shape):  # (the stored half spectrum of a float64 field)
    z = torch.randn(shape, dtype=torch.complex128, device=dev)
    return xrft.DataArray(z, ("t", "freq_y", "freq_x"), {"freq_y": np.fft.fftshift(np.fft.fftfreq(shape[1], 1.0)), "freq_x": np.fft.rfftfreq(2 * (shape[2] - 1), 1.0)})
F = spec128((64, 1440, 361))
add("ifft real_dim (irfftn) (64, 1440, 361) complex128 -> (64, 1440, 720) float64 (the C5 grid back; both stages on the table kernels)", 64 * 1440 * 720, 16,
    timeit(lambda: xrft.ifft(F, dim=["freq_y"], real_dim="freq_x")))
del F
F = spec((16384, 2049), real_x=True)
add("ifft real_dim 1-D (irfft) (16384, 2049) -> 4096 real samples", 16384 * 4096, 8, timeit(lambda: xrft.ifft(F, dim="freq_x", real_dim="freq_x")))
del F
z = torch.randn((16, 4096, 4096), dtype=torch.complex64, device=dev); dz = xrft.DataArray(z, ("t", "y", "x"), {"y": np.arange(4096.), "x": np.arange(4096.)})
add("fft 2-D of complex data (16,4096,4096) complex64", z.numel(), 16, timeit(lambda: xrft.fft(dz, dim=["y", "x"])))
add("   power_spectrum of the same", z.numel(), 12, timeit(lambda: xrft.power_spectrum(dz, dim=["y", "x"])))
del z, dz
print(f"{'workload':58s} {'GFFT/s':>8s} {'ms':>9s} {'B/pt':>5s} {'frac of 8 TB/s':>15s}  path")
for name, g, t, bpp, frac, path in rows:
    print(f"{name:58s} {g:8.2f} {t*1e3:9.3f} {bpp:5.0f} {frac:15.3f}  {path}")
