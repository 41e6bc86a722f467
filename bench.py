#!/usr/bin/env python3
"""bench.py -- headline benchmark of BASELINE.json:
    xrft.power_spectrum 2-D, detrend='linear' + Hann window, (nt, 4096, 4096) float32 per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W        (no launcher: the script starts its N ranks itself, same JSON line)

--workload c2 | c4 | c5 runs BASELINE.json configs[1] / [3] / [4] through the same harness (dft along x of (1024, 65536) float32;
cross_spectrum + isotropic_power_spectrum of two (nt, 2048, 2048) fields with the all_gathers; power_spectrum of (450, 1440, 720)
float64, the per-GPU share of the configuration's 3600 slabs) -- same JSON line, roofline on that configuration's own algorithmic bytes.

One "step" = one call of ``xrft_amd.power_spectrum`` over the rank's whole (nt, ny, nx) cube, input already
resident in HBM.  Batches shard over ranks as independent time slabs (weak scaling: nt per GPU is fixed); there is
no data-path collective.  value = GFFT/s = 1e-9 * (points transformed by all ranks) / (max-over-ranks wall time).

Extra objects on the JSON line:
  roofline     : achieved = 8 B/point (4 B read + 4 B written per input point) * points per step / wall
                 time of the step (all kernels + gaps), frac = achieved / 8 TB/s.  The timed region is the plain product call,
                 nothing recorded inside it; "kernel" holds the same figure for the longest kernel alone (its launch's points / its
                 average launch duration, from HIP events recorded by the library on the launch stream, xrfthip_plan_set_profiling,
                 over K more steps run right after the timed region); "traffic" the HBM bytes of one step measured with rocprofv3
                 PMC counters (profiles/r0X_traffic.json, used only when its stamp matches the SHA-1 of xrft_amd/csrc; null
                 otherwise); "measured_floor" = what the two passes' memory accesses cost with the transforms removed and what a
                 plain copy reaches, timed by the library itself IN THIS RUN (xrfthip_selftest_floor, csrc/selftest.h).
  cpu_baseline : the CPU oracle (numpy/scipy restatement of the reference; the reference itself needs xarray,
                 which the image lacks) timed on a bounded sample of the same workload, 1 thread.
  extra_workloads : (N = 1, the headline command) BASELINE.json configs[1] / [3] / [4] run as short legs in the same process after the
                 headline: value, ms_per_step, frac on that configuration's algorithmic bytes, kernels, parity of one unit vs the oracle.
"""
import argparse
import json
import os
import sys
import time
import warnings

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md)
BYTES_PER_POINT = 8.0  # f32 in (4 B) + f32 out (4 B) per input point


def _cpu_pool_init():
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = "1"


def _cpu_pool_slab(arg):
    """One slab through the oracle in a worker process (what dask chunks {time: 1} would give the reference)."""
    slab, coords = arg
    import warnings

    import numpy as np
    warnings.simplefilter("ignore")
    from oracle import xrft_oracle as oracle

    if slab is None:
        return 0.0
    try:
        from threadpoolctl import threadpool_limits
        lim = threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        lim = None
    oc = {"time": np.arange(1), "y": coords[0], "x": coords[1]}
    r = oracle.power_spectrum(oracle.OArr(slab[None], ("time", "y", "x"), oc), dim=["y", "x"], detrend="linear", window="hann")
    return float(r.values[0, 0, 0])


def own_share(name, workload, ny, nx, pts_per_launch, avg_s):
    """The dominant kernel's OWN compulsory bytes per input point (not the whole path's): a column pass reads the input and writes the
    half-spectrum intermediate ((ny/2 + 1) complex rows), a row pass reads that intermediate and writes the result."""
    esz = 8.0 if workload == "c5" else 4.0  # bytes of a real sample
    half = 2.0 * esz * (ny // 2 + 1) / max(ny, 1)  # intermediate, bytes per input point
    if "cols" in name:
        b, what = esz + half, "input read + half-spectrum intermediate written"
    elif "rows" in name:
        outb = {"ps": esz, "c5": esz, "c4": 0.0, "c2": 2 * esz}.get(workload, esz)
        b, what = half + outb, "half-spectrum intermediate read + result written"
    else:  # a one-pass kernel: the path's bytes ARE its own
        b = {"ps": 2 * esz, "c5": 2 * esz, "c2": 3 * esz, "c4": esz}.get(workload, 2 * esz)
        what = "input read + result written (one pass)"
    ach = b * pts_per_launch / avg_s
    return {"bytes_per_point": round(b, 3), "what": what, "achieved": round(ach / 1e9, 2), "frac": round(ach / HBM_PEAK, 4)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nt", type=int, default=None, help="time slabs per GPU (default: 64; c5: 450 = 3600 / 8 under weak scaling; c2: 1024 rows)")
    ap.add_argument("--ny", type=int, default=4096)
    ap.add_argument("--nx", type=int, default=4096)
    ap.add_argument("--cpu-slabs", type=int, default=10, help="slabs timed through the CPU oracle on one thread (0 = skip)")
    ap.add_argument("--cpu-pool", type=int, default=-1, help="worker processes for the all-cores CPU figure, one slab each "
                    "(-1 = as many as host cores, slabs and memory allow; 0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--no-extra", action="store_true", help="skip the short c2 / c4 / c5 legs that follow the headline at N = 1 (`extra_workloads`)")
    ap.add_argument("--extra-steps", type=int, default=10, help="timed steps of each extra leg")
    ap.add_argument("--no-floor", action="store_true", help="skip the library's memory-floor self-test (`roofline.measured_floor`)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --nt slabs PER GPU (default); strong: --nt slabs in total, contiguous blocks per rank")
    ap.add_argument("--workload", choices=["ps", "c2", "c4", "c5"], default="ps",
                    help="ps: BASELINE.json configs[2] (power_spectrum, the headline metric); c2: configs[1] -- dft along x of "
                         "(1024, 65536) float32; c4: configs[3] -- cross_spectrum + isotropic_power_spectrum of two fields per rank, the "
                         "isotropic results all-gathered over RCCL; c5: configs[4] -- power_spectrum of (450, 1440, 720) float64 per GPU (3600 / 8), linear detrend + Hann")
    args = ap.parse_args(argv)
    args.argv = list(sys.argv[1:] if argv is None else argv)
    return args


class GpuEnv:
    """Where the bench runs: one MI355X per rank, RCCL ("nccl") between ranks, the HIP library or nothing.  The only environment
    this script knows; tests/bench_ranks_harness.py drives run() with its own (gloo, CPU tensors, the emulated test build) to
    exercise the rank logic without a GPU."""
    backend = "nccl"
    backend_label = "nccl (RCCL over xGMI)"
    data_label = "synthetic"
    measures = True
    script = os.path.abspath(__file__)  # what launch_ranks() starts once per rank

    def visible_devices(self):
        import torch

        return torch.cuda.device_count()

    def device(self, local):
        import torch

        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        return dev

    def init_process_group(self, dist, local):
        import torch

        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    def sync(self, dev):
        import torch

        torch.cuda.synchronize(dev)

    def load_library(self):
        from xrft_amd import _lib

        _lib.load()  # no fallback: raises if the HIP library is missing


def csrc_sha1():
    """SHA-1 over the library's sources (file names + contents, sorted): what a committed traffic profile is stamped with."""
    import hashlib

    h = hashlib.sha1()
    d = os.path.join(REPO, "xrft_amd", "csrc")
    for name in sorted(os.listdir(d)):
        h.update(name.encode())
        with open(os.path.join(d, name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, env):
    """`python bench.py --gpus N` with no launcher around it (no WORLD_SIZE in the environment): start the N ranks here, one
    process per GPU under torch.distributed.run on 127.0.0.1 (rank -> GPU by LOCAL_RANK in run()), and hand their exit status
    back.  Rank 0 of the children prints the ONE JSON line on this process's stdout.  Fewer than N visible devices is an
    error, said before anything is started."""
    import subprocess

    have = env.visible_devices()
    if have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} device(s) visible "
                         f"(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES = {os.environ.get('HIP_VISIBLE_DEVICES')!r} / "
                         f"{os.environ.get('ROCR_VISIBLE_DEVICES')!r}); not starting any rank\n")
        return 3
    child_env = dict(os.environ)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver supports dmabuf IPC only (RCCL over xGMI needs it)
    child_env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), env.script] + list(args.argv)
    r = subprocess.run(cmd, env=child_env)
    if r.returncode != 0:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank run failed (exit status {r.returncode}): {' '.join(cmd)}\n")
    return r.returncode


def main(argv=None):
    out = run(parse_args(argv), GpuEnv())
    if isinstance(out, int):  # the exit status of a self-launched multi-rank run
        sys.exit(out)
    return out


# BASELINE.json configs[1] / [3] / [4] at their single-GPU sizes: the legs `extra_workloads` runs after the headline (N = 1)
EXTRA_LEGS = ("c2", "c4", "c5")


class Workload:
    """One BASELINE.json configuration on this rank: the synthetic input resident in HBM, step() = one pass of the hot path over it,
    the algorithmic bytes per point, the labels of the JSON line, and parity of ONE unit against the CPU oracle."""

    def __init__(self, name, args, env, dev, rank, world, dist, default_sizes=False):
        import numpy as np
        import torch

        import xrft_amd as xrft
        from xrft_amd import dist as xdist

        self.name, self.env, self.dev, self.rank, self.world, self.dist = name, env, dev, rank, world, dist
        ny, nx, nt = args.ny, args.nx, args.nt
        scaling = args.scaling
        if default_sizes:  # an extra leg: the configuration's own sizes, whatever the headline's flags were
            ny, nx, nt, scaling = 4096, 4096, None, "weak"
        default_shape = (ny, nx) == (4096, 4096)
        if name == "c4" and default_shape:
            ny = nx = 2048  # BASELINE.json configs[3]
        if name == "c5" and default_shape:
            ny, nx = 1440, 720  # configs[4]
            if nt is None and scaling == "weak":
                nt = 450  # the configuration's per-GPU share: 3600 slabs over 8 GPUs (3.7 GB in, 3.7 GB out)
        if name == "c2":
            ny, nx = 1, (65536 if default_shape else nx)  # configs[1]: (1024, 65536) per GPU, one long axis
            if nt is None:
                nt = 1024
        if nt is None:  # (an explicit --nt is always what runs)
            nt = 64
        self.ny, self.nx, self.scaling = ny, nx, scaling
        fdt = torch.float64 if name == "c5" else torch.float32
        self.dtype = "f64" if name == "c5" else "f32"
        # slabs of this rank: weak = --nt each; strong = contiguous block of --nt in total (never splits a slab)
        if scaling == "strong":
            lo, hi = xdist.shard_bounds(nt, rank, world)
            self.nt, self.nt_total = hi - lo, nt
        else:
            self.nt, self.nt_total = nt, nt * world
        nt = self.nt
        # ---- synthetic cube generated on the device: N(0,1) + plane + offset so that the linear detrend works
        gen = torch.Generator(device=dev)
        gen.manual_seed(20260927 + 1000 * {"ps": 3, "c4": 4, "c2": 2, "c5": 5}[name] + rank)
        x = torch.randn((nt, ny, nx), dtype=fdt, device=dev, generator=gen)
        x += (0.01 * torch.arange(ny, device=dev, dtype=fdt))[None, :, None]
        x += (-0.02 * torch.arange(nx, device=dev, dtype=fdt) * (4096.0 / nx) + 3.0)[None, None, :]
        self.x = x
        self.coords = coords = {"time": np.arange(nt), "y": np.arange(ny, dtype=np.float64), "x": np.arange(nx, dtype=np.float64)}
        da = xrft.DataArray(x, ("time", "y", "x"), coords)
        if name == "c2":
            da = xrft.DataArray(x.reshape(nt, nx), ("time", "x"), {"time": coords["time"], "x": coords["x"]})
        self.collective = None
        self.x2 = None
        if name == "c4":
            self.x2 = 0.5 * x + torch.randn((nt, ny, nx), dtype=torch.float32, device=dev, generator=gen)
            db = xrft.DataArray(self.x2, ("time", "y", "x"), coords)
        nt_total = self.nt_total
        if name in ("ps", "c5"):
            def step():
                return xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")
        elif name == "c2":
            def step():
                return xrft.dft(da, dim="x")
        else:
            nbins = min(ny, nx) // 4
            self.collective = {"op": "all_gather", "backend": env.backend_label,
                               "bytes_per_rank": int(-(-nt_total // world) * nbins * 16), "per_step": 2}

            def step():  # BASELINE.json configs[3]: cross spectrum (stays sharded) + isotropic power spectra of the two fields (gathered)
                cs = xrft.cross_spectrum(da, db, dim=["y", "x"], window="hann")
                ia = xrft.isotropic_power_spectrum(da, dim=["y", "x"], window="hann")
                ib = xrft.isotropic_power_spectrum(db, dim=["y", "x"], window="hann")
                if dist is not None:
                    ia = xdist.all_gather_batch(ia, "time", nt_total)
                    ib = xdist.all_gather_batch(ib, "time", nt_total)
                return cs, ia, ib
        self.step = step
        # c4: two float32 fields in, one complex64 cross spectrum out per point (16 B/point); the two isotropic calls read the
        # two fields again (4 B/point each, their output is negligible); c2: float32 in, complex64 out; c5: float64 in, float64 out
        self.bpp = {"ps": BYTES_PER_POINT, "c4": 16.0 + 8.0, "c2": 12.0, "c5": 16.0}[name]
        self.points_per_step = float(self.nt_total) * ny * nx  # points of ONE field transformed by all ranks per step
        if name == "ps":
            self.wl = f"xrft.power_spectrum dim=[y,x] detrend=linear window=hann on ({nt},{ny},{nx}) float32 per GPU (BASELINE.json configs[2])"
            self.metric = f"2-D power_spectrum GFFT/s (nt,{ny},{nx}) fp32"
            self.par = f"time-slab shards x{world}, no collective"
        elif name == "c5":
            self.wl = f"xrft.power_spectrum dim=[y,x] detrend=linear window=hann on ({nt},{ny},{nx}) float64 per GPU (BASELINE.json configs[4])"
            self.metric = f"2-D power_spectrum GFFT/s (nt,{ny},{nx}) fp64"
            self.par = f"time-slab shards x{world}, no collective"
        elif name == "c2":
            self.wl = f"xrft.dft dim=x on ({nt},{nx}) float32 per GPU (BASELINE.json configs[1])"
            self.metric = f"1-D dft GFFT/s (nt,{nx}) fp32"
            self.par = f"row shards x{world}, no collective"
        else:
            self.wl = (f"xrft.cross_spectrum + xrft.isotropic_power_spectrum (of each field) window=hann on two ({nt},{ny},{nx}) float32 fields "
                       f"per GPU (BASELINE.json configs[3]); GFFT/s counts the points of one field")
            self.metric = f"2-D cross_spectrum + isotropic_power_spectrum GFFT/s (nt,{ny},{nx}) fp32"
            self.par = (f"time-slab shards x{world}; full cross spectra stay sharded, one all_gather of the ({nt_total}, {min(ny, nx) // 4}) "
                        f"isotropic result per field and step")
        self.definition = {
            "ps": "algorithmic bytes (8 B per input point: 4 read + 4 written) of one step / wall time of the step, per GPU",
            "c4": "algorithmic bytes (cross spectrum 16 B + two isotropic power spectra 2 x 4 B per point of one field) of one step / wall time of the step, per GPU",
            "c2": "algorithmic bytes (12 B per point: float32 read + complex64 written) of one step / wall time of the step, per GPU",
            "c5": "algorithmic bytes (16 B per point: float64 read + float64 written) of one step / wall time of the step, per GPU"}[name]

    def barrier(self):
        self.env.sync(self.dev)
        if self.dist is not None:
            self.dist.barrier()
        self.env.sync(self.dev)

    def measure(self, steps, warmup, profile):
        """W warm-up steps, then EXACTLY K steps between barriers with nothing else switched on (the wall clock of the product call);
        then, outside the timed region, K more steps with the library's per-kernel HIP events for `roofline.kernel`.
        Returns (seconds of the timed region, {kernel: (launches, ms)} of the profiled steps, the last result, the plan)."""
        from xrft_amd import api

        # setup (not a step): prime torch's caching allocator so that no hipMalloc of a multi-GiB output lands in the timed
        # region -- a step holds the previous result while the next one is produced, i.e. two output blocks are live
        res = self.step()
        prev = res
        res = self.step()
        del prev
        self.barrier()
        plan = next(reversed(api._plan_cache.values())) if api._plan_cache else None
        for _ in range(warmup):
            res = self.step()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = self.step()
        self.barrier()
        dt = time.perf_counter() - t0
        prof = {}
        if plan is not None and profile:
            # the first launch that carries timestamps stalls its queue once (0.5 ms): one untimed step with the events on, records dropped
            plan.set_profiling(True)
            res = self.step()
            self.barrier()
            plan.set_profiling(True)  # (clears the records)
            for _ in range(steps):
                res = self.step()
            self.barrier()
            prof = plan.read_profile()
            plan.set_profiling(False)
        return dt, prof, res, plan

    def parity(self, res):
        """max |hip - oracle| / max |oracle| on the first unit of the workload (one slab / one row / one slab pair), the oracle = the CPU
        restatement of the reference (oracle/xrft_oracle.py) in float64 on the same samples."""
        import numpy as np

        from oracle import xrft_oracle as oracle

        def rel(got, ref):
            return float(np.abs(np.asarray(got) - ref).max() / np.abs(ref).max())

        c1 = {"time": np.arange(1), "y": self.coords["y"], "x": self.coords["x"]}
        if self.name in ("ps", "c5"):
            sub = self.x[:1].cpu().numpy()
            ref = oracle.power_spectrum(oracle.OArr(sub, ("time", "y", "x"), c1), dim=["y", "x"], detrend="linear", window="hann")
            return {"power_spectrum": rel(res.data[:1].cpu().numpy(), ref.values)}
        if self.name == "c2":
            sub = self.x[:2].reshape(2, self.nx).cpu().numpy()
            ref = oracle.dft(oracle.OArr(sub, ("time", "x"), {"time": np.arange(2), "x": self.coords["x"]}), dim="x")
            return {"dft": rel(res.data[:2].cpu().numpy(), ref.values)}
        cs, ia, ib = res
        a = oracle.OArr(self.x[:1].cpu().numpy().astype(np.float64), ("time", "y", "x"), c1)
        b = oracle.OArr(self.x2[:1].cpu().numpy().astype(np.float64), ("time", "y", "x"), c1)
        rcs = oracle.cross_spectrum(a, b, dim=["y", "x"], window="hann")
        ria = oracle.isotropic_power_spectrum(a, dim=["y", "x"], window="hann")
        rib = oracle.isotropic_power_spectrum(b, dim=["y", "x"], window="hann")
        return {"cross_spectrum": rel(cs.data[:1].cpu().numpy(), rcs.values),
                "isotropic_power_spectrum": max(rel(np.asarray(ia.values)[:1], ria.values), rel(np.asarray(ib.values)[:1], rib.values))}


def kernel_roofline(wl, prof, steps):
    """`roofline.kernel` + the per-kernel milliseconds of one step from the library's HIP events (recorded OUTSIDE the timed region)."""
    kern = dict(prof)
    dom = max(kern, key=lambda k: kern[k][1])
    launches, total_ms = kern[dom]
    avg_s = 1e-3 * total_ms / launches
    launches_per_step = launches / steps
    pts_per_launch = float(wl.nt) * wl.ny * wl.nx / max(launches_per_step, 1e-9)
    # the HIP events cover the LAST plan of the step: the whole step for ps / c2 / c5, one isotropic_power_spectrum call (one field read,
    # nothing but the radial sums written: 4 B per point) for c4
    bpp_prof = 4.0 if wl.name == "c4" else wl.bpp
    k_achieved = bpp_prof * pts_per_launch / avg_s
    kernel_ms = sum(v[1] for v in kern.values()) / steps
    return {"name": dom, "avg_launch_us": round(avg_s * 1e6, 2), "points_per_launch": pts_per_launch,
            "achieved": round(k_achieved / 1e9, 2), "frac": round(k_achieved / HBM_PEAK, 4),
            # what THIS kernel itself must move (a pass of a two-pass transform reads or writes the intermediate, too): the figure to hold
            # against the copy rate; `frac` above prices the whole path's bytes against one pass's time
            "own": own_share(dom, wl.name, wl.ny, wl.nx, pts_per_launch, avg_s),
            "definition": "algorithmic bytes OF THE WHOLE PATH for the slabs one launch of the longest kernel processes / its average launch "
                          "duration (HIP events on the launch stream, K profiled steps run right after the timed region); `own` = the bytes this "
                          "kernel alone has to move (input or intermediate read + intermediate or result written)"
                          + (" -- of one isotropic_power_spectrum call, the plan the events are recorded on: 4 B per point" if wl.name == "c4" else "")
            }, {k: round(v[1] / steps, 3) for k, v in kern.items()}, round(kernel_ms, 3)


def measured_floor(wl):
    """The memory floor of the headline path measured IN THIS RUN by the library's own skeleton kernels (csrc/selftest.h,
    xrfthip_selftest_floor): a plain copy, and the access patterns of the two passes with the transforms removed, on this run's input."""
    import ctypes as C

    import torch

    from xrft_amd import _lib

    if (wl.ny, wl.nx) != (4096, 4096) or wl.name != "ps":
        return None
    dll = _lib.load()
    nt = wl.nt
    w2 = torch.empty((nt, 2052, 4096, 2), dtype=torch.float32, device=wl.dev)
    out = torch.empty((nt, 4096, 4096), dtype=torch.float32, device=wl.dev)
    us = (C.c_double * 3)()
    reps = 5
    _lib.check(dll.xrfthip_selftest_floor(C.c_void_p(wl.x.data_ptr()), C.c_void_p(w2.data_ptr()), C.c_void_p(out.data_ptr()), nt, reps, us,
                                          C.c_void_p(torch.cuda.current_stream(wl.dev).cuda_stream)))
    del w2, out
    slab = 4096.0 * 4096.0
    copy_us, cols_us, rows_us = float(us[0]), float(us[1]), float(us[2])
    floor_us = cols_us + rows_us
    return {"measured_in_this_run": True, "reps": reps, "slabs": nt,
            "copy_us_per_slab": round(copy_us, 2), "copy_rate_measured_TBps": round(2 * 4 * slab / (copy_us * 1e-6) / 1e12, 3),
            "cols_skeleton_us_per_slab": round(cols_us, 2), "rows_skeleton_us_per_slab": round(rows_us, 2),
            "two_launch_floor_us_per_slab": round(floor_us, 2),
            "two_launch_floor_frac": round(BYTES_PER_POINT * slab / (floor_us * 1e-6) / HBM_PEAK, 4),
            "two_launch_floor_GFFTps": round(slab / (floor_us * 1e-6) / 1e9, 1),
            "what": "xrfthip_selftest_floor: the two passes' memory accesses with the transforms removed (same workgroup shape, LDS footprint, unit "
                    "order, store policy as fasty_cols / fasty_rows at 4096) and a plain 16-byte copy, HIP events, on this run's input"}


def traffic_from_profile(wl):
    """HBM bytes of one step: rocprofv3 cannot run inside the timed process, so the figure comes from the committed PMC profile of this same
    command -- and only if that profile was taken on the kernels that just ran (it is stamped with the SHA-1 of xrft_amd/csrc)."""
    traffic = tnote = None
    try:
        tname = {"c2": "traffic_c2.json", "ps": "traffic.json"}.get(wl.name)
        if tname is None or (wl.name == "c2" and wl.nx != 65536) or (wl.name == "ps" and (wl.ny, wl.nx) != (4096, 4096)):
            return None, None
        for rnd in ("r06", "r05"):
            path = os.path.join(REPO, "profiles", f"{rnd}_{tname}")
            if os.path.exists(path):
                break
        with open(path) as fh:
            tj = json.load(fh)
        if tj.get("csrc_sha1") == csrc_sha1():
            traffic = tj["path_hbm_bytes_per_slab"] * wl.nt
            tnote = tj.get("note")
        else:
            tnote = (f"{os.path.relpath(path, REPO)} was measured on other kernel sources (csrc SHA-1 {str(tj.get('csrc_sha1'))[:12]} != "
                     f"{csrc_sha1()[:12]}): re-run scripts/gpu_profile.sh")
    except Exception as e:
        tnote = f"no traffic profile: {e!r}"
    return traffic, tnote


def cpu_baseline(wl, res, args):
    """The CPU oracle (numpy pocketfft + the reference's plane-fit algorithm) on a bounded sample of the headline workload: 1 thread,
    then one slab per worker process over the host cores; and parity of the same slabs."""
    import numpy as np

    from oracle import xrft_oracle as oracle

    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    nt, ny, nx = wl.nt, wl.ny, wl.nx
    ns = min(args.cpu_slabs, nt)
    sub = wl.x[:ns].cpu().numpy()
    oc = {"time": np.arange(ns), "y": wl.coords["y"], "x": wl.coords["x"]}
    import contextlib

    limiter = threadpool_limits(limits=1) if threadpool_limits else contextlib.nullcontext()
    with limiter:
        t0 = time.perf_counter()
        ref = oracle.power_spectrum(oracle.OArr(sub, ("time", "y", "x"), oc), dim=["y", "x"], detrend="linear", window="hann")
        tc = time.perf_counter() - t0
    cpu = {"value": round(1e-9 * ns * ny * nx / tc, 6), "unit": "GFFT/s", "cores": 1, "kind": "port",
           "sample": f"{ns} of {nt} slabs ({ny}x{nx} f32) through oracle.power_spectrum(detrend='linear', "
                     f"window='hann') [numpy pocketfft + the reference's plane-fit algorithm], 1 thread, "
                     f"{tc:.1f} s; host has {os.cpu_count()} cores"}
    got = res.data[:ns].cpu().numpy()
    parity = float(np.abs(got - ref.values).max() / np.abs(ref.values).max())
    # one slab per worker (what dask chunks {time: 1} would give the reference): as many workers as cores, slabs in the
    # workload and memory allow (a slab's plane fit holds ~2.5 GB of float64 temporaries)
    ncores = os.cpu_count() or 1
    want = ncores if args.cpu_pool < 0 else args.cpu_pool
    try:
        import psutil
        mem_cap = max(1, int(psutil.virtual_memory().available * 0.6 / 3.0e9))
    except Exception:  # pragma: no cover
        mem_cap = 16
    npool = min(want, ncores, nt, mem_cap)
    if npool > 1:  # the same work spread over host cores, one slab per process (the reference would need dask for this)
        import multiprocessing as mp

        try:
            slabs = wl.x[:npool].cpu().numpy()
            with mp.get_context("spawn").Pool(npool, initializer=_cpu_pool_init) as pool:
                pool.map(_cpu_pool_slab, [(None, None)] * npool)  # workers up, numpy/scipy imported
                t0 = time.perf_counter()
                pool.map(_cpu_pool_slab, [(slabs[i], (wl.coords["y"], wl.coords["x"])) for i in range(npool)], chunksize=1)
                tp = time.perf_counter() - t0
            cpu["all_cores"] = {"value": round(1e-9 * npool * ny * nx / tp, 6), "unit": "GFFT/s", "cores": npool,
                                "sample": f"{npool} slabs, one per worker process (1 thread each), {tp:.1f} s; workers = "
                                          f"min(host cores {ncores}, slabs in the workload {nt}, memory cap {mem_cap})"}
        except Exception as e:  # pragma: no cover
            cpu["all_cores"] = {"error": repr(e)}
    return cpu, parity


def extra_leg(name, args, env, dev):
    """One of BASELINE.json configs[1] / [3] / [4] at its own single-GPU size, in this process, after the headline: value, ms per step, the
    roofline fraction on that configuration's algorithmic bytes, its kernels, and parity of one unit against the oracle."""
    import gc

    import torch

    from xrft_amd import api

    api.clear_plan_cache()
    wl = Workload(name, args, env, dev, 0, 1, None, default_sizes=True)
    steps = max(args.extra_steps, 50) if name == "c2" else args.extra_steps  # (c2's step is one 0.18-ms kernel: the barrier + synchronize bracket of a timed region is 0.4 ms)
    dt, prof, res, plan = wl.measure(steps, 1, True)
    value = 1e-9 * wl.points_per_step * steps / dt
    leg = {"metric": wl.metric, "value": round(value, 3), "unit": "GFFT/s", "steps": steps, "warmup": 1,
           "ms_per_step": round(1e3 * dt / steps, 3), "dtype": wl.dtype, "bytes_per_point": wl.bpp,
           "frac": round(wl.bpp * value * 1e9 / HBM_PEAK, 4), "config": {"workload": wl.wl, "nt_per_gpu": wl.nt, "ny": wl.ny, "nx": wl.nx}}
    if prof:
        k, per_kernel, total = kernel_roofline(wl, prof, steps)
        leg["kernel"] = {kk: k[kk] for kk in ("name", "avg_launch_us", "achieved", "frac", "own")}
        leg["kernels_ms_per_step"] = per_kernel
    try:
        leg["parity_max_rel_err_vs_oracle"] = wl.parity(res)
    except Exception as e:  # pragma: no cover
        leg["parity_max_rel_err_vs_oracle"] = {"error": repr(e)}
    if plan is not None:
        leg["plan"] = plan.describe().strip().split("\n")[:3]
    del wl, res, plan
    gc.collect()
    api.clear_plan_cache()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    return leg


def run(args, env):
    import numpy as np
    import torch

    warnings.simplefilter("ignore")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, env)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.init_process_group(dist, local)
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if dist is not None and dist.get_world_size() != world:
        raise SystemExit(f"bench.py: the process group reports {dist.get_world_size()} ranks, WORLD_SIZE={world}")
    dev = env.device(local)

    env.load_library()
    ranks_reported = dist.get_world_size() if dist is not None else 1  # what the process group (RCCL on GPUs) says, not the env
    wl = Workload(args.workload, args, env, dev, rank, world, dist)
    nt, nt_total, ny, nx = wl.nt, wl.nt_total, wl.ny, wl.nx
    collective = wl.collective

    dt, prof, ps, plan = wl.measure(args.steps, args.warmup, not args.no_profile)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # after the timed region: what every rank holds, so that a multi-GPU line can be judged on sight -- the shard sizes, and for the c4
    # workload whether the all-gathered isotropic blocks are the same bytes on every rank
    shard_sizes = [nt]
    if dist is not None:
        import hashlib

        shard_sizes = [None] * world
        dist.all_gather_object(shard_sizes, int(nt))
        if collective is not None:
            _cs, ia_, ib_ = ps
            dig = hashlib.sha1(np.ascontiguousarray(np.asarray(ia_.values)).tobytes() + np.ascontiguousarray(np.asarray(ib_.values)).tobytes()).hexdigest()
            digs = [None] * world
            dist.all_gather_object(digs, dig)
            collective["gathered_shape"] = [int(v) for v in np.asarray(ia_.values).shape]
            collective["identical_on_all_ranks"] = len(set(digs)) == 1

    value = 1e-9 * wl.points_per_step * args.steps / dt
    ms_per_step = 1e3 * dt / args.steps

    out = None
    if rank == 0:
        # ---- roofline: achieved = algorithmic bytes / wall of the whole hot path; the dominant kernel's own
        # figure (algorithmic bytes of its launch / its average launch duration) is kept beside it as `kernel`
        roof = None
        if prof:
            kroof, per_kernel, kernel_ms = kernel_roofline(wl, prof, args.steps)
            path_achieved = wl.bpp * value * 1e9 / world  # B/s per GPU
            traffic, tnote = traffic_from_profile(wl)
            floor = None
            if world == 1 and env.measures and not args.no_floor:
                try:
                    floor = measured_floor(wl)
                except Exception as e:  # pragma: no cover
                    floor = {"error": repr(e)}
            roof = {
                "bound": "hbm", "achieved": round(path_achieved / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(path_achieved / HBM_PEAK, 4),
                "definition": wl.definition,
                "traffic": traffic, "traffic_note": tnote,
                "kernel": kroof,
                "bytes_per_point": wl.bpp,
                "kernels_ms_per_step": per_kernel,
                "sum_kernels_ms_per_step": kernel_ms,
                "timed_region": "K steps between barriers with no per-kernel events recorded; `kernel` and `kernels_ms_per_step` come from K more "
                                "steps run right after it with the library's HIP events on",
                "measured_floor": floor,
            }
        # ---- CPU baseline (the oracle on a bounded sample, 1 thread) + parity of the same slabs
        cpu = None
        parity = None
        if args.cpu_slabs > 0 and world == 1 and args.workload == "ps" and env.measures:
            cpu, parity = cpu_baseline(wl, ps, args)
        out = {
            "metric": wl.metric, "value": round(value, 3), "unit": "GFFT/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": wl.dtype, "data": env.data_label,
            "config": {"workload": wl.wl, "nt_per_gpu": nt, "nt_total": nt_total, "ny": ny, "nx": nx, "parallelism": wl.par,
                       "collective": collective, "shard_sizes": shard_sizes, "slabs_per_s": round(nt_total * args.steps / dt, 2),
                       "ranks_in_process_group": ranks_reported, "process_group_backend": env.backend if dist is not None else None},
            "roofline": roof, "cpu_baseline": cpu, "parity_max_rel_err_vs_oracle": parity,
        }
        if plan is not None:
            out["plan"] = plan.describe().strip().split("\n")
        # ---- the other BASELINE.json configurations, in this same process (N = 1, the headline command only): short legs after the headline
        if world == 1 and args.workload == "ps" and env.measures and not args.no_extra and (ny, nx) == (4096, 4096):
            import gc

            del ps, wl, plan
            gc.collect()
            from xrft_amd import api

            api.clear_plan_cache()
            torch.cuda.empty_cache()
            legs = {}
            for name in EXTRA_LEGS:
                try:
                    legs[name] = extra_leg(name, args, env, dev)
                except Exception as e:  # pragma: no cover  (a leg that fails says so; the headline line still goes out)
                    legs[name] = {"error": repr(e)}
            out["extra_workloads"] = legs
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    main()
