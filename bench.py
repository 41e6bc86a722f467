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
                 time of the step (all kernels + gaps), frac = achieved / 8 TB/s.  "kernel" holds the same figure for the
                 longest kernel alone (its launch's points / its average launch duration, from HIP events recorded by
                 the library on the launch stream inside the timed region, xrfthip_plan_set_profiling); "traffic" the
                 HBM bytes of one step measured with rocprofv3 PMC counters (profiles/r05_traffic.json, used only when its
                 stamp matches the SHA-1 of xrft_amd/csrc; null otherwise); "claimed_floor" is NOT a measurement of this run: the
                 builder's claim of what the two passes' access patterns cost with no arithmetic (scripts/ubench/fused.hip as
                 timed in profiles/r03_ubench_fused.txt), carried with the SHA-1 of that skeleton's source and of the kernel
                 sources it was shaped after, and dropped (null) once either no longer matches the tree.
  cpu_baseline : the CPU oracle (numpy/scipy restatement of the reference; the reference itself needs xarray,
                 which the image lacks) timed on a bounded sample of the same workload, 1 thread.
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

    import xrft_amd as xrft
    from xrft_amd import api
    from xrft_amd import dist as xdist

    env.load_library()
    ranks_reported = dist.get_world_size() if dist is not None else 1  # what the process group (RCCL on GPUs) says, not the env
    ny, nx = args.ny, args.nx
    default_shape = (args.ny, args.nx) == (4096, 4096)
    if args.workload == "c4" and default_shape:
        ny = nx = 2048  # BASELINE.json configs[3]
    if args.workload == "c5" and default_shape:
        ny, nx = 1440, 720  # configs[4]
        if args.nt is None and args.scaling == "weak":
            args.nt = 450  # the configuration's per-GPU share: 3600 slabs over 8 GPUs (3.7 GB in, 3.7 GB out)
    if args.workload == "c2":
        ny, nx = 1, (65536 if default_shape else args.nx)  # configs[1]: (1024, 65536) per GPU, one long axis
        if args.nt is None:
            args.nt = 1024
    if args.nt is None:  # (an explicit --nt is always what runs)
        args.nt = 64
    fdt = torch.float64 if args.workload == "c5" else torch.float32
    # slabs of this rank: weak = --nt each; strong = contiguous block of --nt in total (never splits a slab)
    if args.scaling == "strong":
        lo, hi = xdist.shard_bounds(args.nt, rank, world)
        nt, nt_total = hi - lo, args.nt
    else:
        nt, nt_total = args.nt, args.nt * world

    # ---- synthetic cube generated on the device: N(0,1) + plane + offset so that the linear detrend works
    gen = torch.Generator(device=dev)
    gen.manual_seed(20260927 + 1000 * {"ps": 3, "c4": 4, "c2": 2, "c5": 5}[args.workload] + rank)
    x = torch.randn((nt, ny, nx), dtype=fdt, device=dev, generator=gen)
    x += (0.01 * torch.arange(ny, device=dev, dtype=fdt))[None, :, None]
    x += (-0.02 * torch.arange(nx, device=dev, dtype=fdt) * (4096.0 / nx) + 3.0)[None, None, :]
    coords = {"time": np.arange(nt), "y": np.arange(ny, dtype=np.float64), "x": np.arange(nx, dtype=np.float64)}
    da = xrft.DataArray(x, ("time", "y", "x"), coords)
    if args.workload == "c2":
        da = xrft.DataArray(x.reshape(nt, nx), ("time", "x"), {"time": coords["time"], "x": coords["x"]})
    collective = None
    if args.workload == "c4":
        x2 = 0.5 * x + torch.randn((nt, ny, nx), dtype=torch.float32, device=dev, generator=gen)
        db = xrft.DataArray(x2, ("time", "y", "x"), coords)

    if args.workload in ("ps", "c5"):
        def step():
            return xrft.power_spectrum(da, dim=["y", "x"], detrend="linear", window="hann")
    elif args.workload == "c2":
        def step():
            return xrft.dft(da, dim="x")
    else:
        nbins = min(ny, nx) // 4
        collective = {"op": "all_gather", "backend": env.backend_label,
                      "bytes_per_rank": int(-(-nt_total // world) * nbins * 16), "per_step": 2}

        def step():  # BASELINE.json configs[3]: cross spectrum (stays sharded) + isotropic power spectra of the two fields (gathered)
            cs = xrft.cross_spectrum(da, db, dim=["y", "x"], window="hann")
            ia = xrft.isotropic_power_spectrum(da, dim=["y", "x"], window="hann")
            ib = xrft.isotropic_power_spectrum(db, dim=["y", "x"], window="hann")
            if dist is not None:
                ia = xdist.all_gather_batch(ia, "time", nt_total)
                ib = xdist.all_gather_batch(ib, "time", nt_total)
            return cs, ia, ib

    def barrier():
        env.sync(dev)
        if dist is not None:
            dist.barrier()
        env.sync(dev)

    # setup (not a step): prime torch's caching allocator so that no hipMalloc of a 4 GiB output lands in the timed
    # region -- a step holds the previous result while the next one is produced, i.e. two output blocks are live
    ps = step()
    ps_prev = ps
    ps = step()
    del ps_prev
    barrier()
    # the per-kernel HIP events are switched on BEFORE the warm-up: the first launch that carries timestamps stalls its queue once (0.5 ms -- a
    # quarter of the whole timed region of the C2 workload); the records of the warm-up steps are dropped below
    plan = next(reversed(api._plan_cache.values())) if api._plan_cache else None
    if plan is not None and not args.no_profile:
        plan.set_profiling(True)
    for _ in range(args.warmup):
        ps = step()
    barrier()
    if plan is not None and not args.no_profile:
        plan.set_profiling(True)  # (clears the records)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ps = step()
    barrier()
    dt = time.perf_counter() - t0
    prof = plan.read_profile() if (plan is not None and not args.no_profile) else {}
    if plan is not None:
        plan.set_profiling(False)
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

    points_per_step = float(nt_total) * ny * nx  # points of ONE field transformed by all ranks per step
    value = 1e-9 * points_per_step * args.steps / dt
    ms_per_step = 1e3 * dt / args.steps

    out = None
    if rank == 0:
        # ---- roofline: achieved = algorithmic bytes / wall of the whole hot path; the dominant kernel's own
        # figure (algorithmic bytes of its launch / its average launch duration) is kept beside it as `kernel`
        roof = None
        if prof:
            kern = {k: v for k, v in prof.items()}
            dom = max(kern, key=lambda k: kern[k][1])
            launches, total_ms = kern[dom]
            avg_s = 1e-3 * total_ms / launches
            launches_per_step = launches / args.steps
            # c4: two float32 fields in, one complex64 cross spectrum out per point (16 B/point); the two
            # isotropic calls read the two fields again (4 B/point each, their output is negligible)
            # c2: float32 in, complex64 out (12 B/point); c5: float64 in, float64 out (16 B/point)
            bpp = {"ps": BYTES_PER_POINT, "c4": 16.0 + 8.0, "c2": 12.0, "c5": 16.0}[args.workload]
            pts_per_launch = float(nt) * ny * nx / max(launches_per_step, 1e-9)
            # the HIP events cover the LAST plan of the step: the whole step for ps / c2 / c5, one isotropic_power_spectrum call
            # (one field read, nothing but the radial sums written: 4 B per point) for c4
            bpp_prof = 4.0 if args.workload == "c4" else bpp
            k_achieved = bpp_prof * pts_per_launch / avg_s
            kernel_ms = sum(v[1] for v in kern.values()) / args.steps
            path_achieved = bpp * value * 1e9 / world  # B/s per GPU
            # HBM traffic of one step: rocprofv3 cannot run inside the timed process, so the figure comes from the committed PMC
            # profile of this same command (scripts/gpu_profile_r05.sh, --nt 64) -- and only if that profile was taken on the
            # kernels that just ran: it is stamped with the SHA-1 of xrft_amd/csrc, a mismatch leaves traffic null
            traffic = None
            tnote = None
            ceiling = None
            try:
                tname = "r05_traffic_c2.json" if args.workload == "c2" else "r05_traffic.json"
                with open(os.path.join(REPO, "profiles", tname)) as fh:
                    tj = json.load(fh)
                if args.workload == "c2" and nx == 65536:
                    if tj.get("csrc_sha1") == csrc_sha1():
                        traffic = tj["path_hbm_bytes_per_slab"] * nt
                        tnote = tj.get("note")
                    else:
                        tnote = (f"profiles/{tname} was measured on other kernel sources (csrc SHA-1 "
                                 f"{str(tj.get('csrc_sha1'))[:12]} != {csrc_sha1()[:12]}): re-run scripts/gpu_profile_r05.sh")
                if args.workload == "ps" and (ny, nx) == (4096, 4096):
                    ceiling = tj.get("claimed_floor") or tj.get("two_pass_floor")
                    if ceiling is not None:  # a claim, not a measurement of this run: it travels with the skeleton's source hash
                        import hashlib

                        with open(os.path.join(REPO, "scripts", "ubench", "fused.hip"), "rb") as fh:
                            sha = hashlib.sha1(fh.read()).hexdigest()
                        ceiling = dict(ceiling, claim="builder's claim from a committed skeleton timing, not measured in this run",
                                       ubench_sha1=sha)
                        if tj.get("ubench_sha1") not in (None, sha):
                            ceiling = None  # the skeleton changed since it was timed
                    if tj.get("csrc_sha1") == csrc_sha1():
                        traffic = tj["path_hbm_bytes_per_slab"] * nt
                        tnote = tj.get("note")
                    else:
                        tnote = (f"profiles/{tname} was measured on other kernel sources (csrc SHA-1 "
                                 f"{str(tj.get('csrc_sha1'))[:12]} != {csrc_sha1()[:12]}): re-run scripts/gpu_profile_r05.sh")
            except Exception as e:
                tnote = f"no traffic profile: {e!r}"
            roof = {
                "bound": "hbm", "achieved": round(path_achieved / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(path_achieved / HBM_PEAK, 4),
                "definition": {"ps": "algorithmic bytes (8 B per input point: 4 read + 4 written) of one step / wall time of the step, per GPU",
                               "c4": "algorithmic bytes (cross spectrum 16 B + two isotropic power spectra 2 x 4 B per point of one field) of one "
                                     "step / wall time of the step, per GPU",
                               "c2": "algorithmic bytes (12 B per point: float32 read + complex64 written) of one step / wall time of the step, per GPU",
                               "c5": "algorithmic bytes (16 B per point: float64 read + float64 written) of one step / wall time of the step, per GPU",
                               }[args.workload],
                "traffic": traffic, "traffic_note": tnote,
                "kernel": {"name": dom, "avg_launch_us": round(avg_s * 1e6, 2), "points_per_launch": pts_per_launch,
                           "achieved": round(k_achieved / 1e9, 2), "frac": round(k_achieved / HBM_PEAK, 4),
                           # what THIS kernel itself must move (a pass of a two-pass transform reads or writes the intermediate, too): the
                           # figure to hold against the copy rate; `frac` above prices the whole path's bytes against one pass's time
                           "own": own_share(dom, args.workload, ny, nx, pts_per_launch, avg_s),
                           "definition": "algorithmic bytes OF THE WHOLE PATH for the slabs one launch of the longest kernel processes / its average "
                                         "launch duration (HIP events on the launch stream inside the timed region); `own` = the bytes this "
                                         "kernel alone has to move (input or intermediate read + intermediate or result written)"
                                         + (" -- of one isotropic_power_spectrum call, the plan the events are recorded on: 4 B per point"
                                            if args.workload == "c4" else "")},
                "bytes_per_point": bpp,
                "kernels_ms_per_step": {k: round(v[1] / args.steps, 3) for k, v in kern.items()},
                "sum_kernels_ms_per_step": round(kernel_ms, 3),
                "claimed_floor": ceiling,
            }
        # ---- CPU baseline (the oracle on a bounded sample, 1 thread) + parity of the same slabs
        cpu = None
        parity = None
        if args.cpu_slabs > 0 and world == 1 and args.workload == "ps" and env.measures:
            from oracle import xrft_oracle as oracle

            try:
                from threadpoolctl import threadpool_limits
            except Exception:  # pragma: no cover
                threadpool_limits = None
            ns = min(args.cpu_slabs, nt)
            sub = x[:ns].cpu().numpy()
            oc = {"time": np.arange(ns), "y": coords["y"], "x": coords["x"]}
            import contextlib

            limiter = threadpool_limits(limits=1) if threadpool_limits else contextlib.nullcontext()
            with limiter:
                t0 = time.perf_counter()
                ref = oracle.power_spectrum(oracle.OArr(sub, ("time", "y", "x"), oc), dim=["y", "x"],
                                            detrend="linear", window="hann")
                tc = time.perf_counter() - t0
            cpu = {"value": round(1e-9 * ns * ny * nx / tc, 6), "unit": "GFFT/s", "cores": 1, "kind": "port",
                   "sample": f"{ns} of {nt} slabs ({ny}x{nx} f32) through oracle.power_spectrum(detrend='linear', "
                             f"window='hann') [numpy pocketfft + the reference's plane-fit algorithm], 1 thread, "
                             f"{tc:.1f} s; host has {os.cpu_count()} cores"}
            got = ps.data[:ns].cpu().numpy()
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
                    slabs = x[:npool].cpu().numpy()
                    with mp.get_context("spawn").Pool(npool, initializer=_cpu_pool_init) as pool:
                        pool.map(_cpu_pool_slab, [(None, None)] * npool)  # workers up, numpy/scipy imported
                        t0 = time.perf_counter()
                        pool.map(_cpu_pool_slab, [(slabs[i], (coords["y"], coords["x"])) for i in range(npool)], chunksize=1)
                        tp = time.perf_counter() - t0
                    cpu["all_cores"] = {"value": round(1e-9 * npool * ny * nx / tp, 6), "unit": "GFFT/s", "cores": npool,
                                        "sample": f"{npool} slabs, one per worker process (1 thread each), {tp:.1f} s; workers = "
                                                  f"min(host cores {ncores}, slabs in the workload {nt}, memory cap {mem_cap})"}
                except Exception as e:  # pragma: no cover
                    cpu["all_cores"] = {"error": repr(e)}
        if args.workload == "ps":
            wl = (f"xrft.power_spectrum dim=[y,x] detrend=linear window=hann on ({nt},{ny},{nx}) float32 per GPU "
                  f"(BASELINE.json configs[2])")
            metric = f"2-D power_spectrum GFFT/s (nt,{ny},{nx}) fp32"
            par = f"time-slab shards x{world}, no collective"
        elif args.workload == "c5":
            wl = (f"xrft.power_spectrum dim=[y,x] detrend=linear window=hann on ({nt},{ny},{nx}) float64 per GPU "
                  f"(BASELINE.json configs[4])")
            metric = f"2-D power_spectrum GFFT/s (nt,{ny},{nx}) fp64"
            par = f"time-slab shards x{world}, no collective"
        elif args.workload == "c2":
            wl = f"xrft.dft dim=x on ({nt},{nx}) float32 per GPU (BASELINE.json configs[1])"
            metric = f"1-D dft GFFT/s (nt,{nx}) fp32"
            par = f"row shards x{world}, no collective"
        else:
            wl = (f"xrft.cross_spectrum + xrft.isotropic_power_spectrum (of each field) window=hann on two ({nt},{ny},{nx}) float32 fields "
                  f"per GPU (BASELINE.json configs[3]); GFFT/s counts the points of one field")
            metric = f"2-D cross_spectrum + isotropic_power_spectrum GFFT/s (nt,{ny},{nx}) fp32"
            par = f"time-slab shards x{world}; full cross spectra stay sharded, one all_gather of the ({nt_total}, {min(ny, nx) // 4}) isotropic result per field and step"
        out = {
            "metric": metric, "value": round(value, 3), "unit": "GFFT/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64" if args.workload == "c5" else "f32", "data": env.data_label,
            "config": {"workload": wl, "nt_per_gpu": nt, "nt_total": nt_total, "ny": ny, "nx": nx, "parallelism": par,
                       "collective": collective, "shard_sizes": shard_sizes, "slabs_per_s": round(nt_total * args.steps / dt, 2),
                       "ranks_in_process_group": ranks_reported, "process_group_backend": env.backend if dist is not None else None},
            "roofline": roof, "cpu_baseline": cpu, "parity_max_rel_err_vs_oracle": parity,
        }
        if plan is not None:
            out["plan"] = plan.describe().strip().split("\n")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    main()
