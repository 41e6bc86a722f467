"""Multi-process (world_size = 2, gloo) test of the leading-axis sharding in xrft_amd.dist: every rank transforms its own
contiguous block of time slabs (no data-path collective), the isotropic results are all-gathered / all-reduced.
The ranks compute through the emulated C-ABI library (CPU), the reference is the oracle on the full cube."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import warnings

    warnings.simplefilter("ignore")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import build_emu
    from xrft_amd import _lib

    _lib._load_for_testing(build_emu.build())
    import xrft_amd as xa
    from xrft_amd import dist as xd
    from oracle import xrft_oracle as o

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(99)  # same cube on every rank; each takes its shard
        nt, ny, nx = 5, 16, 32
        v = rng.standard_normal((nt, ny, nx)) + 0.05 * np.arange(nx)
        c = {"time": np.arange(nt), "y": np.arange(ny) * 0.5, "x": np.arange(nx) * 0.25}
        full = xa.DataArray(v, ("time", "y", "x"), c)
        local = xd.shard(full, "time")
        lo, hi = xd.shard_bounds(nt, rank, world)
        assert local.shape[0] == hi - lo and (lo, hi) == ((0, 3) if rank == 0 else (3, 5))
        ref_full = o.OArr(v, ("time", "y", "x"), c)
        # full spectra stay sharded: the local block equals the corresponding block of the full transform
        ps_local = xa.power_spectrum(local, dim=["y", "x"], detrend="linear", window="hann")
        ps_ref = o.power_spectrum(ref_full, dim=["y", "x"], detrend="linear", window="hann")
        assert np.abs(ps_local.values - ps_ref.values[lo:hi]).max() / np.abs(ps_ref.values).max() < 1e-10
        # isotropic spectrum: local reduce, then ONE small collective
        iso_local = xa.isotropic_power_spectrum(local, dim=["y", "x"], detrend="constant", window="hann")
        iso_ref = o.isotropic_power_spectrum(ref_full, dim=["y", "x"], detrend="constant", window="hann")
        gathered = xd.all_gather_batch(iso_local, "time", nt)
        assert gathered.shape == iso_ref.shape
        assert np.abs(gathered.values - iso_ref.values).max() / np.abs(iso_ref.values).max() < 1e-10
        mean = xd.batch_mean_allreduce(iso_local, "time", nt)
        assert np.abs(mean.values - iso_ref.values.mean(axis=0)).max() / np.abs(iso_ref.values).max() < 1e-10
        # complex (cross) results gather too
        v2 = rng.standard_normal((nt, ny, nx))
        full2 = xa.DataArray(v2, ("time", "y", "x"), c)
        ics_local = xa.isotropic_cross_spectrum(local, xd.shard(full2, "time"), dim=["y", "x"], window="hann")
        ics_ref = o.isotropic_cross_spectrum(ref_full, o.OArr(v2, ("time", "y", "x"), c), dim=["y", "x"], window="hann")
        g2 = xd.all_gather_batch(ics_local, "time", nt)
        assert np.abs(g2.values - ics_ref.values).max() / np.abs(ics_ref.values).max() < 1e-10
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        out.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_shard_bounds():
    sys.path.insert(0, REPO)
    from xrft_amd.dist import shard_bounds

    assert [shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [shard_bounds(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert shard_bounds(64, 7, 8) == (56, 64)


def test_two_rank_gloo_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
