"""Multi-GPU use of the spectral path: one process per GPU, leading-axis (e.g. time) slabs sharded over ranks.

Every slab is independent through detrend, window, FFT, |F|^2 / cross and its own radial reduce, so
the full spectra stay sharded on their GPUs and need no collective.  The only exchange the path has is the small
isotropic result -- (nt/world, nbins) per rank -- which is all-gathered (or all-reduced for a batch mean) over
torch.distributed (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .labeled import Coordinate, DataArray

__all__ = ["shard_bounds", "shard", "all_gather_batch", "batch_mean_allreduce"]


def shard_bounds(n, rank, world):
    """Contiguous block [lo, hi) of ``n`` slabs owned by ``rank`` (first n % world ranks get one extra)."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(da, dim, rank=None, world=None):
    """The rank's contiguous block of ``da`` along ``dim`` (never a transform dimension: slabs are not split)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(da.sizes[dim], rank, world)
    return da.isel(**{dim: slice(lo, hi)})


def all_gather_batch(local, dim, total, group=None):
    """All-gather per-rank results along ``dim`` (sizes may differ by one) -> the global array on every rank."""
    world = dist.get_world_size(group)
    ax = local.get_axis_num(dim)
    t = local.data if isinstance(local.data, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(local.data))
    backend = dist.get_backend(group)
    if backend == "nccl" and not t.is_cuda:
        t = t.cuda()
    t = t.movedim(ax, 0).contiguous()
    nmax = -(-int(total) // world)
    pad = torch.zeros((nmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    if pad.is_complex():
        buf = torch.view_as_real(pad).contiguous()
    else:
        buf = pad
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    parts = []
    for r, o in enumerate(outs):
        lo, hi = shard_bounds(total, r, world)
        o = torch.view_as_complex(o) if pad.is_complex() else o
        parts.append(o[: hi - lo])
    full = torch.cat(parts, 0).movedim(0, ax)
    coords = {k: c for k, c in local.coords.items() if dim not in c.dims}
    return DataArray(full, local.dims, coords, local.name, local.attrs)


def batch_mean_allreduce(local, dim, total, group=None):
    """Mean over the sharded batch dimension: every rank's slabs are summed and scaled by 1 / total in ONE library kernel
    (xrfthip_reduce_axis: float64 accumulation in slab order, bit-reproducible per rank), then one all_reduce(SUM) of the nbins
    values per rank finishes the mean -- no torch arithmetic on the data path (test_xrft.py:1011-1013)."""
    from . import engine

    ax = local.get_axis_num(dim)
    t = local.data if isinstance(local.data, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(local.data))
    if dist.get_backend(group) == "nccl" and not t.is_cuda:
        t = t.cuda()
    if (t.is_floating_point() or t.is_complex()) and t.device.type == engine.device_type() and t.shape[ax] > 0:
        s = engine.reduce_axis(t, ax, 1.0 / float(total))
    else:
        # host-resident data under a CPU backend (they stay where the process group can reduce them), integer / bool data, an empty
        # shard: the local term on the host, same order and accumulation type (float64, slab order) as the library kernel.
        # Never for floating-point data on a GPU backend: there the library kernel above is the only path (no torch arithmetic
        # on the product's data path)
        if dist.get_backend(group) == "nccl" and (t.is_floating_point() or t.is_complex()) and t.shape[ax] > 0:
            raise RuntimeError("xrft_amd.dist.batch_mean_allreduce: floating-point data under the nccl backend must be reduced by the HIP library "
                               f"(tensor on {t.device}, library device type {engine.device_type()!r})")
        acc = torch.complex128 if t.is_complex() else torch.float64
        s = t.to(acc).cumsum(ax).select(ax, -1) if t.shape[ax] > 0 else torch.zeros(t.shape[:ax] + t.shape[ax + 1:], dtype=acc, device=t.device)
        s = (s * (1.0 / float(total))).to(t.dtype if (t.is_floating_point() or t.is_complex()) else torch.float64).contiguous()
    buf = torch.view_as_real(s) if s.is_complex() else s
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    dims = [d for d in local.dims if d != dim]
    coords = {k: c for k, c in local.coords.items() if dim not in c.dims}
    return DataArray(s, dims, coords, local.name, local.attrs)
