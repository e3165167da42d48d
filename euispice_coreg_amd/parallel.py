"""
Multi-GPU lag sharding: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm,
"gloo" on CPU for tests).  The C-order raveled lag range is cut in `world` contiguous slices -- the same
partition `np.array_split`-style fan-out the reference applies to its worker processes
(hdrshift/alignment.py:677-687) -- every rank sweeps its slice on its own GPU with full replicas of both images,
and ONE all-gather of the per-lag coefficients (a few KB) assembles the correlation map on every rank.
There is no other data-path collective.

Sweeps with very few lag-points per GPU (less than half a 256-lag batch each, where lag sharding would leave most
lanes idle) shard the GRID instead (SURVEY 8e fallback):
every rank sweeps all the lag-points over its share of the target grid's points and ONE all-reduce(SUM) of the six
Pearson sums per lag slot replaces the all-gather (`point_sharded_sweep`).
"""
from __future__ import annotations

import numpy as np


def world_info(group=None):
    """(rank, world_size) of the default / given process group, (0, 1) when torch.distributed is not in use."""
    try:
        import torch.distributed as dist
    except ImportError:
        return 0, 1
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def shard_bounds(n_lags: int, world: int, rank: int):
    """Contiguous, equal-size (last one ragged) slices: chunk = ceil(n / world)."""
    chunk = (n_lags + world - 1) // world
    lo = min(rank * chunk, n_lags)
    hi = min((rank + 1) * chunk, n_lags)
    return lo, hi, chunk


def block_grid(n1: int, n2: int, world: int):
    """Factor `world` = g1 x g2 so that the (CRVAL1, CRVAL2) lag plane n1 x n2 is cut in g1 x g2 blocks as square as
    possible (compact lag patches per GPU = small LDS windows, no padded lanes).  Returns (g1, g2)."""
    best, best_cost = (world, 1), None
    for g1 in range(1, world + 1):
        if world % g1:
            continue
        g2 = world // g1
        b1, b2 = -(-n1 // g1), -(-n2 // g2)  # ceil
        if g1 > n1 or g2 > n2:
            continue
        cost = (abs(b1 - b2), b1 * b2)
        if best_cost is None or cost < best_cost:
            best, best_cost = (g1, g2), cost
    return best


def block_bounds(n1: int, n2: int, world: int, rank: int):
    """Rank's block of the lag plane: (lo1, hi1, lo2, hi2); blocks are ceil-sized, trailing ones may be smaller/empty."""
    g1, g2 = block_grid(n1, n2, world)
    b1, b2 = -(-n1 // g1), -(-n2 // g2)
    r1, r2 = rank // g2, rank % g2
    return min(r1 * b1, n1), min((r1 + 1) * b1, n1), min(r2 * b2, n2), min((r2 + 1) * b2, n2)


def block_gather_index(shape5, world: int):
    """Index array `perm` (int64, prod(shape5) long) with full.ravel() = gathered[perm], where `gathered` is the
    concatenation over ranks of each rank's C-order block [hi1-lo1, hi2-lo2, n3, n4, n5] padded to `chunk` values.
    Returns (perm, chunk)."""
    n1, n2, n3, n4, n5 = shape5
    inner = n3 * n4 * n5
    g1, g2 = block_grid(n1, n2, world)
    chunk = (-(-n1 // g1)) * (-(-n2 // g2)) * inner
    perm = np.empty((n1, n2, inner), dtype=np.int64)
    for r in range(world):
        lo1, hi1, lo2, hi2 = block_bounds(n1, n2, world, r)
        if hi1 <= lo1 or hi2 <= lo2:
            continue
        w2 = hi2 - lo2
        i1 = np.arange(lo1, hi1)[:, None, None]
        i2 = np.arange(lo2, hi2)[None, :, None]
        k = np.arange(inner)[None, None, :]
        perm[lo1:hi1, lo2:hi2, :] = r * chunk + ((i1 - lo1) * w2 + (i2 - lo2)) * inner + k
    return perm.reshape(-1), chunk


def allgather_lag_slices(local, n_lags: int, group=None):
    """Concatenate every rank's slice (rank r holds lags [r*chunk, min((r+1)*chunk, n))) into the full raveled map.

    `local`: this rank's values -- a float64 torch tensor already padded to `chunk` elements (device tensor for
    nccl), or a numpy array of the unpadded slice (staged through the backend's device).  Returns a torch tensor of
    n_lags float64 values on the same device as the collective ran on."""
    import torch
    import torch.distributed as dist
    rank, world = world_info(group)
    lo, hi, chunk = shard_bounds(n_lags, world, rank)
    backend = dist.get_backend(group) if world > 1 else None
    if isinstance(local, np.ndarray):
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        buf = torch.full((chunk,), float("nan"), dtype=torch.float64, device=dev)
        if hi > lo:
            buf[:hi - lo] = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64)).to(dev)
        local = buf
    if world == 1:
        return local[:n_lags]
    if local.numel() != chunk:
        raise ValueError(f"local slice must be padded to chunk={chunk} elements")
    out = torch.empty((chunk * world,), dtype=torch.float64, device=local.device)
    try:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local.contiguous(), group=group)
        out = torch.cat(parts)
    # rank r's valid part is [r*chunk, r*chunk + len_r): with chunk = ceil(n/world) the concatenation of the valid
    # parts is simply the first n_lags elements when only the LAST non-empty slice is ragged -- which is the case here
    return out[:n_lags]


# Below half a 256-lag batch per GPU the grid is sharded instead of the lags.  (Measured, profiles/r02_slice_timing.log:
# on the headline sweep, 450 lag-points per GPU at N = 8, lag blocks are FASTER than point shares -- 0.51 against
# 0.59 ms per step before the collective -- because two rounds of 512 smaller workgroups balance better than one
# round of 240 large ones; point shares pay when a rank would otherwise hold a fraction of one batch.)
POINT_SHARD_MAX_LAGS_PER_RANK = 128


def use_point_sharding(n_lags: int, world: int) -> bool:
    return world > 1 and n_lags < POINT_SHARD_MAX_LAGS_PER_RANK * world


def point_sharded_sweep(handle, run_sweep, n_out, group=None, out_dev_ptr=None):
    """Grid-sharded sweep: `run_sweep()` must call handle.sweep_*(...) over the WHOLE lag range wanted; this rank's
    share of the grid is selected here, the six sums per lag slot are all-reduced (the one collective of this mode) and
    finalised.  Returns the coefficients (numpy, C-order lag slice) or fills `out_dev_ptr`."""
    import torch
    import torch.distributed as dist
    rank, world = world_info(group)
    handle.set_point_shard(rank, world)
    try:
        run_sweep()
        if world == 1:
            return handle.finalize_sums(handle.copy_sums(), n_out, out_dev_ptr)
        backend = dist.get_backend(group)
        n = handle.sums_size()
        if backend == "nccl":
            buf = torch.empty(n, dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
            handle.copy_sums(buf.data_ptr())  # stream-ordered on the handle's stream
            handle.synchronize()
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            torch.cuda.current_stream().synchronize()
            return handle.finalize_sums(buf.data_ptr(), n_out, out_dev_ptr)
        buf = torch.from_numpy(handle.copy_sums())
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        return handle.finalize_sums(buf.numpy(), n_out, out_dev_ptr)
    finally:
        handle.set_point_shard(0, 1)
