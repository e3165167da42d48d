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
    """(rank, world_size) of the default / given process group, (0, 1) when torch.distributed is not in use.
    A process group exists only if the caller imported torch: a plain script that never did is not made to pay for
    that import (0.8 s of the first call, profiles/first_call.py)."""
    import sys
    if "torch" not in sys.modules:
        return 0, 1
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


# Cost model of the partition planner, in units of the sweep time of ONE batch of 256 lags (a workgroup's lag patch):
# every (cdelt1, cdelt2, crota) combination is one precompute + one sweep launch; its fixed cost -- precompute, tile
# list, ramp-up and drain of the launch -- was measured at 0.13-0.15 ms against 0.2 ms per batch on the headline
# (DESIGN.md section 6), the same ratio on the 4096^2 grid of cfg5.
LAUNCH_OVERHEAD_BATCHES = 0.75


def lag_batches(b1: int, b2: int) -> int:
    """Lag batches (workgroup patches of sw x sh <= 256 lags) the library needs for a b1 x b2 block of the CRVAL plane:
    the minimum over patch shapes, as csrc/host_plan.hpp:choose_plan searches it."""
    if b1 < 1 or b2 < 1:
        return 0
    best = None
    for sw in range(1, min(b1, 256) + 1):
        sh = min(b2, 256 // sw)
        n = (-(-b1 // sw)) * (-(-b2 // sh))
        best = n if best is None else min(best, n)
    return best


def combo_bounds(inner: int, g_combo: int, k: int):
    """Contiguous, balanced run of inner combinations for combo-rank k of g_combo (np.array_split's sizes)."""
    q, r = divmod(inner, g_combo)
    lo = k * q + min(k, r)
    return lo, lo + q + (1 if k < r else 0)


def lag_plan(shape5, world: int, per_combo_launch: bool = True):
    """How a sweep over the lag set `shape5` = (n_crval1, n_crval2, n_cdelt1, n_cdelt2, n_crota) is spread over `world`
    GPUs: (mode, g_combo, g1, g2).  The GPUs form a g_combo x (g1 x g2) grid: rank r = kc * (g1 * g2) + kb sweeps the
    inner (cdelt1, cdelt2, crota) combinations `combo_bounds(inner, g_combo, kc)` over block kb of the (CRVAL1, CRVAL2)
    plane.
      'none'    one GPU;
      'points'  every rank sweeps all lag-points over its share of the grid (few lag-points per GPU, one all-reduce);
      'blocks'  g_combo = 1: the plane cut in `world` near-square blocks (compact lag patches = small LDS windows);
      'combos'  g_combo > 1: the combinations are dealt to the GPUs, each sweeps the whole plane (or a block of it) at
                one-GPU efficiency and runs 1 / g_combo of the precomputes -- what SURVEY 8(e) sketched for 3-D / 5-D
                sweeps;
      'slices'  contiguous slices of the raveled C-order index, the literal np.array_split fan-out of the reference
                (alignment.py:677-687), when no grid gives every rank work (e.g. 3 x 3 CRVAL lags on 8 GPUs).
    Among the grids that give every rank a non-empty share the planner takes the cheapest by the model above (cost of
    the busiest rank: launches x fixed cost + lag batches); ties go to the larger g_combo.  `per_combo_launch`: the
    Carrington and plate-carree sweeps run one precompute + one launch per combination; the helioprojective sweep
    carries the combination in each lane's homography and runs ONE launch whatever the lag set (False)."""
    n1, n2 = int(shape5[0]), int(shape5[1])
    inner = int(shape5[2]) * int(shape5[3]) * int(shape5[4])
    n = n1 * n2 * inner
    if world <= 1:
        return "none", 1, 1, 1
    if use_point_sharding(n, world):
        return "points", 1, 1, 1
    best = None
    for gc in range(1, world + 1):
        if world % gc or gc > inner:
            continue
        gb = world // gc
        if not all(b[1] > b[0] and b[3] > b[2] for b in (block_bounds(n1, n2, gb, r) for r in range(gb))):
            continue
        g1, g2 = block_grid(n1, n2, gb)
        n_c, n_b = -(-inner // gc), lag_batches(-(-n1 // g1), -(-n2 // g2))
        cost = n_c * (LAUNCH_OVERHEAD_BATCHES + n_b) if per_combo_launch else LAUNCH_OVERHEAD_BATCHES + n_c * n_b
        if best is None or cost < best[0] - 1e-9 or (abs(cost - best[0]) <= 1e-9 and gc > best[1]):
            best = (cost, gc, g1, g2)
    if best is None:
        return "slices", 1, 1, 1
    return ("combos" if best[1] > 1 else "blocks"), best[1], best[2], best[3]


def lag_sharding(shape5, world: int, per_combo_launch: bool = True) -> str:
    """The mode of `lag_plan` alone ('none', 'points', 'blocks', 'combos', 'slices')."""
    return lag_plan(shape5, world, per_combo_launch)[0]


def grid_share(shape5, world: int, rank: int, per_combo_launch: bool = True):
    """This rank's share under a 'blocks' / 'combos' plan: (lo1, hi1, lo2, hi2, c_lo, c_hi)."""
    _, gc, g1, g2 = lag_plan(shape5, world, per_combo_launch)
    gb = g1 * g2
    kc, kb = rank // gb, rank % gb
    inner = int(shape5[2]) * int(shape5[3]) * int(shape5[4])
    c_lo, c_hi = combo_bounds(inner, gc, kc)
    return block_bounds(int(shape5[0]), int(shape5[1]), gb, kb) + (c_lo, c_hi)


def grid_gather_index(shape5, world: int, per_combo_launch: bool = True):
    """Index array `perm` (int64, prod(shape5) long) with full.ravel() = gathered[perm], where `gathered` is the
    concatenation over ranks of each rank's C-order share [hi1-lo1, hi2-lo2, c_hi-c_lo] padded to `chunk` values.
    Returns (perm, chunk).  With g_combo = 1 this is `block_gather_index`."""
    n1, n2, n3, n4, n5 = (int(v) for v in shape5)
    inner = n3 * n4 * n5
    _, gc, g1, g2 = lag_plan(shape5, world, per_combo_launch)
    chunk = (-(-n1 // g1)) * (-(-n2 // g2)) * (-(-inner // gc))
    perm = np.empty((n1, n2, inner), dtype=np.int64)
    for r in range(world):
        lo1, hi1, lo2, hi2, c_lo, c_hi = grid_share(shape5, world, r, per_combo_launch)
        if hi1 <= lo1 or hi2 <= lo2 or c_hi <= c_lo:
            continue
        w2, nc = hi2 - lo2, c_hi - c_lo
        i1 = np.arange(lo1, hi1)[:, None, None]
        i2 = np.arange(lo2, hi2)[None, :, None]
        k = np.arange(nc)[None, None, :]
        perm[lo1:hi1, lo2:hi2, c_lo:c_hi] = r * chunk + ((i1 - lo1) * w2 + (i2 - lo2)) * nc + k
    return perm.reshape(-1), chunk


def allgather_lag_blocks(local, shape5, group=None, per_combo_launch: bool = True):
    """Every rank holds the coefficients of its share of the lag set -- a block of the (CRVAL1, CRVAL2) plane times a run
    of inner combinations (`grid_share`; C order [hi1-lo1, hi2-lo2, c_hi-c_lo], numpy): ONE all-gather of ceil-sized chunks
    + one index permutation give the full raveled C-order map (numpy) on every rank."""
    import torch
    import torch.distributed as dist
    rank, world = world_info(group)
    perm, chunk = grid_gather_index(tuple(int(v) for v in shape5), world, per_combo_launch)
    backend = dist.get_backend(group) if world > 1 else None
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    buf = torch.full((chunk,), float("nan"), dtype=torch.float64, device=dev)
    local = np.ascontiguousarray(local, dtype=np.float64).ravel()
    if local.size > chunk:
        raise ValueError(f"a block holds at most chunk={chunk} lag-points, got {local.size}")
    if local.size:
        buf[:local.size] = torch.from_numpy(local).to(dev)
    if world == 1:
        return buf.cpu().numpy()[perm]
    out = torch.empty((chunk * world,), dtype=torch.float64, device=dev)
    try:
        dist.all_gather_into_tensor(out, buf, group=group)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
        out = torch.cat(parts)
    return out.cpu().numpy()[perm]


def replicate_image(img, group=None):
    """The multi-GPU hand-over of one host image: rank r sends rows [r * ceil(H / N), ...) over ITS OWN PCIe link and one
    RCCL all-gather over xGMI assembles the full replica on every GPU -- 1/N of the bytes per link instead of N uploads
    of everything through one host's memory (the reference hands its workers the images through shared memory,
    alignment.py:692-720).  `img`: the same 2-D float32 / float64 numpy array on every rank.  Returns a device tensor
    [H, W] (keep it alive while the library reads it), or None when the process group is not RCCL (gloo rehearsals,
    one rank): the caller then uploads the whole image itself."""
    import torch
    import torch.distributed as dist
    rank, world = world_info(group)
    if world <= 1 or dist.get_backend(group) != "nccl":
        return None
    img = np.ascontiguousarray(img)
    if img.ndim != 2 or img.dtype not in (np.float32, np.float64):
        raise ValueError("replicate_image: 2-D float32 / float64 array expected")
    H, W = img.shape
    rows = -(-H // world)
    dev = torch.device("cuda", torch.cuda.current_device())
    tdt = torch.float32 if img.dtype == np.float32 else torch.float64
    full = torch.empty((rows * world, W), dtype=tdt, device=dev)
    lo, hi = min(rank * rows, H), min((rank + 1) * rows, H)
    mine = full[rank * rows:(rank + 1) * rows]  # all_gather_into_tensor accepts the in-place form
    if hi > lo:
        mine[:hi - lo].copy_(torch.from_numpy(img[lo:hi]), non_blocking=False)
    if hi - lo < rows:
        mine[hi - lo:].zero_()
    try:
        dist.all_gather_into_tensor(full, mine.clone(), group=group)
    except (RuntimeError, NotImplementedError):
        return None  # (raised on every rank alike: the callers then upload the whole image themselves)
    return full[:H]


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
    if world == 1:
        return run_sweep()  # nothing to shard: the ordinary sweep, its own result
    backend = dist.get_backend(group)
    # the ranks' sums only add up when all of them subtracted the same two pivots: rank 0's are used everywhere
    piv = torch.tensor(handle.get_pivots(), dtype=torch.float64)
    mine = piv.clone()
    if backend == "nccl":
        piv = piv.to(torch.device("cuda", torch.cuda.current_device()))
    dist.broadcast(piv, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    piv = piv.cpu()
    if not torch.equal(piv, mine):
        handle.set_pivots(float(piv[0]), float(piv[1]))
    handle.set_point_shard(rank, world)
    try:
        run_sweep()
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
