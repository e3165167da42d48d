#!/usr/bin/env python3
"""
bench.py -- lag-points/sec of the alignment correlation sweep on MI355X.

Workload (BASELINE.json metric / SURVEY.md 8d "headline"): Carrington 'fa' sweep, 2048x2048 lon/lat grid,
lon (200, 300) deg, lat (-20, 20) deg, lag_crval1 = lag_crval2 = arange(-30, 30, 1) arcsec (60 x 60 = 3600
lag-points), crota/cdelt fixed, solar_r 1.004, order 2, method 'correlation'; seeded synthetic HRIEUV-like
2048^2 image to align and FSI-like 3072^2 reference (euispice_coreg_amd/synthetic.py).

A "step" = one full sweep (all 3600 lag-points) through the C ABI with both images already resident in HBM
(upload + once-only reference preparation happen before the timed region).  `--streams` (default 2) steps are in
flight at a time, each on its own HIP stream and library context, so that the drain of one sweep overlaps the start of
the next; every step is a complete sweep.  N > 1 (torchrun, one rank per GPU):
the (CRVAL1, CRVAL2) lag plane is cut in N blocks (the multi-GPU form of the reference's np.array_split fan-out,
alignment.py:677-687), every rank sweeps its block with full image replicas and ONE all-gather (RCCL) of the per-lag
coefficients, followed by an index permutation, assembles the map on every rank.  Total work is fixed as N grows
-> "scaling": "strong".

Prints ONE JSON line on rank 0 (see the driver contract) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GRID_SHAPE = (2048, 2048)
LONLIMS = (200.0, 300.0)
LATLIMS = (-20.0, 20.0)
SOLAR_R = 1.004
ORDER = 2
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TF = 78.6  # MI355X vector FP64 spec
FLOP_PER_POINT_LAG = 70.0  # ~70 fp64 flop per (grid point, lag)


METRIC = "lag-points/sec (whole node) on 2048\u00b2 grid, 60\u00d760 CRVAL sweep; argmax-shift match"
try:
    with open(os.path.join(ROOT, "BASELINE.json")) as _f:
        METRIC = json.load(_f).get("metric", METRIC)
except Exception:
    pass


def pmc_traffic_bytes():
    """HBM bytes per k_sweep launch from the committed rocprofv3 PMC summary of this same command
    (profiles/r01_pmc_summary.txt; FETCH_SIZE/WRITE_SIZE are KiB per dispatch, FETCH_SIZE doubled: gfx950 counts
    128-B read requests as 64 B -- MI355X_MICROARCH.md, HBM section).  None when no summary is committed."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.txt")
    try:
        vals = {}
        for line in open(path):
            parts = line.split()
            if len(parts) >= 4 and parts[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[parts[0]] = float(parts[3])
        return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    except Exception:
        return None


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(small, hs, large, hl, lags, n_sample, cores):
    """The oracle's restatement of the reference's parallelism=True path (process fan-out over np.array_split
    chunks, images in shared memory), timed on a seeded random subsample of the same 3600 lag-points."""
    from oracle import coreg_oracle as O
    from tests import helpers as H
    st = H.oracle_state(small, hs, large, hl, lags, order=ORDER, shape=list(GRID_SHAPE), lonlims=list(LONLIMS),
                        latlims=list(LATLIMS), solar_r=(SOLAR_R,))
    n_total = len(lags[0]) * len(lags[1])
    rng = np.random.default_rng(1234)
    subset = np.sort(rng.choice(n_total, size=min(n_sample, n_total), replace=False))
    # the once-only reference preparation is outside the timed region on both sides
    O.set_initial_header_values(st)
    ref = O.prepare_reference(st, "carrington", SOLAR_R)

    t0 = time.perf_counter()
    corr = O.find_best_header_parameters(st, "carrington", counts=cores, lag_subset=subset, prepared_reference=ref)
    dt = time.perf_counter() - t0
    return {"value": len(subset) / dt, "unit": "lag-points/s", "cores": int(cores), "kind": "port",
            "sample": f"{len(subset)} seeded random lag-points of the same 60x60 sweep, {cores} worker processes, "
                      f"{dt:.1f} s wall"}, corr, subset


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=2, help="sweeps in flight (one HIP stream + library context each)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive measurement after the timed region")
    ap.add_argument("--cpu-sample", type=int, default=0, help="lag-points in the CPU sample (0 = 48 per core)")
    ap.add_argument("--use-lds", type=int, default=1)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    # native libraries (RCCL prints a version banner on first use) write to fd 1: keep stdout for the ONE JSON line
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from euispice_coreg_amd import _lib, synthetic

    # COREG_BENCH_BACKEND=gloo: rehearse the N > 1 path on a box with fewer GPUs than ranks (ranks share devices,
    # the all-gather runs on the CPU); the driver's runs use the default, nccl (= RCCL over xGMI)
    backend = os.environ.get("COREG_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    # COREG_BENCH_FORCE_DIST=1: initialise the process group (and run the all-gather) even with one rank, to exercise
    # the RCCL code path on a one-GPU box
    use_dist = world > 1 or os.environ.get("COREG_BENCH_FORCE_DIST", "0") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    t0 = time.time()
    small, hs, large, hl, truth = synthetic.make_scene()
    lag1 = np.arange(-30, 30, 1, dtype=np.float64)
    lag2 = np.arange(-30, 30, 1, dtype=np.float64)
    lags = (lag1, lag2, None, None, None)
    L = lag1.size * lag2.size
    if rank == 0:
        log(f"[bench] scene built in {time.time() - t0:.1f} s; L = {L}")

    # `--streams S` (default 2) sweeps are in flight at a time, each on its own HIP stream with its own library context:
    # the tail of one sweep (workgroups draining, the small finalize / next precompute kernels) overlaps the head of
    # the next.  Every step is still one complete sweep of all lag-points of this rank.
    n_streams = max(1, args.streams)
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(n_streams - 1)]
    grid = _lib.Grid(LONLIMS, LATLIMS, GRID_SHAPE, numpy_lat_trig=True)
    lagset = _lib.LagSet(*lags)
    small_m = small.copy()
    handles = []
    for s in streams:
        hk = _lib.CoregHandle(local_rank)
        hk.set_option("use_lds", args.use_lds)
        for kv in filter(None, os.environ.get("COREG_BENCH_OPTS", "").split(",")):  # tuning: "opt=val,opt=val"
            k, v = kv.split("=")
            hk.set_option(k, int(v))
        hk.set_stream(s.cuda_stream)
        # inputs resident in HBM before the timed region
        hk.set_small(small_m)
        hk.prepare_reference_carrington(large, hl, grid, SOLAR_R, ORDER)
        handles.append(hk)
    h = handles[0]

    from euispice_coreg_amd import parallel
    lo1, hi1, lo2, hi2 = parallel.block_bounds(lag1.size, lag2.size, world, rank)
    my_lags = _lib.LagSet(lag1[lo1:hi1], lag2[lo2:hi2], None, None, None) if (hi1 > lo1 and hi2 > lo2) else None
    perm_np, chunk = parallel.block_gather_index((lag1.size, lag2.size, 1, 1, 1), world)
    mine = [torch.full((chunk,), float("nan"), dtype=torch.float64, device="cuda") for _ in streams]
    gathered = [torch.empty((chunk * world,), dtype=torch.float64, device="cuda") if use_dist else m for m in mine]
    perm = torch.from_numpy(perm_np).to("cuda")
    torch.cuda.synchronize()
    result = [None]
    step_no = [0]

    def step():
        k = step_no[0] % n_streams
        step_no[0] += 1
        with torch.cuda.stream(streams[k]):
            if my_lags is not None:
                handles[k].sweep_carrington(hs, grid, SOLAR_R, my_lags, order=ORDER, out_dev_ptr=mine[k].data_ptr())
            if use_dist:
                if backend == "nccl":
                    dist.all_gather_into_tensor(gathered[k], mine[k])  # the ONE collective of the path
                    result[0] = gathered[k][perm]
                else:
                    parts = [torch.empty(chunk, dtype=torch.float64) for _ in range(world)]
                    dist.all_gather(parts, mine[k].cpu())
                    result[0] = torch.cat(parts)[perm.cpu()]
            else:
                result[0] = mine[k][perm]

    for _ in range(args.warmup):
        step()
    kernel_ms, pre_ms = [], []
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()  # asynchronous: the host plans step k+1 while the GPU runs step k; nothing is read back in here
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # dominant-kernel duration: the library brackets every k_sweep launch with HIP events on the launch stream.  Reading
    # them waits for the sweep, and with several sweeps in flight a kernel's event interval also contains the time its
    # workgroups waited for the other sweep's to leave the CUs -- so the kernel is timed here, right after the timed
    # region, on identical steps issued ONE at a time
    stats = {"sweep_kernel_ms": 0.0, "precompute_ms": 0.0, "n_active_points": 0, "small_is_f32": 1, "used_lds": 1}
    for _ in range(min(8, args.steps)):
        step_no[0] = 0  # handle / stream 0
        step()
        if my_lags is not None:
            stats = h.last_stats()
            kernel_ms.append(stats["sweep_kernel_ms"])
            pre_ms.append(stats["precompute_ms"])
        else:
            torch.cuda.synchronize()
    if not kernel_ms:
        kernel_ms, pre_ms = [0.0], [0.0]
    if use_dist:
        dist.barrier()
    corr = result[0].cpu().numpy()
    corr = corr.reshape(lag1.size, lag2.size)

    # the boundary hands over host buffers: the same step with both images uploaded (and the reference re-prepared) and
    # the map copied back to the host every call -- reported beside `value`, never as `value`
    pcie = None
    if world == 1 and not args.no_pcie:
        full_lags = _lib.LagSet(*lags)
        times = []
        for _ in range(4):
            t0 = time.perf_counter()
            h.set_small(small_m)
            h.prepare_reference_carrington(large, hl, grid, SOLAR_R, ORDER)
            h.sweep_carrington(hs, grid, SOLAR_R, full_lags, order=ORDER)
            times.append(time.perf_counter() - t0)
        best = min(times[1:])
        pcie = {"value": L / best, "unit": "lag-points/s", "ms_per_step": 1e3 * best,
                "what": "host float64 images in (2048^2 image to align + 3072^2 reference, 104 MiB, reference "
                        "re-prepared), host correlation map out, every call"}

    if rank == 0:
        value = L * args.steps / elapsed
        ms_per_step = 1e3 * elapsed / args.steps
        G = GRID_SHAPE[0] * GRID_SHAPE[1]
        S = small.shape[0] * small.shape[1]
        b_lag = G * 8 + S * 8  # reference grid value + small-image pixel, fp64, touched once per lag
        k_ms = float(np.mean(kernel_ms))
        lags_per_launch = (hi1 - lo1) * (hi2 - lo2)
        achieved = b_lag * lags_per_launch / (k_ms * 1e-3) / 1e9
        act = stats["n_active_points"]
        valu_tf = act * lags_per_launch * FLOP_PER_POINT_LAG / (k_ms * 1e-3) / 1e12
        am = np.unravel_index(np.nanargmax(corr), corr.shape)
        out = {
            "metric": METRIC,
            "value": value, "unit": "lag-points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "headline: Carrington 'fa' 2048x2048 grid lon(200,300) lat(-20,20), 60x60 CRVAL "
                                   "lags arange(-30,30,1) arcsec, small 2048^2 HRIEUV-like, ref 3072^2 FSI-like, "
                                   "order 2, solar_r 1.004",
                       "lag_points": L, "grid": list(GRID_SHAPE), "parallelism": f"lag-plane blocks x{world} + 1 all-gather",
                       "sweeps_in_flight": n_streams,
                       "small_stored_f32": bool(stats["small_is_f32"]), "use_lds": bool(stats["used_lds"])},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes(),
                         "traffic_unit": "bytes per launch (rocprofv3 PMC, profiles/r01_pmc_summary.txt)",
                         "algorithmic_bytes_per_launch": b_lag * lags_per_launch,
                         "kernel": "k_sweep<TRANSLATE,2,f32>", "kernel_ms": k_ms,
                         "timed_region_ms_per_launch": ms_per_step,  # wall per sweep with `sweeps_in_flight` overlapped
                         "algorithmic_bytes_per_lag": b_lag, "lags_per_launch": lags_per_launch,
                         "note": "algorithmic bytes = one-lag-per-pass model (SURVEY 8d); the kernel batches 256 lags "
                                 "per workgroup and culls grid points outside the small FOV, so frac can exceed 1; "
                                 "the binding resource is fp64 VALU (see valu_fp64); kernel_ms = HIP events around single, "
                                 "non-overlapped launches right after the timed region"},
            "valu_fp64": {"achieved": valu_tf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                          "frac": valu_tf / FP64_VALU_PEAK_TF, "active_points": int(act),
                          "flop_per_point_lag": FLOP_PER_POINT_LAG},
            "precompute_ms": float(np.mean(pre_ms)),
            "pcie_inclusive": pcie,
            "argmax_lag_arcsec": [float(lag1[am[0]]), float(lag2[am[1]])],
            "injected_shift_arcsec": [truth["lag_crval1"], truth["lag_crval2"]],
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is timed at N = 1 only
            # the GPU box gives one GPU's job a 16-core share whatever os.cpu_count() says
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            cores = int(os.environ.get("COREG_CPU_CORES", min(avail, 16)))
            n_sample = args.cpu_sample or 48 * cores
            log(f"[bench] CPU baseline: {n_sample} lag-points on {cores} cores ...")
            cb, corr_cpu, subset = cpu_baseline(small, hs, large, hl, lags, n_sample, cores)
            out["cpu_baseline"] = cb
            d = np.abs(corr.ravel()[subset] - corr_cpu.ravel()[subset])
            out["parity_vs_cpu_sample"] = {"max_abs_dcorr": float(np.nanmax(d)), "n": int(len(subset)),
                                           "argmax_on_sample_equal": bool(np.nanargmax(corr.ravel()[subset]) ==
                                                                          np.nanargmax(corr_cpu.ravel()[subset]))}
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    for hk in handles:
        hk.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
