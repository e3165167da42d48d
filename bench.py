#!/usr/bin/env python3
"""
bench.py -- lag-points/sec of the alignment correlation sweep on MI355X.

Workload (BASELINE.json metric / SURVEY.md 8d "headline"): Carrington 'fa' sweep, 2048x2048 lon/lat grid,
lon (200, 300) deg, lat (-20, 20) deg, lag_crval1 = lag_crval2 = arange(-30, 30, 1) arcsec (60 x 60 = 3600
lag-points), crota/cdelt fixed, solar_r 1.004, order 2, method 'correlation'; seeded synthetic HRIEUV-like
2048^2 image to align and FSI-like 3072^2 reference (euispice_coreg_amd/synthetic.py).

A "step" = one full sweep (all 3600 lag-points) through the C ABI with both images already RESIDENT in HBM
(upload + once-only reference preparation happen before the timed region; `config.resident` says so).  That is what the
driver's contract defines `value` on (inputs resident when the timed region starts).  The metric as BASELINE.md section
2 words it -- L / wall time of ONE call that is handed host images: upload, reference preparation, sweep, all-gather,
map on the host -- is measured right after the timed region AT EVERY N and printed as `pcie_inclusive` (never as
`value`).  `--streams` (default 2) steps are in flight at a time, each on its own HIP stream and library context, so that
the drain of one sweep overlaps the start of the next; every step is a complete sweep, and the rate with ONE sweep in
flight is printed as `one_sweep_in_flight`.  The scene masks 0.5 % of the pixels of the image to align at random (SURVEY
8d); the same sweep on that image with the NaN pixels filled in is printed as `all_finite_image` (its interior tile
visits run without the sample mask; `--nan-frac 0` makes that the scene of the whole run, for profiling).

N > 1: `python bench.py --gpus N` starts its own N rank processes (torch.distributed.run, one rank per GPU) as CHILDREN,
before anything in this process has touched the GPU, and exits with their return code; launched under torchrun
(WORLD_SIZE set) it is a rank.  The lag set is spread exactly as `hdrshift.Alignment` spreads it
(euispice_coreg_amd.parallel.lag_sharding): the (CRVAL1, CRVAL2) lag plane is cut in N blocks (the multi-GPU form of the
reference's np.array_split fan-out, alignment.py:677-687), every rank sweeps its block with full image replicas and
ONE all-gather (RCCL) of the per-lag coefficients, followed by an index permutation, assembles the map on every rank.
Total work is fixed as N grows -> "scaling": "strong".  The N > 1 line verifies itself: after the timed region rank 0
sweeps the FULL lag set on its own GPU (`map_vs_single_gpu`), a CPU sample of 128 lag-points is evaluated by the oracle
(`parity_vs_cpu_sample`), and `n_ranks_seen` is the size of the communicator the collective ran on.

Prints ONE JSON line on rank 0 (see the driver contract) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GRID_SHAPE = (2048, 2048)
LONLIMS = (200.0, 300.0)
LATLIMS = (-20.0, 20.0)
SOLAR_R = 1.004
ORDER = 2
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TF = 78.6  # MI355X vector FP64 spec: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4.0  # wave-instructions/s: one VALU instruction per SIMD every 4 cycles
LDS_CYCLES_PEAK = 256 * 2.4e9            # LDS-array cycles/s, all CUs
# Counted float64 operations of k_sweep<TRANSLATE, 2> per (grid point, lag), interior-window path (kernels.hpp
# point_lag): 2 add (coordinate) + 2 fract + 10 (doubled spline weights, 5 per axis) + 3 x (mul + 2 fma) rows
# + (mul + 2 fma) column + sums (2 add + 3 fma) = 31 float64 instructions = 42 flop with fma = 2.  (The whole loop
# body is 37 VALU instructions: + 2 cvt, 1 class test, the integer count and 2 of address arithmetic; the same 42 is
# used for the all-finite variant, which executes 39 of them per sample and the other 3 per chunk of four points.)
FLOP_PER_POINT_LAG = 42.0
F64_INSTR_PER_POINT_LAG = 31.0

METRIC = "lag-points/sec (whole node) on 2048² grid, 60×60 CRVAL sweep; argmax-shift match"
try:
    with open(os.path.join(ROOT, "BASELINE.json")) as _f:
        METRIC = json.load(_f).get("metric", METRIC)
except Exception:
    pass


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ---- N > 1 never ends without a line -------------------------------------------------------------
# The reference joins its workers in a busy-wait with no limit (alignment.py:723-744): a worker that dies leaves zeros,
# one that hangs leaves the caller waiting for ever.  Here every wait has a limit and every way out prints ONE JSON line:
#   * the process group is created with a timeout (COREG_BENCH_PG_TIMEOUT_S, default 300 s: rendezvous and, with RCCL,
#     every collective -- torch's watchdog ends the rank when one stays out that long);
#   * every rank has a deadline (COREG_BENCH_DEADLINE_S, default 900 s) and a SIGTERM watcher (torchrun ends the
#     surviving ranks with SIGTERM when one of them dies) on a thread of its own, so that a main thread stuck inside a
#     collective cannot keep the line from being printed; rank 0 prints {"error": ..., "value": null, ...}, exit != 0;
#   * `python bench.py --gpus N` (the self-launcher) watches its children: when they end non-zero, or the deadline
#     passes, it ends THEIR process group (never a pattern), and prints the error line itself if rank 0 did not.
_LINE_PRINTED = [False]
_STDOUT_FD = [None]     # the real stdout once fd 1 has been pointed at stderr
_RUN_INFO = {"n_gpus": None, "n_ranks_seen": 0, "steps": None, "warmup": None}


def _mark(what, rank):
    d = os.environ.get("COREG_BENCH_MARK_DIR")
    if d:
        try:
            open(os.path.join(d, f"{what}_{rank}"), "w").close()
        except OSError:
            pass


def error_line(msg, **extra):
    out = {"metric": METRIC, "value": None, "unit": "lag-points/s", "n_gpus": _RUN_INFO["n_gpus"],
           "n_ranks_seen": _RUN_INFO["n_ranks_seen"], "steps": _RUN_INFO["steps"], "warmup": _RUN_INFO["warmup"],
           "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic", "error": str(msg)[:600]}
    out.update(extra)
    return json.dumps(out)


def print_line_once(text):
    """The ONE line of this run on the real stdout (whatever fd 1 points at by now); False when it was printed before."""
    if _LINE_PRINTED[0]:
        return False
    _LINE_PRINTED[0] = True
    data = (text + "\n").encode()
    fd = _STDOUT_FD[0] if _STDOUT_FD[0] is not None else 1
    try:
        sys.stdout.flush()
    except Exception:
        pass
    os.write(fd, data)
    return True


def arm_rank_watchdogs(rank):
    """Deadline + SIGTERM watcher of a rank process, both on daemon threads (see above)."""
    import signal
    import threading
    deadline = float(os.environ.get("COREG_BENCH_DEADLINE_S", "900"))

    def give_up(msg, code):
        if rank == 0:
            print_line_once(error_line(msg))
        log(f"[bench] rank {rank}: {msg}")
        os._exit(code)

    t = threading.Timer(deadline, give_up, (f"deadline of {deadline:.0f} s passed (COREG_BENCH_DEADLINE_S)", 124))
    t.daemon = True
    t.start()
    try:
        r, w = os.pipe()
        os.set_blocking(w, False)
        signal.set_wakeup_fd(w, warn_on_full_buffer=False)
        signal.signal(signal.SIGTERM, lambda *_: None)  # (the wake-up fd is what is acted on)

        def watch():
            while True:
                b = os.read(r, 1)
                if b and b[0] == signal.SIGTERM:
                    give_up("terminated by the launcher (SIGTERM): another rank failed or the launcher's limit passed", 143)
        th = threading.Thread(target=watch, daemon=True)
        th.start()

        # processes forked from this one (the oracle's worker pool of the CPU parity sample) inherit the wake-up pipe: a
        # SIGTERM sent to one of THEM -- multiprocessing.Pool ends its workers that way -- would be read here as ours
        def in_child():
            signal.set_wakeup_fd(-1)
            signal.signal(signal.SIGTERM, signal.SIG_DFL)
        os.register_at_fork(after_in_child=in_child)
    except (ValueError, OSError):  # not the main thread / no signals here: the deadline still holds
        pass
    return t


def self_launch(n, argv):
    """--gpus N > 1 without a torchrun environment: start the N ranks as children and relay their result.  Runs before
    torch / HIP are imported here (a process that has initialised the GPU must never exec or fork GPU workers).  The
    children's stdout comes through a pipe, so that this process knows whether the ONE line was printed."""
    import shutil
    import signal
    import socket
    import subprocess
    import tempfile
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "2")
    deadline = float(env.get("COREG_BENCH_DEADLINE_S", "900"))
    env["COREG_BENCH_DEADLINE_S"] = str(deadline)
    marks = tempfile.mkdtemp(prefix="coreg_bench_marks_")
    env["COREG_BENCH_MARK_DIR"] = marks
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    log("[bench] launching", n, "ranks:", " ".join(cmd))
    t0 = time.time()
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
    seen = []

    def relay():
        for raw in p.stdout:
            line = raw.decode(errors="replace")
            if line.strip():
                seen.append(line)
                sys.stdout.write(line)
                sys.stdout.flush()
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    why = None
    try:
        rc = p.wait(timeout=deadline + 30.0)  # (the ranks' own deadline comes first and prints the line itself)
    except subprocess.TimeoutExpired:
        why = f"the ranks did not end within {deadline + 30.0:.0f} s"
        rc = 124
    if why is not None or rc != 0:
        # end exactly the process group started above (torchrun and its ranks), first politely
        for sig, wait in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 5.0)):
            try:
                os.killpg(p.pid, sig)
            except (ProcessLookupError, PermissionError):
                break
            try:
                p.wait(timeout=wait)
                break
            except subprocess.TimeoutExpired:
                continue
    th.join(timeout=5.0)
    joined = len([f for f in os.listdir(marks) if f.startswith("joined_")])
    started = len([f for f in os.listdir(marks) if f.startswith("started_")])
    shutil.rmtree(marks, ignore_errors=True)
    if rc != 0 and not any('"metric"' in ln for ln in seen):
        _RUN_INFO.update(n_gpus=n, n_ranks_seen=joined)
        print(error_line(why or f"the ranks ended with return code {rc} before rank 0 printed its line",
                         ranks_started=started, launcher_seconds=round(time.time() - t0, 1)), flush=True)
    return rc if rc != 0 else 0


def pmc_summary():
    """Counter means per k_sweep launch from the newest committed rocprofv3 PMC summary of this command
    (profiles/rNN_pmc_summary.txt, written by profiles/run_pmc.sh).  {} when none is committed."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary.txt")))
    if not paths:
        return {}, None
    vals = {}
    try:
        for line in open(paths[-1]):
            parts = line.split()
            if len(parts) >= 4 and parts[1] == "mean/dispatch":
                vals[parts[0]] = float(parts[3])
    except Exception:
        return {}, None
    return vals, os.path.relpath(paths[-1], ROOT)


def kernel_profile_ms():
    """Average duration of the headline sweep kernel in the newest committed rocprofv3 --kernel-trace --stats summary of
    this command with one sweep in flight (profiles/rNN_kernel_stats_streams1.csv); (None, None) when none is committed."""
    import csv
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_kernel_stats_streams1.csv")))
    if not paths:
        return None, None
    try:
        with open(paths[-1], newline="") as f:
            for row in csv.DictReader(f):
                if "k_sweep" in row.get("Name", ""):
                    return float(row["AverageNs"]) * 1e-6, os.path.relpath(paths[-1], ROOT)
    except Exception:
        pass
    return None, None


def cpu_baseline(np, small, hs, large, hl, lags, n_sample, cores):
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


def main_threads(args):
    """`--launch threads`: this ONE process drives args.gpus GPUs through the library's multi-GPU driver (no
    torch.distributed).  A step = one whole sweep call: per-device sweeps of the lag-plane blocks, the one RCCL all-gather,
    the map on the host (the call is synchronous, so steps do not overlap)."""
    import numpy as np
    from euispice_coreg_amd import _lib, synthetic
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    small, hs, large, hl, truth = synthetic.make_scene(float32_exact=not args.small_f64, nan_frac=args.nan_frac)
    lag1 = lag2 = np.arange(-30, 30, 1, dtype=np.float64)
    lags = _lib.LagSet(lag1, lag2, None, None, None)
    L = lags.size
    grid = _lib.Grid(LONLIMS, LATLIMS, GRID_SHAPE, numpy_lat_trig=True)
    n = args.gpus if args.gpus > 0 else 0
    with _lib.MultiHandle(n) as m:
        m.set_option("use_lds", args.use_lds)
        m.set_small(small)
        m.prepare_reference_carrington(large, hl, grid, SOLAR_R, ORDER)
        for _ in range(args.warmup):
            corr = m.sweep_carrington(hs, grid, SOLAR_R, lags, order=ORDER)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            corr = m.sweep_carrington(hs, grid, SOLAR_R, lags, order=ORDER)
        elapsed = time.perf_counter() - t0
        st = m.last_stats()
        corr = corr.reshape(lag1.size, lag2.size)
        single = m.handle(0).sweep_carrington(hs, grid, SOLAR_R, lags, order=ORDER).reshape(corr.shape)
        small_h = small if args.small_f64 else small.astype(np.float32)
        large_h = large.astype(np.float32)
        times = []
        for _ in range(5):
            t0 = time.perf_counter()
            m.set_small(small_h)
            m.prepare_reference_carrington(large_h, hl, grid, SOLAR_R, ORDER)
            out_host = m.sweep_carrington(hs, grid, SOLAR_R, lags, order=ORDER)
            times.append(time.perf_counter() - t0)
        best = min(times[1:])
        am = np.unravel_index(np.nanargmax(corr), corr.shape)
        out = {"metric": METRIC, "value": L * args.steps / elapsed, "unit": "lag-points/s", "n_gpus": m.size,
               "n_ranks_seen": m.size, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "headline: Carrington 'fa' 2048x2048 grid, 60x60 CRVAL lags (see bench.py)",
                          "lag_points": L, "grid": list(GRID_SHAPE), "resident": True, "sweeps_in_flight": 1,
                          "launch": "threads", "lag_sharding": st["lag_sharding"],
                          "parallelism": f"one process, {m.size} device threads, lag-plane blocks + 1 all-gather "
                                         f"({st['collective']})"},
               "per_rank": [{"rank": k, "kernel_ms": ms} for k, ms in enumerate(st["per_device_sweep_kernel_ms"])],
               "map_vs_single_gpu": {"max_abs_diff": float(np.nanmax(np.abs(single - corr))),
                                     "same_argmax": bool(np.nanargmax(single) == np.nanargmax(corr)), "tolerance": 1e-12},
               "pcie_inclusive": {"value": L / best, "unit": "lag-points/s", "ms_per_step": 1e3 * best, "n_gpus": m.size,
                                  "identical_to_resident_map": bool(np.array_equal(out_host.reshape(corr.shape), corr,
                                                                                   equal_nan=True))},
               "argmax_lag_arcsec": [float(lag1[am[0]]), float(lag2[am[1]])],
               "injected_shift_arcsec": [truth["lag_crval1"], truth["lag_crval2"]],
               "roofline": None, "cpu_baseline": None}
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)


# ---- the other sweep variants / BASELINE configs, one GPU, same schema -----------------------------
# Counted float64 operations per (active grid point, lag) of each variant's interior LDS path (kernels.hpp point_lag),
# with the conventions of the headline's 42: fma = 2 flop elsewhere, the spline-weight instructions 1 each, conversions /
# class tests / address arithmetic 0.  coordinate: TRANSLATE 2 add; HOMOGRAPHY_SERIES eps (fma + mul) + q (fma) + two
# numerators (4 fma) + two corrections (2 fma) + 2 window offsets = 19; HOMOGRAPHY w (2 fma) + rcp + Newton (2 fma) +
# numerators (4 fma) + 2 mul + 2 offsets = 21; CAR rotation (6 fma + 3 mul) + hypot (fma + mul + sqrt) + 2 atan2 counted 1
# each + linear map (4 fma) + 2 offsets = 31.  2 fract.  weights: order 1: 2, order 2: 10, order 3: 30.  taps, N = order
# + 1: N rows of (mul + (N - 1) fma) + one column of the same = (N + 1)(2N - 1): 9 / 20 / 35 (+ 1 scale at order 3).
# float32 rounding of the sample (helioprojective frames): 1 sub of the pivot.  sums: 2 add + 3 fma = 8.
def _variant_flop(mode, order, rounded):
    coord = {"TRANSLATE": 2, "HOMOGRAPHY_SERIES": 19, "HOMOGRAPHY": 21, "CAR": 31}[mode]
    weights = {1: 2, 2: 10, 3: 30}[order]
    n = order + 1
    taps = (n + 1) * (2 * n - 1) + (1 if order == 3 else 0)
    return float(coord + 2 + weights + taps + (1 if rounded else 0) + 8)


def _committed(kind, name):
    """Newest committed profile summary of `bench.py --config name`: kind 'pmc' -> profiles/rNN_pmc_<name>.txt (counter
    means per launch), 'stats' -> profiles/rNN_kernel_stats_<name>.csv (rocprofv3 --kernel-trace --stats)."""
    import glob
    pat = {"pmc": f"r[0-9][0-9]_pmc_{name}.txt", "stats": f"r[0-9][0-9]_kernel_stats_{name}.csv"}[kind]
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
    return paths[-1] if paths else None


def main_config(args):
    """`--config NAME`: one of the other BASELINE configs / sweep variants on ONE GPU, inputs resident, K synchronous
    sweep calls (map copied to the host every step).  Same line as the headline's, `roofline` from the variant's own
    counted flop; the headline metric is NOT what `value` is here -- `config.workload` says what it is."""
    import numpy as np
    from euispice_coreg_amd import _lib, synthetic
    name = args.config
    sys.stdout.flush()
    _STDOUT_FD[0] = os.dup(1)
    os.dup2(2, 1)
    h = _lib.CoregHandle(0)
    h.set_option("use_lds", args.use_lds)
    for kv in filter(None, os.environ.get("COREG_BENCH_OPTS", "").split(",")):
        k, v = kv.split("=")
        h.set_option(k, int(v))
    carr = None
    order, mode, rounded, frame, par = 2, "TRANSLATE", False, "carrington", True
    if name == "cfg4":
        small, hs, large, hl, truth = synthetic.make_scene(small_shape=(832, 192), small_cdelt=(4.0, 1.098),
                                                           small_unit="deg", large_n=3072)
    elif name == "car":
        small, hs, large, hl, truth = synthetic.make_car_scene(small_shape=(768, 1024), large_shape=(1200, 1600), n_blobs=60)
    else:
        small, hs, large, hl, truth = synthetic.make_scene()
    if name in ("order1", "order3", "cfg3", "cfg5"):
        order = {"order1": 1, "order3": 3}.get(name, 2)
        shape = (4096, 4096) if name == "cfg5" else GRID_SHAPE
        carr = dict(shape=list(shape), lonlims=list(LONLIMS), latlims=list(LATLIMS))
        if name == "cfg3":
            lag_arrays = (np.arange(-60, 61, 1.0), np.arange(-60, 61, 1.0), None, None, None)
            what = "BASELINE configs[2]: Carrington 'fa' 2048^2 grid, 121x121 CRVAL lags"
        elif name == "cfg5":
            lag_arrays = (np.arange(-20, 21, 1.0), np.arange(-20, 21, 1.0), np.round(np.arange(-2, 3) * 0.01, 10),
                          np.round(np.arange(-2, 3) * 0.01, 10), np.round(np.arange(-5, 6) * 0.1, 10))
            what = "BASELINE configs[4]: 5-D sweep, Carrington 4096^2 grid, 41x41 CRVAL x 5x5 CDELT x 11 CROTA lags"
        else:
            lag_arrays = (np.arange(-30, 30, 1.0), np.arange(-30, 30, 1.0), None, None, None)
            what = f"the headline sweep at reprojection_order={order}"
        grid = _lib.Grid(LONLIMS, LATLIMS, shape, numpy_lat_trig=True)
        h.set_small(small)
        h.prepare_reference_carrington(large, hl, grid, SOLAR_R, order)
        lags = _lib.LagSet(*lag_arrays)

        def sweep():
            return h.sweep_carrington(hs, grid, SOLAR_R, lags, order=order)
        b_lag = shape[0] * shape[1] * 8 + small.size * 8
    elif name in ("cfg2", "cfg4"):
        frame, rounded = "helioprojective", True
        if name == "cfg2":
            mode = "HOMOGRAPHY_SERIES"
            lag_arrays = (np.arange(-30, 31, 1.0), np.arange(-30, 31, 1.0), None, None, None)
            what = "BASELINE configs[1]: helioprojective, sub-map semantics, 2048^2 vs 3072^2, 61x61 CRVAL lags"
        else:
            mode = "HOMOGRAPHY"
            lag_arrays = (np.arange(-30, 31, 1.0) / 3600, np.arange(-30, 31, 1.0) / 3600, None, None,
                          np.round(np.arange(-10, 11) * 0.1, 10))
            what = ("BASELINE configs[3]: SPICE-like raster 192x832 (header in degrees) vs 3072^2, helioprojective, "
                    "61x61 CRVAL x 21 CROTA lags")
        h.set_small(small)
        h.prepare_reference_helioprojective(large, hl, hs, 2)
        lags = _lib.LagSet(*lag_arrays)

        def sweep():
            return h.sweep_helioprojective(hs, hs, lags)
        b_lag = small.size * 4 + small.size * 8  # float32 sub-map on the small grid + the float64 image
    elif name == "car":
        frame, mode, rounded, par = "helioprojective", "CAR", True, False
        lag_arrays = (np.arange(-15, 16) * 0.004, np.arange(-15, 16) * 0.004, None, None, None)
        what = ("plate-carree maps (align_using_initial_carrington geometry): 768x1024 map against a 1200x1600 reference "
                "map on the reference's grid, 31x31 CRVAL lags of 0.004 deg")
        h.set_small(small)
        h.set_reference_on_grid(np.asarray(large, dtype=np.float32))
        lags = _lib.LagSet(*lag_arrays)

        def sweep():
            return h.sweep_helioprojective(hl, hs, lags)
        b_lag = large.size * 4 + small.size * 8
    else:
        raise SystemExit(f"unknown --config {name}")
    L = lags.size
    steps = args.steps if args.steps_given else {"cfg5": 3, "cfg2": 20, "cfg4": 20}.get(name, 50)
    warm = args.warmup if args.warmup_given else {"cfg5": 1}.get(name, 5)
    _RUN_INFO.update(n_gpus=1, n_ranks_seen=1, steps=steps, warmup=warm)
    for _ in range(warm):
        corr = sweep()
    h.synchronize()
    t0 = time.perf_counter()
    kern, pre = [], []
    for _ in range(steps):
        corr = sweep()
        st = h.last_stats()
        kern.append(st["sweep_kernel_ms"])
        pre.append(st["precompute_ms"])
    h.synchronize()
    elapsed = time.perf_counter() - t0
    corr = np.asarray(corr).reshape(lags.shape)
    n_launch = max(1, int(st["n_sweep_launches"]))
    k_ms = float(np.mean(kern))                    # all launches of one sweep
    act = int(st["n_active_points"])               # per launch
    lags_per_launch = L // n_launch
    flop = _variant_flop(mode, order, rounded)
    k_s = k_ms * 1e-3
    achieved_tf = flop * act * L / k_s / 1e12
    roof = {"bound": "valu_fp64+lds", "achieved": achieved_tf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
            "frac": achieved_tf / FP64_VALU_PEAK_TF,
            "kernel": f"k_sweep<{mode},{order},{'f32' if st['small_is_f32'] else 'f64'}>", "launches_per_sweep": n_launch,
            "kernel_ms": k_ms / n_launch, "kernel_ms_per_sweep": k_ms, "lags_per_launch": lags_per_launch,
            "active_points": act, "flop_per_point_lag": flop,
            "ps_per_point_lag": k_ms * 1e9 / max(1, act * L), "traffic": None,
            "hbm_model": {"algorithmic_bytes_per_lag": b_lag, "algorithmic_bytes_per_launch": b_lag * lags_per_launch,
                          "achieved_gbs": b_lag * L / k_s / 1e9, "peak_gbs": HBM_PEAK_GBS,
                          "hbm_model_frac": b_lag * L / k_s / 1e9 / HBM_PEAK_GBS,
                          "note": "SURVEY 8d one-lag-per-pass byte model; every staged byte is shared between the 256 "
                                  "lags of a workgroup, so this is not the bound (see traffic)"},
            "note": "achieved = counted float64 flop of the variant's interior gather path x active points x lags / "
                    "kernel time (HIP events around every k_sweep launch of the timed steps)"}
    pmc_path = _committed("pmc", name)
    if pmc_path:
        vals = {}
        for line in open(pmc_path):
            parts = line.split()
            if len(parts) >= 4 and parts[1] == "mean/dispatch":
                vals[parts[0]] = float(parts[3])
        per_launch_s = k_s / n_launch
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            roof["traffic"] = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0  # gfx950 half-count of FETCH_SIZE
            roof["traffic_unit"] = "HBM bytes per launch"
            roof["hbm_frac_measured"] = roof["traffic"] / per_launch_s / 1e9 / HBM_PEAK_GBS
        if "SQ_INSTS_VALU" in vals:
            roof["valu_issue_frac"] = vals["SQ_INSTS_VALU"] / per_launch_s / VALU_ISSUE_PEAK
            roof["valu_instr_per_wave_sample"] = vals["SQ_INSTS_VALU"] / max(1.0, act * lags_per_launch / 64.0)
        if "SQ_LDS_IDX_ACTIVE" in vals:
            roof["lds_busy_frac"] = vals["SQ_LDS_IDX_ACTIVE"] / per_launch_s / LDS_CYCLES_PEAK
            if "SQ_LDS_BANK_CONFLICT" in vals:
                roof["lds_conflict_cycle_frac"] = vals["SQ_LDS_BANK_CONFLICT"] / vals["SQ_LDS_IDX_ACTIVE"]
        roof["pmc_source"] = os.path.relpath(pmc_path, ROOT)
    stats_path = _committed("stats", name)
    if stats_path:
        import csv
        with open(stats_path, newline="") as f:
            for row in csv.DictReader(f):
                if "k_sweep" in row.get("Name", ""):
                    roof["kernel_profile_ms"] = float(row["AverageNs"]) * 1e-6
                    roof["kernel_profile_source"] = os.path.relpath(stats_path, ROOT)
                    break
    am = np.unravel_index(np.nanargmax(corr), corr.shape)
    out = {"metric": f"lag-points/sec, one GPU, config '{name}' (NOT the headline metric: see config.workload)",
           "value": L * steps / elapsed, "unit": "lag-points/s", "n_gpus": 1, "n_ranks_seen": 1, "steps": steps,
           "warmup": warm, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": what, "name": name, "lag_points": L, "lag_shape": list(lags.shape), "resident": True,
                      "sweeps_in_flight": 1, "order": order, "parallelism": "single GPU, no collective",
                      "step": "one synchronous sweep call, correlation map copied to the host"},
           "roofline": roof, "precompute_ms": float(np.mean(pre)), "argmax_index": [int(i) for i in am],
           "cpu_baseline": None}
    if not args.no_cpu_baseline:
        from oracle import coreg_oracle as O
        from tests import helpers as H
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        cores = int(os.environ.get("COREG_CPU_CORES", min(avail, 16)))
        n_sample = args.cpu_sample or {"cfg4": 64 * cores, "car": 8 * cores, "order1": 32 * cores}.get(name, 2 * cores)
        n_sample = min(n_sample, L)
        st_o = H.oracle_state(small, hs, large, hl, lag_arrays, order=order, unit_lag=hs["CUNIT1"],
                              **({} if carr is None else dict(shape=carr["shape"], lonlims=carr["lonlims"],
                                                              latlims=carr["latlims"], solar_r=(SOLAR_R,))))
        O.set_initial_header_values(st_o, use_ang2pipi=(name != "car"))
        ref = O.prepare_reference(st_o, frame, SOLAR_R, parallelism=par)
        subset = np.sort(np.random.default_rng(1234).choice(L, size=n_sample, replace=False))
        log(f"[bench] CPU baseline: {n_sample} lag-points on {cores} cores ...")
        t0 = time.perf_counter()
        c_cpu = O.find_best_header_parameters(st_o, frame, parallelism=par, counts=cores, lag_subset=subset,
                                              prepared_reference=ref, use_ang2pipi=(name != "car"))
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n_sample / dt, "unit": "lag-points/s", "cores": cores, "kind": "port",
                               "sample": f"{n_sample} seeded random lag-points of this config, {cores} worker processes, "
                                         f"{dt:.1f} s wall"}
        d = np.abs(corr.ravel()[subset] - c_cpu.ravel()[subset])
        out["parity_vs_cpu_sample"] = {"max_abs_dcorr": float(np.nanmax(d)), "n": int(n_sample),
                                       "tolerance": 1e-10 if frame == "carrington" else 1e-7}
    h.close()
    print_line_once(json.dumps(out))



def reference_run_points(np, small, large, corr):
    """89 entries of the headline map as the REFERENCE ITSELF computed them in the build container (its own
    Alignment.align_using_carrington on a sub-lattice of the lags: tests/golden/make_golden_headline_reference.py ->
    tests/golden/headline_reference.npz, data only).  None when this run's scene is not the fixture's (--nan-frac,
    --small-f64): the fingerprint of the pixels decides."""
    path = os.path.join(ROOT, "tests", "golden", "headline_reference.npz")
    if not os.path.isfile(path):
        return None
    r = np.load(path)
    fp = np.array([np.nansum(small), np.nansum(large), float(np.isnan(small).sum()), small[1000, 1000], large[1500, 1500]])
    if not np.array_equal(fp, r["fingerprint"]):
        return None
    d = np.abs(corr.ravel()[r["index"]] - r["corr"])
    return {"max_abs_dcorr": float(np.max(d)), "n": int(r["index"].size),
            "argmax_equal": bool(int(r["index"][np.argmax(r["corr"])]) == int(np.nanargmax(corr))),
            "what": "entries of this map computed by the reference's own Alignment in the build container"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)   # default 300: ~3.6 ms each, a timed region above one second
    ap.add_argument("--warmup", type=int, default=None)  # default 30
    ap.add_argument("--config", choices=["headline", "cfg2", "cfg3", "cfg4", "cfg5", "order1", "order3", "car"],
                    default="headline",
                    help="headline (default, the driver's line) or one of the other BASELINE configs / sweep variants on "
                         "one GPU: same schema, roofline of that variant's kernel (profiles/r06_*_<config>.*)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=2, help="sweeps in flight (one HIP stream + library context each)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive measurement after the timed region")
    ap.add_argument("--cpu-sample", type=int, default=0, help="lag-points in the CPU sample (0 = 48 per core)")
    ap.add_argument("--use-lds", type=int, default=1)
    ap.add_argument("--no-all-finite", action="store_true",
                    help="skip the side measurement on the NaN-free image after the timed region (profiling runs: its "
                         "sweeps would be averaged into the per-kernel figures)")
    ap.add_argument("--nan-frac", type=float, default=0.005,
                    help="fraction of NaN pixels in the image to align (scene default 0.005; 0: every LDS window is "
                         "all-finite and interior visits run without the sample mask)")
    ap.add_argument("--sustained-seconds", type=float, default=2.0,
                    help="after the timed region the same steps are issued for at least this long and reported as "
                         "`sustained` (0: skip; profiling runs)")
    ap.add_argument("--small-f64", action="store_true",
                    help="image to align with full float64 pixels (not float32-exact): times the TS = double kernel")
    ap.add_argument("--launch", choices=["ranks", "threads"], default="ranks",
                    help="N > 1: one process per GPU over torch.distributed / RCCL (the driver's form), or ONE process "
                         "driving all N GPUs through the library's own multi-GPU driver (coreg_multi: a host thread per "
                         "device, one RCCL all-gather) -- what a plain `Alignment(parallelism=True)` script uses")
    ap.add_argument("--shard", choices=["auto", "lags", "points"], default="auto",
                    help="N > 1: cut the lag plane in blocks (+ one all-gather) or the target grid in point shares "
                         "(+ one all-reduce of the six sums per lag); auto = points below 128 lag-points per GPU")
    args = ap.parse_args()
    args.steps_given, args.warmup_given = args.steps is not None, args.warmup is not None
    if args.config != "headline":
        if args.gpus > 1 or "WORLD_SIZE" in os.environ:
            raise SystemExit("--config other than headline runs on one GPU")
        return main_config(args)
    args.steps = 300 if args.steps is None else args.steps
    args.warmup = 30 if args.warmup is None else args.warmup

    if args.launch == "threads" and "WORLD_SIZE" not in os.environ:
        return main_threads(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    _RUN_INFO.update(n_gpus=world, n_ranks_seen=0, steps=args.steps, warmup=args.warmup)
    _mark("started", rank)

    # native libraries (RCCL prints a version banner on first use) write to fd 1: keep stdout for the ONE JSON line
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    _STDOUT_FD[0] = saved_stdout
    os.dup2(2, 1)
    if world > 1 or "COREG_BENCH_DEADLINE_S" in os.environ:
        arm_rank_watchdogs(rank)
    fail_rank = int(os.environ.get("COREG_BENCH_TEST_FAIL_RANK", "-1"))  # tests of the launcher: a rank that dies
    fail_at = os.environ.get("COREG_BENCH_TEST_FAIL_AT", "init")
    if rank == fail_rank and fail_at == "init":
        os._exit(7)

    import numpy as np
    import torch
    import torch.distributed as dist
    from euispice_coreg_amd import parallel

    # COREG_BENCH_BACKEND=gloo: rehearse the N > 1 path on a box with fewer GPUs than ranks (ranks share devices,
    # the all-gather runs on the CPU); the driver's runs use the default, nccl (= RCCL over xGMI)
    backend = os.environ.get("COREG_BENCH_BACKEND", "nccl")
    # COREG_BENCH_DRY=1 (tests of the launcher on a box without a GPU): every GPU call is left out -- each rank fills
    # its block with the raveled lag indices instead of sweeping -- so that process start-up, the process group, the
    # collective, the block permutation and the JSON line can be checked.  The line says "dry_run": true, value null.
    dry = os.environ.get("COREG_BENCH_DRY", "0") == "1"
    if dry and backend == "nccl":
        raise SystemExit("COREG_BENCH_DRY=1 needs COREG_BENCH_BACKEND=gloo")
    dev = "cpu" if dry else "cuda"
    if not dry:
        from euispice_coreg_amd import _lib, synthetic
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
        import datetime
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get("COREG_BENCH_PG_TIMEOUT_S", "300")))
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
        else:
            dist.init_process_group(backend=backend, timeout=pg_timeout)
    n_ranks_seen = dist.get_world_size() if use_dist else 1
    _RUN_INFO["n_ranks_seen"] = n_ranks_seen
    _mark("joined", rank)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    lag1 = np.arange(-30, 30, 1, dtype=np.float64)
    lag2 = np.arange(-30, 30, 1, dtype=np.float64)
    lags = (lag1, lag2, None, None, None)
    L = lag1.size * lag2.size
    # the partition Alignment would use for this lag set (parallel.lag_sharding), unless --shard forces one
    auto_mode = parallel.lag_sharding((lag1.size, lag2.size, 1, 1, 1), world)
    by_points = world > 1 and (args.shard == "points" or (args.shard == "auto" and auto_mode == "points"))
    if by_points:  # every rank sweeps ALL lag-points over its share of the grid
        lo1, hi1, lo2, hi2 = 0, lag1.size, 0, lag2.size
    else:
        lo1, hi1, lo2, hi2 = parallel.block_bounds(lag1.size, lag2.size, world, rank)
    have_lags = hi1 > lo1 and hi2 > lo2
    perm_np, chunk = parallel.block_gather_index((lag1.size, lag2.size, 1, 1, 1), world)
    if by_points:
        perm_np, chunk = np.arange(L, dtype=np.int64), L
    n_streams = max(1, args.streams)

    if dry:
        streams, handles, h = [None] * n_streams, [], None
        truth = {"lag_crval1": 17.0, "lag_crval2": -9.0}
        block = (np.arange(lo1, hi1)[:, None] * lag2.size + np.arange(lo2, hi2)[None, :]).astype(np.float64).ravel()
    else:
        t0 = time.time()
        small, hs, large, hl, truth = synthetic.make_scene(float32_exact=not args.small_f64, nan_frac=args.nan_frac)
        if rank == 0:
            log(f"[bench] scene built in {time.time() - t0:.1f} s; L = {L}")
        # `--streams S` sweeps are in flight at a time, each on its own HIP stream with its own library context: the
        # tail of one sweep (workgroups draining, the small finalize / next precompute kernels) overlaps the head of
        # the next.  Every step is still one complete sweep of all lag-points of this rank.
        streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(n_streams - 1)]
        grid = _lib.Grid(LONLIMS, LATLIMS, GRID_SHAPE, numpy_lat_trig=True)
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
            if by_points:
                hk.set_point_shard(rank, world)
            handles.append(hk)

        def same_pivots(hk):
            """Point shares: the ranks' six sums only add up about identical pivots -- rank 0's are used everywhere."""
            if not (by_points and use_dist):
                return
            piv = torch.tensor(hk.get_pivots(), dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.broadcast(piv, src=0)
            hk.set_pivots(float(piv[0]), float(piv[1]))

        for hk in handles:
            same_pivots(hk)
        h = handles[0]
        my_lags = _lib.LagSet(lag1[lo1:hi1], lag2[lo2:hi2], None, None, None) if have_lags else None

    mine = [torch.full((chunk,), float("nan"), dtype=torch.float64, device=dev) for _ in range(n_streams)]
    gathered = [torch.empty((chunk * world,), dtype=torch.float64, device=dev) if use_dist and not by_points else m
                for m in mine]
    sums = [None] * n_streams  # point sharding: the six sums per lag slot, all-reduced (sized after the first sweep)
    perm = torch.from_numpy(perm_np).to(dev)
    sync()
    result = [None]
    step_no = [0]

    def step(n_in_flight=n_streams):
        k = step_no[0] % n_in_flight
        step_no[0] += 1
        if dry:
            if have_lags:
                mine[k][:block.size] = torch.from_numpy(block)
            if by_points:  # stand-in for the sums: every rank contributes 1/world of the final values
                t = mine[k] / world
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                result[0] = t
            elif use_dist:
                parts = [torch.empty(chunk, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(parts, mine[k])
                result[0] = torch.cat(parts)[perm]
            else:
                result[0] = mine[k][perm]
            return
        with torch.cuda.stream(streams[k]):
            if have_lags:
                handles[k].sweep_carrington(hs, grid, SOLAR_R, my_lags, order=ORDER, out_dev_ptr=mine[k].data_ptr())
            if by_points:
                if sums[k] is None:
                    sums[k] = torch.empty(handles[k].sums_size(), dtype=torch.float64, device="cuda")
                if backend == "nccl":
                    handles[k].copy_sums(sums[k].data_ptr())
                    dist.all_reduce(sums[k], op=dist.ReduceOp.SUM)  # the ONE collective of this mode
                    handles[k].finalize_sums(sums[k].data_ptr(), L, out_dev_ptr=mine[k].data_ptr())
                else:
                    t = torch.from_numpy(handles[k].copy_sums())
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                    handles[k].finalize_sums(t.numpy(), L, out_dev_ptr=mine[k].data_ptr())
                result[0] = mine[k]
            elif use_dist:
                if backend == "nccl":
                    dist.all_gather_into_tensor(gathered[k], mine[k])  # the ONE collective of the path
                    result[0] = gathered[k][perm]
                else:
                    parts = [torch.empty(chunk, dtype=torch.float64) for _ in range(world)]
                    dist.all_gather(parts, mine[k].cpu())
                    result[0] = torch.cat(parts)[perm.cpu()]
            else:
                result[0] = mine[k]  # one rank: the block IS the map (the permutation is the identity)

    def timed(n_steps, n_in_flight):
        """EXACTLY n_steps steps between barrier + synchronize on both sides; max over ranks.  Seconds."""
        step_no[0] = 0
        if use_dist:
            dist.barrier()
        sync()
        t_start = time.perf_counter()
        for _ in range(n_steps):
            step(n_in_flight)  # asynchronous: the host plans step k+1 while the GPU runs step k; nothing is read back
        if use_dist:
            dist.barrier()
        sync()
        el = time.perf_counter() - t_start
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    for _ in range(args.warmup):
        step()
    if rank == fail_rank and fail_at == "step":
        os._exit(7)
    if rank == fail_rank and fail_at == "hang":  # a rank that stays out of the next collective for ever
        time.sleep(1e6)
    elapsed = timed(args.steps, n_streams)
    # the same steps for >= 2 s, so that the GPU is visibly busy to an outside observer and a clock that falls under
    # sustained load shows (the K-step region above lasts a second at most)
    sustained = None
    if args.sustained_seconds > 0 and not dry:
        n_sus = max(args.steps, int(1.1 * args.sustained_seconds / max(elapsed / args.steps, 1e-6)) + 1)
        t_sus = timed(n_sus, n_streams)
        sustained = {"steps": n_sus, "seconds": t_sus, "ms_per_step": 1e3 * t_sus / n_sus, "value": L * n_sus / t_sus,
                     "unit": "lag-points/s", "sweeps_in_flight": n_streams}
    # the same steps with ONE sweep in flight (stream 0 only), for the record
    one_in_flight = None
    if n_streams > 1 and not dry:
        n1 = max(1, min(args.steps, 100))
        one_in_flight = timed(n1, 1) / n1

    # dominant-kernel duration: the library brackets every k_sweep launch with HIP events on the launch stream.  Reading
    # them waits for the sweep, and with several sweeps in flight a kernel's event interval also contains the time its
    # workgroups waited for the other sweep's to leave the CUs -- so the kernel is timed here, right after the timed
    # region, on identical steps issued ONE at a time
    stats = {"sweep_kernel_ms": 0.0, "precompute_ms": 0.0, "n_active_points": 0, "small_is_f32": 1, "used_lds": 1}
    kernel_ms, pre_ms = [], []
    if not dry:
        for _ in range(min(16, args.steps)):
            step_no[0] = 0  # handle / stream 0
            step(1)
            if have_lags:
                stats = h.last_stats()
                kernel_ms.append(stats["sweep_kernel_ms"])
                pre_ms.append(stats["precompute_ms"])
            else:
                torch.cuda.synchronize()
    if not kernel_ms:
        kernel_ms, pre_ms = [0.0], [0.0]
    k_ms = float(np.mean(kernel_ms))
    per_rank = [{"rank": rank, "kernel_ms": k_ms, "lags": (hi1 - lo1) * (hi2 - lo2) if have_lags else 0,
                 "active_points": int(stats["n_active_points"])}]
    if use_dist:
        dist.barrier()
        gathered_stats = [None] * world
        dist.all_gather_object(gathered_stats, per_rank[0])
        per_rank = gathered_stats
    corr = result[0].cpu().numpy().reshape(lag1.size, lag2.size)

    # ---- BASELINE.md section 2 at every N: ONE call that is handed host images.  Every rank uploads its share of both
    # images (the float32 pixels a BITPIX=-32 FITS file holds: 16 + 36 MiB in all; image to align: with RCCL 1/N per rank +
    # one all-gather over xGMI, parallel.replicate_image, else whole; reference: the library sends only the rectangle the
    # grid can touch), prepares the reference, sweeps its block,
    # the all-gather assembles the map and every rank copies it to the host.  Barrier on both sides, max over ranks,
    # best of 4 after one warm-up.
    pcie = None
    if not args.no_pcie and not dry:
        small_h = small_m if args.small_f64 else small_m.astype(np.float32)
        large_h = large.astype(np.float32)
        times = []
        out_host = None
        for _ in range(5):
            if use_dist:
                dist.barrier()
            sync()
            t0 = time.perf_counter()
            with torch.cuda.stream(streams[0]):
                ts = parallel.replicate_image(small_h) if use_dist else None
                if ts is None:
                    # (round 5: the image to align goes up on the handle's upload stream, staged with non-temporal
                    # copies; the reference preparation below does not queue behind it.  The opt-in upload THREAD
                    # -- option "async_upload" -- measured +-0.03 ms here and is not used: profiles/r05_pcie_breakdown.log)
                    h.set_small(small_h)
                else:  # (the handle runs on streams[0] = torch's current stream here: stream-ordered, no sync needed)
                    h.set_small_from_device(ts.data_ptr(), ts.shape, small_h.dtype)
                # the reference: every rank sends the rectangle its grid can touch (the library crops: ~2 % of 36 MiB)
                h.prepare_reference_carrington(large_h, hl, grid, SOLAR_R, ORDER)
                same_pivots(h)
            if not use_dist and have_lags:
                # one rank: the call `hdrshift.Alignment` makes -- the library's sweep with the map written straight to
                # host memory (its own pinned read-back + stream sync; no tensor round trip in between)
                with torch.cuda.stream(streams[0]):
                    out_host = h.sweep_carrington(hs, grid, SOLAR_R, my_lags, order=ORDER)
            else:
                step_no[0] = 0
                step(1)  # sweep of this rank's block (+ the one collective) on stream / handle 0
                out_host = result[0].cpu().numpy()  # the map on the host: waits for everything above
            el = time.perf_counter() - t0
            if use_dist:
                t = torch.tensor([el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            times.append(el)
        best = min(times[1:])
        pcie = {"value": L / best, "unit": "lag-points/s", "ms_per_step": 1e3 * best, "n_gpus": world,
                "image_hand_over": ("image to align: 1/N per rank over its own PCIe link + one all-gather over xGMI; "
                                    "reference: every rank uploads the rectangle its grid can touch"
                                    if (use_dist and backend == "nccl" and world > 1) else
                                    "every rank uploads the image to align whole and the rectangle of the reference its "
                                    "grid can touch"),
                "identical_to_resident_map": bool(np.array_equal(out_host.reshape(corr.shape), corr, equal_nan=True)),
                "what": "BASELINE.md section 2 / SURVEY 8d: wall time of one call with host images in (2048^2 image to "
                        "align + 3072^2 reference, float32 = 52 MiB; reference re-prepared), sweep, all-gather, host "
                        "correlation map out; barrier on both sides, max over ranks, best of 4 after one warm-up"}

    # ---- for the record, N = 1: the same sweep on an image to align WITHOUT NaN pixels (the scene masks 0.5 % of them,
    # so every LDS window holds a NaN and every sample goes through the mask; level-2 imager data have none, and
    # interior visits then run without the mask -- kernels.hpp point_lag, CLEAN).  One sweep in flight, handle 0.
    all_finite = None
    visits = None
    if world == 1 and not dry and have_lags and not by_points:
        visits = h.last_visit_counts()
    if world == 1 and not dry and have_lags and not by_points and not args.no_all_finite:
        filled = np.where(np.isfinite(small_m), small_m, np.float64(np.nanmedian(small_m)))
        if not args.small_f64:
            filled = filled.astype(np.float32).astype(np.float64)
        h.set_small(filled)
        for _ in range(max(args.warmup, 30)):  # (the clocks fall while the PCIe-inclusive calls above leave the GPU idle)
            step_no[0] = 0
            step(1)
        n1 = max(1, min(args.steps, 100))
        t_one = timed(n1, 1) / n1
        kk = []
        for _ in range(min(16, args.steps)):
            step_no[0] = 0
            step(1)
            kk.append(h.last_stats()["sweep_kernel_ms"])
        all_finite = {"kernel_ms": float(np.mean(kk)), "ms_per_step": 1e3 * t_one,
                      "value": L / t_one, "unit": "lag-points/s", "sweeps_in_flight": 1,
                      "tile_visits": h.last_visit_counts(),
                      "what": "the headline sweep with the NaN pixels of the image to align filled in: interior tile "
                              "visits take the unmasked path; measured like one_sweep_in_flight / roofline.kernel_ms"}
        h.set_small(small_m)  # (back to the scene's image)

    # ---- N > 1: the gathered map against ONE GPU sweeping everything (rank 0, after the timed region)
    vs_single = None
    if world > 1 and not dry and rank == 0:
        single = h.sweep_carrington(hs, grid, SOLAR_R, _lib.LagSet(*lags), order=ORDER).reshape(corr.shape) \
            if not by_points else None
        if by_points:  # the handles are in point-shard mode: a fresh context for the unsharded sweep
            with _lib.CoregHandle(local_rank) as h1:
                h1.set_small(small_m)
                h1.prepare_reference_carrington(large, hl, grid, SOLAR_R, ORDER)
                single = h1.sweep_carrington(hs, grid, SOLAR_R, _lib.LagSet(*lags), order=ORDER).reshape(corr.shape)
        vs_single = {"max_abs_diff": float(np.nanmax(np.abs(single - corr))),
                     "same_nan_pattern": bool(np.array_equal(np.isnan(single), np.isnan(corr))),
                     "same_argmax": bool(np.nanargmax(single) == np.nanargmax(corr)), "tolerance": 1e-12,
                     "what": "rank 0 sweeps all 3600 lag-points on its own GPU after the timed region; compared with "
                             "the map the N ranks assembled"}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        G = GRID_SHAPE[0] * GRID_SHAPE[1]
        S = 2048 * 2048
        b_lag = G * 8 + S * 8  # reference grid value + small-image pixel, fp64, touched once per lag
        lags_per_launch = (hi1 - lo1) * (hi2 - lo2)
        act = int(stats["n_active_points"])
        k_s = max(k_ms, 1e-9) * 1e-3
        flops = act * lags_per_launch * FLOP_PER_POINT_LAG
        achieved_tf = flops / k_s / 1e12
        pmc, pmc_path = pmc_summary()
        roof = {
            # the binding resources are float64 VALU issue and the LDS gather, co-limited
            "bound": "valu_fp64+lds", "achieved": achieved_tf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
            "frac": achieved_tf / FP64_VALU_PEAK_TF,
            "kernel": "k_sweep<TRANSLATE,2,%s>" % ("f32" if stats["small_is_f32"] else "f64"), "kernel_ms": k_ms, "lags_per_launch": lags_per_launch,
            "active_points": act, "tile_visits": visits, "flop_per_point_lag": FLOP_PER_POINT_LAG,
            "f64_instr_per_point_lag": F64_INSTR_PER_POINT_LAG,
            # the workload-independent figure: the throughput above scales with the share of the grid inside the small
            # FOV (631 710 of 4 194 304 points on the README grid); this does not
            "ps_per_point_lag": k_ms * 1e9 / max(1, act * lags_per_launch),
            "traffic": None, "hbm_model": {
                "algorithmic_bytes_per_lag": b_lag, "algorithmic_bytes_per_launch": b_lag * lags_per_launch,
                "achieved_gbs": b_lag * lags_per_launch / k_s / 1e9, "peak_gbs": HBM_PEAK_GBS,
                "hbm_model_frac": b_lag * lags_per_launch / k_s / 1e9 / HBM_PEAK_GBS,
                "note": "SURVEY 8d one-lag-per-pass byte model; the kernel shares every staged byte between 256 lags "
                        "and culls grid points outside the small FOV, so this fraction exceeds 1 and HBM is not the "
                        "bound (see traffic)"},
            "note": "achieved = counted float64 flop of the interior gather path x active points x lags / kernel_ms "
                    "(HIP events around single, non-overlapped launches right after the timed region)"}
        if pmc and world == 1:
            if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                # KiB per dispatch; FETCH_SIZE doubled: gfx950 counts 128-B read requests as 64 B (MI355X_MICROARCH.md)
                roof["traffic"] = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
                roof["traffic_unit"] = "HBM bytes per launch"
            if "SQ_INSTS_VALU" in pmc:
                roof["valu_issue_frac"] = pmc["SQ_INSTS_VALU"] / k_s / VALU_ISSUE_PEAK
            if "SQ_LDS_IDX_ACTIVE" in pmc:
                roof["lds_busy_frac"] = pmc["SQ_LDS_IDX_ACTIVE"] / k_s / LDS_CYCLES_PEAK
                if "SQ_LDS_BANK_CONFLICT" in pmc:
                    roof["lds_conflict_cycle_frac"] = pmc["SQ_LDS_BANK_CONFLICT"] / pmc["SQ_LDS_IDX_ACTIVE"]
            roof["pmc_source"] = f"{pmc_path} (rocprofv3 --pmc means per launch of this command, committed)"
        prof_ms, prof_path = kernel_profile_ms()
        if prof_ms is not None:
            roof["kernel_profile_ms"] = prof_ms
            roof["kernel_profile_source"] = f"{prof_path} (rocprofv3 --kernel-trace --stats of this command, committed)"
        am = np.unravel_index(np.nanargmax(corr), corr.shape)
        out = {
            "metric": METRIC,
            "value": None if dry else L * args.steps / elapsed, "unit": "lag-points/s", "n_gpus": world,
            "n_ranks_seen": n_ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "headline: Carrington 'fa' 2048x2048 grid lon(200,300) lat(-20,20), 60x60 CRVAL "
                                   "lags arange(-30,30,1) arcsec, small 2048^2 HRIEUV-like, ref 3072^2 FSI-like, "
                                   "order 2, solar_r 1.004",
                       "lag_points": L, "grid": list(GRID_SHAPE),
                       "parallelism": ("single GPU, no collective" if world == 1 and not use_dist else
                                       f"grid point shares x{world} + 1 all-reduce of the six sums per lag ({backend})"
                                       if by_points else f"lag-plane blocks x{world} + 1 all-gather ({backend})"),
                       "resident": True, "sweeps_in_flight": n_streams,
                       "small_stored_f32": bool(stats["small_is_f32"]), "use_lds": bool(stats["used_lds"]),
                       "nan_pixel_fraction": args.nan_frac,
                       "lag_sharding": "points" if by_points else ("blocks" if world > 1 else "none")},
            "one_sweep_in_flight": None if one_in_flight is None else
            {"value": L / one_in_flight, "unit": "lag-points/s", "ms_per_step": 1e3 * one_in_flight},
            # the rate that goes with roofline.kernel_ms / kernel_profile_ms (single, non-overlapped launches): `value`
            # has two sweeps in flight, whose ramp-up and drain overlap
            "value_consistent_with_kernel_profile": None if one_in_flight is None else L / one_in_flight,
            "sustained": sustained,
            "roofline": roof,
            "per_rank": per_rank,
            "precompute_ms": float(np.mean(pre_ms)),
            "pcie_inclusive": pcie,
            "all_finite_image": all_finite,
            "map_vs_single_gpu": vs_single,
            "argmax_lag_arcsec": [float(lag1[am[0]]), float(lag2[am[1]])],
            "injected_shift_arcsec": [truth["lag_crval1"], truth["lag_crval2"]],
        }
        if dry:
            out["dry_run"] = True
            out["dry_run_map_ok"] = bool(np.array_equal(corr.ravel(), np.arange(L, dtype=np.float64)))
            # what the CPU parity sample of a real run does on rank 0: a forked worker pool, ended with SIGTERM by
            # multiprocessing -- which must not reach this rank's SIGTERM watcher
            import multiprocessing as mp
            with mp.get_context("fork").Pool(2) as pool:
                out["dry_run_pool_ok"] = pool.map(abs, [-1, -2]) == [1, 2]
            time.sleep(0.3)
        if not args.no_cpu_baseline and world > 1 and not dry:
            # N > 1: no CPU timing (that leg belongs to the N = 1 line), but the assembled map is still checked against
            # the oracle on a seeded sample of 128 lag-points
            log("[bench] CPU parity sample: 128 lag-points ...")
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            cores = max(1, min(avail // max(world, 1), 8))
            _, corr_cpu, subset = cpu_baseline(np, small, hs, large, hl, lags, 128, cores)
            d = np.abs(corr.ravel()[subset] - corr_cpu.ravel()[subset])
            out["parity_vs_cpu_sample"] = {"max_abs_dcorr": float(np.nanmax(d)), "n": int(len(subset)),
                                           "argmax_on_sample_equal": bool(np.nanargmax(corr.ravel()[subset]) ==
                                                                          np.nanargmax(corr_cpu.ravel()[subset]))}
            out["cpu_baseline"] = None
        elif not args.no_cpu_baseline and world == 1 and not dry:  # the CPU leg is timed at N = 1 only
            # the GPU box gives one GPU's job a 16-core share whatever os.cpu_count() says
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            cores = int(os.environ.get("COREG_CPU_CORES", min(avail, 16)))
            n_sample = args.cpu_sample or 48 * cores
            log(f"[bench] CPU baseline: {n_sample} lag-points on {cores} cores ...")
            cb, corr_cpu, subset = cpu_baseline(np, small, hs, large, hl, lags, n_sample, cores)
            out["cpu_baseline"] = cb
            d = np.abs(corr.ravel()[subset] - corr_cpu.ravel()[subset])
            out["parity_vs_cpu_sample"] = {"max_abs_dcorr": float(np.nanmax(d)), "n": int(len(subset)),
                                           "argmax_on_sample_equal": bool(np.nanargmax(corr.ravel()[subset]) ==
                                                                          np.nanargmax(corr_cpu.ravel()[subset]))}
        else:
            out["cpu_baseline"] = None
        if not dry:
            out["parity_vs_reference_run"] = reference_run_points(np, small, large, corr)
        print_line_once(json.dumps(out))
    for hk in handles:
        hk.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:  # a rank that fails says so on the ONE line (rank 0) and ends non-zero
        import traceback
        traceback.print_exc()
        if int(os.environ.get("RANK", "0")) == 0:
            print_line_once(error_line(f"{type(e).__name__}: {e}"))
        sys.stderr.flush()
        os._exit(1)  # (not sys.exit: a process group stuck in a collective may never let the interpreter finalise)
