#!/usr/bin/env python3
"""Emulate on ONE GPU what the busiest rank of an N-GPU sweep does under the two partitions of a multi-dimensional lag
set (BASELINE cfg3 / cfg4 / cfg5 at their stated sizes):
  blocks  -- the (CRVAL1, CRVAL2) plane cut in N blocks, every rank runs ALL (cdelt1, cdelt2, crota) combinations
             (rounds 1-3; `parallel.block_bounds`),
  planner -- what `parallel.lag_plan` picks now: for 3-D / 5-D sweeps the combinations are dealt to the ranks
             (`parallel.grid_share`), each sweeps the whole plane (or a block of it) for its run of combinations.
Rank 0 holds the largest share in both.  Times are whole calls through the C ABI (map back on the host), before the
all-gather of the per-lag coefficients; efficiency = T(1) / (N * T(N)).
usage: python profiles/partition_timing.py [cfg3] [cfg4] [cfg5]        -> one JSON line per (config, N, partition)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euispice_coreg_amd import _lib, parallel, synthetic  # noqa: E402


def share_call(h, sweep, lags, share):
    lo1, hi1, lo2, hi2, c_lo, c_hi = share
    a = lags.arrays
    sub = _lib.LagSet(a[0][lo1:hi1], a[1][lo2:hi2], a[2], a[3], a[4])
    inner = lags.shape[2] * lags.shape[3] * lags.shape[4]
    n = sub.shape[0] * sub.shape[1] * (c_hi - c_lo)

    def call():
        if (c_lo, c_hi) != (0, inner):
            h.set_option("combo_begin", c_lo)
            h.set_option("combo_end", c_hi)
        return sweep(sub, n)
    return call, n


def timed(fn, reps):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    which = sys.argv[1:] or ["cfg3", "cfg4", "cfg5"]
    h = _lib.CoregHandle(0)
    small, hs, large, hl, _ = synthetic.make_scene()
    for name in which:
        if name == "cfg3":
            grid = _lib.Grid((200, 300), (-20, 20), (2048, 2048))
            h.set_small(small)
            h.prepare_reference_carrington(large, hl, grid, 1.004, 2)
            lags = _lib.LagSet(np.arange(-60, 61, 1.0), np.arange(-60, 61, 1.0), None, None, None)
            sweep = lambda sub, n: h.sweep_carrington(hs, grid, 1.004, sub, lag_end=n)  # noqa: E731
            reps = 3
        elif name == "cfg5":
            grid = _lib.Grid((200, 300), (-20, 20), (4096, 4096))
            h.set_small(small)
            h.prepare_reference_carrington(large, hl, grid, 1.004, 2)
            lags = _lib.LagSet(np.arange(-20, 21, 1.0), np.arange(-20, 21, 1.0), np.arange(-2, 3) * 0.01,
                               np.arange(-2, 3) * 0.01, np.arange(-5, 6) * 0.1)
            sweep = lambda sub, n: h.sweep_carrington(hs, grid, 1.004, sub, lag_end=n)  # noqa: E731
            reps = 1
        else:
            s4, hs4, l4, hl4, _ = synthetic.make_scene(small_shape=(832, 192), small_cdelt=(4.0, 1.098),
                                                       small_unit="deg", large_n=3072)
            h.set_small(s4)
            h.prepare_reference_helioprojective(l4, hl4, hs4, 2)
            lags = _lib.LagSet(np.arange(-30, 31, 1.0) / 3600, np.arange(-30, 31, 1.0) / 3600, None, None,
                               np.arange(-10, 11) * 0.1)
            sweep = lambda sub, n: h.sweep_helioprojective(hs4, hs4, sub, lag_end=n)  # noqa: E731
            reps = 2
        inner = lags.shape[2] * lags.shape[3] * lags.shape[4]
        per_combo = name != "cfg4"  # (the helioprojective sweep runs one launch whatever the lag set)
        t1 = None
        for world in (1, 2, 4, 8):
            plan = parallel.lag_plan(lags.shape, world, per_combo)
            shares = {"planner": parallel.grid_share(lags.shape, world, 0, per_combo) if plan[0] in ("blocks", "combos") else None,
                      "blocks": parallel.block_bounds(lags.shape[0], lags.shape[1], world, 0) + (0, inner),
                      # (what the planner would pick if every combination were a launch of its own)
                      "combos": parallel.grid_share(lags.shape, world, 0, True)
                      if parallel.lag_plan(lags.shape, world, True)[0] == "combos" else None}
            if world == 1:
                shares = {"planner": (0, lags.shape[0], 0, lags.shape[1], 0, inner)}
            for part, share in shares.items():
                if share is None or (part != "planner" and world > 1 and share == shares["planner"]):
                    continue  # (the planner's choice IS this partition: timed once)
                call, n = share_call(h, sweep, lags, share)
                dt = timed(call, reps)
                st = h.last_stats()
                if world == 1:
                    t1 = dt
                print(json.dumps({"config": name, "n_gpus": world, "partition": part,
                                  "plan": list(plan) if part == "planner" else (
                                      list(parallel.lag_plan(lags.shape, world, True)) if part == "combos" else
                                      ["blocks", 1] + list(parallel.block_grid(lags.shape[0], lags.shape[1], world))),
                                  "rank0_lag_points": n, "rank0_ms": round(dt * 1e3, 3),
                                  "rank0_sweep_launches": st["n_sweep_launches"],
                                  "rank0_kernel_ms": round(st["sweep_kernel_ms"], 3),
                                  "rank0_precompute_ms": round(st["precompute_ms"], 3),
                                  "speedup_over_one_gpu": round(t1 / dt, 2),
                                  "efficiency": round(t1 / dt / world, 3)}), flush=True)
    h.close()


if __name__ == "__main__":
    main()
