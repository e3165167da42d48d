#!/usr/bin/env python3
"""Timings of the sweep variants other than the headline one (helioprojective series / exact division, crota axis,
plate-carree rotation mode, run-time spline orders), one GPU.  Prints ps per (active grid point x lag) so that the
variants can be compared with the TRANSLATE order-2 kernel.
usage: python profiles/bench_paths.py [cfg2] [cfg4] [car] [order3] [order1] [headline]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euispice_coreg_amd import _lib, synthetic  # noqa: E402


def timed(fn, reps=3):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        best = min(best, time.perf_counter() - t0)
    return best, out


def report(name, h, lags, dt, out, extra=None):
    st = h.last_stats()
    n = lags.size
    r = dict(lags=n, ms=round(dt * 1e3, 3), lags_per_s=round(n / dt), kernel_ms=round(st["sweep_kernel_ms"], 3),
             precompute_ms=round(st["precompute_ms"], 3), launches=st["n_sweep_launches"],
             active=st["n_active_points"],
             ps_per_point_lag=round(st["sweep_kernel_ms"] * 1e9 / max(1, st["n_active_points"] * n), 3),
             argmax=[int(i) for i in np.unravel_index(np.nanargmax(out), lags.shape)])
    if extra:
        r.update(extra)
    print(name, json.dumps(r), flush=True)


def main():
    which = sys.argv[1:] or ["headline", "cfg2", "cfg4", "car", "order3", "order1"]
    h = _lib.CoregHandle(0)
    for kv in filter(None, os.environ.get("COREG_BENCH_OPTS", "").split(",")):
        k, v = kv.split("=")
        h.set_option(k, int(v))
    small, hs, large, hl, truth = synthetic.make_scene()
    grid = _lib.Grid((200, 300), (-20, 20), (2048, 2048))
    l60 = _lib.LagSet(np.arange(-30, 30, 1.0), np.arange(-30, 30, 1.0), None, None, None)
    if "headline" in which or "order3" in which or "order1" in which:
        h.set_small(small)
    for name, order in (("headline", 2), ("order1", 1), ("order3", 3)):
        if name in which:
            h.prepare_reference_carrington(large, hl, grid, 1.004, order)
            dt, out = timed(lambda: h.sweep_carrington(hs, grid, 1.004, l60, order=order))
            report(name, h, l60, dt, out)
    if "cfg2" in which:
        lags = _lib.LagSet(np.arange(-30, 31, 1.0), np.arange(-30, 31, 1.0), None, None, None)
        h.set_small(small)
        h.prepare_reference_helioprojective(large, hl, hs, 2)
        dt, out = timed(lambda: h.sweep_helioprojective(hs, hs, lags))
        report("cfg2", h, lags, dt, out)
        h.set_option("h_series", 0)
        dt, out = timed(lambda: h.sweep_helioprojective(hs, hs, lags))
        h.set_option("h_series", 1)
        report("cfg2_exact_division", h, lags, dt, out)
    if "cfg4" in which:
        s4, hs4, l4, hl4, _ = synthetic.make_scene(small_shape=(832, 192), small_cdelt=(4.0, 1.098), small_unit="deg",
                                                   large_n=3072)
        h.set_small(s4)
        h.prepare_reference_helioprojective(l4, hl4, hs4, 2)
        lags = _lib.LagSet(np.arange(-30, 31, 1.0) / 3600, np.arange(-30, 31, 1.0) / 3600, None, None,
                           np.arange(-10, 11) * 0.1)
        dt, out = timed(lambda: h.sweep_helioprojective(hs4, hs4, lags), reps=2)
        report("cfg4", h, lags, dt, out)
    if "car" in which:
        sc, hsc, lc, hlc, _ = synthetic.make_car_scene(small_shape=(768, 1024), large_shape=(1200, 1600), n_blobs=60)
        h.set_small(sc)
        h.set_reference_on_grid(np.asarray(lc, dtype=np.float32))
        lags = _lib.LagSet(np.arange(-15, 16) * 0.004, np.arange(-15, 16) * 0.004, None, None, None)
        dt, out = timed(lambda: h.sweep_helioprojective(hlc, hsc, lags), reps=2)
        report("car", h, lags, dt, out)
    h.close()


if __name__ == "__main__":
    main()
