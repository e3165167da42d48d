#!/usr/bin/env python3
"""Single-GPU timings of the other BASELINE.json configs (not the driver's bench line; see bench.py for that).
usage: python profiles/bench_configs.py [cfg2] [cfg3] [cfg4] [cfg5] [dense]"""
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


def main():
    which = sys.argv[1:] or ["cfg2", "cfg3", "cfg4", "cfg5", "dense"]
    h = _lib.CoregHandle(0)
    for kv in filter(None, os.environ.get("COREG_BENCH_OPTS", "").split(",")):  # tuning experiments: "opt=val,opt=val"
        k, v = kv.split("=")
        h.set_option(k, int(v))
    small, hs, large, hl, truth = synthetic.make_scene()
    res = {}
    if "cfg2" in which:  # helioprojective, sub-map semantics, 61 x 61
        lags = _lib.LagSet(np.arange(-30, 31, 1.0), np.arange(-30, 31, 1.0), None, None, None)
        h.set_small(small)
        h.prepare_reference_helioprojective(large, hl, hs, 2)
        dt, out = timed(lambda: h.sweep_helioprojective(hs, hs, lags))
        st = h.last_stats()
        am = np.unravel_index(np.nanargmax(out), lags.shape)
        res["cfg2"] = dict(lags=lags.size, ms=dt * 1e3, lags_per_s=lags.size / dt, kernel_ms=st["sweep_kernel_ms"],
                           active=st["n_active_points"], argmax=[float(lags.arrays[0][am[0]]), float(lags.arrays[1][am[1]])])
    if "cfg3" in which or "cfg5" in which or "dense" in which:
        h.set_small(small)
    if "cfg3" in which:  # Carrington 2048^2, 121 x 121
        grid = _lib.Grid((200, 300), (-20, 20), (2048, 2048))
        h.prepare_reference_carrington(large, hl, grid, 1.004, 2)
        lags = _lib.LagSet(np.arange(-60, 61, 1.0), np.arange(-60, 61, 1.0), None, None, None)
        dt, out = timed(lambda: h.sweep_carrington(hs, grid, 1.004, lags))
        st = h.last_stats()
        am = np.unravel_index(np.nanargmax(out), lags.shape)
        res["cfg3"] = dict(lags=lags.size, ms=dt * 1e3, lags_per_s=lags.size / dt, kernel_ms=st["sweep_kernel_ms"],
                           active=st["n_active_points"], argmax=[float(lags.arrays[0][am[0]]), float(lags.arrays[1][am[1]])])
    if "dense" in which:  # headline lags on a grid tightened onto the small FOV (near-full overlap)
        grid = _lib.Grid((228, 262), (-12, 22), (2048, 2048))
        h.prepare_reference_carrington(large, hl, grid, 1.004, 2)
        lags = _lib.LagSet(np.arange(-30, 30, 1.0), np.arange(-30, 30, 1.0), None, None, None)
        dt, out = timed(lambda: h.sweep_carrington(hs, grid, 1.004, lags))
        st = h.last_stats()
        res["dense"] = dict(lags=lags.size, ms=dt * 1e3, lags_per_s=lags.size / dt, kernel_ms=st["sweep_kernel_ms"],
                            active=st["n_active_points"], grid_points=st["n_grid_points"])
    if "cfg5" in which:  # 5-D on a 4096^2 Carrington grid
        grid = _lib.Grid((200, 300), (-20, 20), (4096, 4096))
        h.prepare_reference_carrington(large, hl, grid, 1.004, 2)
        lags = _lib.LagSet(np.arange(-20, 21, 1.0), np.arange(-20, 21, 1.0), np.arange(-2, 3) * 0.01,
                           np.arange(-2, 3) * 0.01, np.arange(-5, 6) * 0.1)
        dt, out = timed(lambda: h.sweep_carrington(hs, grid, 1.004, lags), reps=1)
        st = h.last_stats()
        am = np.unravel_index(np.nanargmax(out), lags.shape)
        res["cfg5"] = dict(lags=lags.size, ms=dt * 1e3, lags_per_s=lags.size / dt, kernel_ms=st["sweep_kernel_ms"],
                           precompute_ms=st["precompute_ms"], launches=st["n_sweep_launches"],
                           argmax=[int(i) for i in am])
    if "cfg4" in which:  # SPICE-like raster 192 x 832, header in degrees, 61 x 61 x 21
        s4, hs4, l4, hl4, _ = synthetic.make_scene(small_shape=(832, 192), small_cdelt=(4.0, 1.098), small_unit="deg",
                                                   large_n=3072)
        h.set_small(s4)
        h.prepare_reference_helioprojective(l4, hl4, hs4, 2)
        lags = _lib.LagSet(np.arange(-30, 31, 1.0) / 3600, np.arange(-30, 31, 1.0) / 3600, None, None,
                           np.arange(-10, 11) * 0.1)
        dt, out = timed(lambda: h.sweep_helioprojective(hs4, hs4, lags), reps=2)
        st = h.last_stats()
        am = np.unravel_index(np.nanargmax(out), lags.shape)
        res["cfg4"] = dict(lags=lags.size, ms=dt * 1e3, lags_per_s=lags.size / dt, kernel_ms=st["sweep_kernel_ms"],
                           active=st["n_active_points"], argmax=[int(i) for i in am])
    for k, v in res.items():
        print(k, json.dumps(v), flush=True)
    h.close()


if __name__ == "__main__":
    main()
