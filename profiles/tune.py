#!/usr/bin/env python3
"""Tuning harness: headline sweep (bench.py workload) under different library options; prints sweep-kernel ms.
usage: python profiles/tune.py "opt=val,opt=val" "opt=val" ...   (each argument = one configuration)
NOTE: the FIRST configuration timed in a process runs 5-8 % slower than the following ones (clock ramp): repeat the
baseline after the candidates before believing a difference."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euispice_coreg_amd import _lib, synthetic  # noqa: E402


def main():
    # TUNE_NAN: fraction of NaN pixels in the image to align (bench.py's scene: 0.005; 0 = every LDS window all-finite)
    small, hs, large, hl, truth = synthetic.make_scene(nan_frac=float(os.environ.get("TUNE_NAN", "0.005")))
    n = int(os.environ.get("TUNE_NLAG", "60"))
    lag = (np.arange(n) - n // 2).astype(np.float64)
    lags = _lib.LagSet(lag, lag, None, None, None)
    shape = (int(os.environ.get("TUNE_GRID", "2048")),) * 2
    lon = tuple(float(x) for x in os.environ.get("TUNE_LON", "200,300").split(","))
    lat = tuple(float(x) for x in os.environ.get("TUNE_LAT", "-20,20").split(","))
    grid = _lib.Grid(lon, lat, shape)
    h = _lib.CoregHandle(0)
    h.set_small(small)
    h.prepare_reference_carrington(large, hl, grid, 1.004, 2)
    base = None
    for cfg in (sys.argv[1:] or [""]):
        defaults = {"use_lds": 1, "clean_path": 1, "tile_w": 0, "n_groups": 0, "lds_bytes": int(os.environ.get("TUNE_LDS", 159 * 1024)), "patch_w": 0,
                    "pitch": -1, "taper_frac": -1, "taper_min": 128, "taper_rounds": 6}
        for kv in filter(None, cfg.split(",")):
            k, v = kv.split("=")
            defaults[k] = int(v)
        for k, v in defaults.items():
            h.set_option(k, v)
        ms = []
        for it in range(6):
            t0 = time.perf_counter()
            out = h.sweep_carrington(hs, grid, 1.004, lags)
            wall = time.perf_counter() - t0
            st = h.last_stats()
            if it >= 1:
                ms.append((st["sweep_kernel_ms"], st["precompute_ms"], st["total_gpu_ms"], wall * 1e3))
        ms = np.array(ms)
        if base is None:
            base = out
        am = np.unravel_index(np.nanargmax(out), (n, n)) if np.isfinite(out).any() else (0, 0)
        print(f"{cfg or 'default':40s} kernel {ms[:,0].min():7.3f} ms  pre {ms[:,1].min():6.3f}  gpu {ms[:,2].min():7.3f}  "
              f"wall {ms[:,3].min():7.3f}  argmax {lag[am[0]]:.0f},{lag[am[1]]:.0f}  maxdiff {np.nanmax(np.abs(out-base)) if np.isfinite(out).any() else -1:.1e}  "
              f"active {st['n_active_points']}  visits {h.last_visit_counts()}", flush=True)
    h.close()


if __name__ == "__main__":
    main()
