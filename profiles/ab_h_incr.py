#!/usr/bin/env python3
"""A/B of the incremental homography path (option "h_incr", kernels.hpp tile_points kIncr): cfg2 and cfg4 with the
option off and on -- kernel ms, ps per (point x lag), max |difference| of the maps.  One GPU.
usage: python profiles/ab_h_incr.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euispice_coreg_amd import _lib, synthetic  # noqa: E402


def run(h, fn, lags, reps=5):
    fn()
    ks = []
    for _ in range(reps):
        out = fn()
        ks.append(h.last_stats()["sweep_kernel_ms"])
    st = h.last_stats()
    return np.asarray(out), dict(kernel_ms=round(min(ks), 3), kernel_ms_mean=round(float(np.mean(ks)), 3),
                                 ps_per_point_lag=round(min(ks) * 1e9 / (st["n_active_points"] * lags.size), 4),
                                 visits=h.last_visit_counts())


def main():
    h = _lib.CoregHandle(0)
    small, hs, large, hl, _ = synthetic.make_scene()
    s4, hs4, l4, hl4, _ = synthetic.make_scene(small_shape=(832, 192), small_cdelt=(4.0, 1.098), small_unit="deg", large_n=3072)
    cases = {
        "cfg2": (small, hs, large, hl, _lib.LagSet(np.arange(-30, 31, 1.0), np.arange(-30, 31, 1.0), None, None, None)),
        "cfg4": (s4, hs4, l4, hl4, _lib.LagSet(np.arange(-30, 31, 1.0) / 3600, np.arange(-30, 31, 1.0) / 3600, None, None,
                                               np.round(np.arange(-10, 11) * 0.1, 10))),
    }
    for name, (sm, h_s, lg, h_l, lags) in cases.items():
        h.set_small(sm)
        h.prepare_reference_helioprojective(lg, h_l, h_s, 2)
        maps = {}
        for series in ((1, 0) if name == "cfg2" else (1,)):
            h.set_option("h_series", series)
            for incr in (0, 1, 0, 1):
                h.set_option("h_incr", incr)
                m, r = run(h, lambda: h.sweep_helioprojective(h_s, h_s, lags), lags)
                maps[(series, incr)] = m
                print(name, json.dumps(dict(h_series=series, h_incr=incr, **r)), flush=True)
            d = np.nanmax(np.abs(maps[(series, 1)] - maps[(series, 0)]))
            print(name, f"h_series={series}: max |map(h_incr=1) - map(h_incr=0)| = {d:.3e}", flush=True)
        h.set_option("h_series", 1)
    h.close()


if __name__ == "__main__":
    main()
