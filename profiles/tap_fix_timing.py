#!/usr/bin/env python3
"""What the general noise-decided-sample pass of odd spline orders costs ("tap_fix": k_tap_scan before the sweep, wcslib's
chain on the host for the listed samples, k_tap_fix after it) on cfg2 -- 2048^2 image against the sub-map of a 3072^2
reference, helioprojective, 61 x 61 CRVAL lags -- for orders 1 and 3, with the scene's header as it is and with a roll of
3 degrees (an unrotated header brings whole curves of coordinates back onto integers for pure CRVAL1 / CRVAL2 lags).
usage: python profiles/tap_fix_timing.py       -> one JSON line per case"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euispice_coreg_amd import _lib, synthetic  # noqa: E402


def main():
    h = _lib.CoregHandle(0)
    small, hs, large, hl, _ = synthetic.make_scene()
    lags = _lib.LagSet(np.arange(-30, 31, 1.0), np.arange(-30, 31, 1.0), None, None, None)
    for roll in (0.0, 3.0):
        hdr = dict(hs)
        if True:
            rho, lam = np.deg2rad(roll), hdr["CDELT2"] / hdr["CDELT1"]
            hdr.update(CROTA=roll, PC1_1=float(np.cos(rho)), PC2_2=float(np.cos(rho)), PC1_2=float(-lam * np.sin(rho)),
                       PC2_1=float(np.sin(rho) / lam))
        for order in (1, 3, 2):
            h.set_small(small)
            h.prepare_reference_helioprojective(large, hl, hdr, order)
            out = {}
            for fix in (0, 1):
                h.set_option("tap_fix", fix)
                h.sweep_helioprojective(hdr, hdr, lags, order=order)
                best = 1e9
                for _ in range(2):
                    t0 = time.perf_counter()
                    out[fix] = h.sweep_helioprojective(hdr, hdr, lags, order=order)
                    best = min(best, time.perf_counter() - t0)
                res = {"header_crota": hdr.get("CROTA", 0.0), "order": order, "tap_fix": fix, "sweep_ms": round(best * 1e3, 2),
                       "kernel_ms": round(h.last_stats()["sweep_kernel_ms"], 2)}
                if fix:
                    res.update(h.last_tap_fix())
                    d = np.abs(out[1] - out[0])
                    res["lag_points_changed"] = int((d > 0).sum())
                    res["largest_change"] = float(np.nanmax(d))
                    # round 5: only the samples that can change the result are listed (on the bounds rule, or with a
                    # non-finite pixel in the union of the two tap sets); the unfiltered list gives the same map
                    h.set_option("tap_nan_filter", 0)
                    h.sweep_helioprojective(hdr, hdr, lags, order=order)
                    t0 = time.perf_counter()
                    full = h.sweep_helioprojective(hdr, hdr, lags, order=order)
                    res["unfiltered_sweep_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
                    res["unfiltered_samples"] = h.last_tap_fix()["samples"] if "samples" in h.last_tap_fix() else h.last_tap_fix()
                    res["filtered_vs_unfiltered_max"] = float(np.nanmax(np.abs(full - out[1])))
                    # the round's first filter (a non-finite pixel anywhere in the union of the two footprints), for
                    # comparison with the end-line test that is the default now
                    h.set_option("tap_nan_filter", 1)
                    h.sweep_helioprojective(hdr, hdr, lags, order=order)
                    t0 = time.perf_counter()
                    union = h.sweep_helioprojective(hdr, hdr, lags, order=order)
                    res["union_filter_sweep_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
                    res["union_filter_samples"] = h.last_tap_fix()["samples"]
                    res["union_filter_vs_default_max"] = float(np.nanmax(np.abs(union - out[1])))
                    h.set_option("tap_nan_filter", 2)
                print(json.dumps(res), flush=True)
    # serial semantics (parallelism=False: the target is the reference's own, coarser grid -- every 64-pixel segment of
    # the scan spans many integers and is tested pixel by pixel)
    hdr = dict(hs)
    h.set_small(small)
    h.set_reference_on_grid(np.asarray(large, dtype=np.float64))
    lags21 = _lib.LagSet(np.arange(-10, 11, 1.0), np.arange(-10, 11, 1.0), None, None, None)
    for order in (1, 3):
        for fix in (0, 1):
            h.set_option("tap_fix", fix)
            h.sweep_helioprojective(hl, hdr, lags21, order=order)
            t0 = time.perf_counter()
            h.sweep_helioprojective(hl, hdr, lags21, order=order)
            res = {"semantics": "serial (target = the 3072^2 reference grid)", "lags": lags21.size, "order": order, "tap_fix": fix,
                   "sweep_ms": round((time.perf_counter() - t0) * 1e3, 2), "kernel_ms": round(h.last_stats()["sweep_kernel_ms"], 2)}
            if fix:
                res.update(h.last_tap_fix())
            print(json.dumps(res), flush=True)
    h.close()


if __name__ == "__main__":
    main()
