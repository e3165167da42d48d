#!/usr/bin/env python3
"""
Golden correlation maps for BASELINE.json configs[0]: synthetic 512x512 small-FOV vs 1024x1024 large-FOV,
helioprojective, lag_crval1/2 in [-5, 5] step 1 arcsec, crota/cdelt fixed, parallelism=False CPU path (full large
grid, float64 reference -- quirk Q1) and, for comparison, the parallelism=True semantics (sub-map, float32 reference).
Two lag windows: BASELINE's own [-5, 5] (arrays `*0`; it contains the ZERO lag, whose border pixels on the sub-map
path are decided by wcslib's rounding noise -- oracle: WcslibTan, pinned by border_golden.npz) and the same window
centred on the injected shift (17, -9), where the correlation peak is.  Produced by the CPU oracle
(oracle/coreg_oracle.py, itself pinned by rectify_golden.npz / wcs_golden.npz); inputs are regenerated from the seed.

    python tests/golden/make_golden_cfg1.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from euispice_coreg_amd import synthetic  # noqa: E402
from tests import helpers as H  # noqa: E402

SEED = 20220317


def scene():
    return synthetic.make_scene(small_n=512, large_n=1024, seed=SEED)


def lags(truth):
    return (truth["lag_crval1"] + np.arange(-5, 6, 1.0), truth["lag_crval2"] + np.arange(-5, 6, 1.0), None, None, None)


def lags_baseline():
    return (np.arange(-5, 6, 1.0), np.arange(-5, 6, 1.0), None, None, None)


def fingerprint(small32, large32):
    return np.array([np.nansum(small32.astype(np.float64)), np.nansum(large32.astype(np.float64)),
                     float(np.isnan(small32).sum()), float(small32[200, 300]), float(large32[700, 500])])


def dump_scene(path):
    """The scene as a BITPIX = -32 FITS pair holds it, for make_golden_cfg1_reference.py (the reference's own run)."""
    import json
    small, hs, large, hl, truth = scene()
    s32, l32 = small.astype(np.float32), large.astype(np.float32)
    np.savez(path, small=s32, large=l32, hdr_small=np.array(json.dumps(hs)), hdr_large=np.array(json.dumps(hl)),
             fingerprint=fingerprint(s32, l32))
    print("wrote", path)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--dump-scene":
        dump_scene(sys.argv[2])
        sys.exit(0)
    small, hs, large, hl, truth = scene()
    lg = lags(truth)
    serial = H.oracle_helio(small, hs, large, hl, lg, parallelism=False, counts=os.cpu_count())
    par = H.oracle_helio(small, hs, large, hl, lg, parallelism=True, counts=os.cpu_count())
    carr = H.oracle_carrington(small, hs, large, hl, lg, (512, 512), (228.0, 262.0), (-12.0, 22.0),
                               counts=os.cpu_count())
    l0 = lags_baseline()
    serial0 = H.oracle_helio(small, hs, large, hl, l0, parallelism=False, counts=os.cpu_count())
    par0 = H.oracle_helio(small, hs, large, hl, l0, parallelism=True, counts=os.cpu_count())
    carr0 = H.oracle_carrington(small, hs, large, hl, l0, (512, 512), (228.0, 262.0), (-12.0, 22.0),
                                counts=os.cpu_count())
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg1_corr.npz")
    np.savez_compressed(dst, serial=serial, parallel=par, carrington=carr, lag_crval1=lg[0], lag_crval2=lg[1],
                        serial0=serial0, parallel0=par0, carrington0=carr0,
                        seed=SEED, small_sum=np.nansum(small), large_sum=np.nansum(large))
    for k, v in (("serial", serial), ("parallel", par), ("carrington", carr), ("serial0", serial0),
                 ("parallel0", par0), ("carrington0", carr0)):
        print(k, v.shape, "argmax", np.unravel_index(np.nanargmax(v), v.shape)[:2], "max", np.nanmax(v))
    print("wrote", dst, os.path.getsize(dst), "bytes")
