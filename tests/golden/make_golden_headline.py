#!/usr/bin/env python3
"""
Headline sample: 256 seeded lag-points of the 60 x 60 headline map (2048^2 Carrington grid, lon (200, 300), lat (-20, 20),
lags arange(-30, 30, 1) arcsec, solar_r 1.004, order 2) evaluated by the ORACLE (oracle/coreg_oracle.py, the CPU
restatement of alignment.py:613-797 pinned against the reference's own output by tests/test_reference_golden_cpu.py) on
the synthetic scene `euispice_coreg_amd.synthetic.make_scene()`.  About two minutes on 8 cores.

    python tests/golden/make_golden_headline.py

Stored: the raveled lag indices, the oracle's coefficients, and a fingerprint of the scene (so that a test that
regenerates the scene on another box can tell that it holds the same pixels).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from euispice_coreg_amd import synthetic  # noqa: E402
from oracle import coreg_oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402

LON, LAT, SHAPE = (200.0, 300.0), (-20.0, 20.0), (2048, 2048)


def fingerprint(small, large):
    return np.array([np.nansum(small), np.nansum(large), float(np.isnan(small).sum()), small[1000, 1000], large[1500, 1500]])


def main():
    small, hs, large, hl, truth = synthetic.make_scene()
    lag = np.arange(-30.0, 30.0, 1.0)
    st = H.oracle_state(small, hs, large, hl, (lag, lag, None, None, None), order=2, shape=list(SHAPE), lonlims=list(LON),
                        latlims=list(LAT), solar_r=(1.004,))
    rng = np.random.default_rng(20261004)
    idx = np.sort(rng.choice(3600, size=256, replace=False))
    corr = O.find_best_header_parameters(st, "carrington", counts=os.cpu_count(), lag_subset=idx)
    vals = corr.reshape(-1)[idx]
    assert np.isfinite(vals).all()
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "headline_sample.npz")
    np.savez(dst, index=idx, corr=vals, fingerprint=fingerprint(small, large))
    print("wrote", dst, "max", vals.max(), "at", idx[np.argmax(vals)])


def dump_scene(path):
    """The headline scene as a BITPIX = -32 FITS pair holds it, for make_golden_headline_reference.py (the reference's
    own run of a sub-lattice of the 60 x 60 lags)."""
    import json
    small, hs, large, hl, truth = synthetic.make_scene()
    s32, l32 = small.astype(np.float32), large.astype(np.float32)
    assert np.array_equal(s32.astype(np.float64), small, equal_nan=True) and np.array_equal(l32.astype(np.float64), large)
    np.savez(path, small=s32, large=l32, hdr_small=np.array(json.dumps(hs)), hdr_large=np.array(json.dumps(hl)),
             fingerprint=fingerprint(small, large))
    print("wrote", path)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--dump-scene":
        dump_scene(sys.argv[2])
        sys.exit(0)
    main()
