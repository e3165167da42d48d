#!/usr/bin/env python3
"""
cfg5 sample: 75 seeded lag-points of BASELINE configs[4] -- 4096^2 Carrington grid, lon (200, 300), lat (-20, 20), lags
crval1/2 arange(-20, 21), cdelt1/2 [-0.02 .. 0.02] step 0.01, crota [-0.5 .. 0.5] step 0.1 (41 x 41 x 5 x 5 x 11 =
462 275 lag-points) -- THREE IN EACH OF THE 25 (d_cdelt1, d_cdelt2) PLANES, evaluated by the ORACLE
(oracle/coreg_oracle.py, intended CDELT semantics, itself held to the reference's own code on that slice by
tests/test_reference_cdelt_intended_cpu.py and cdelt_intended_cfg5.npz) on `euispice_coreg_amd.synthetic.make_scene()`.
A few minutes on 8 cores.

    python tests/golden/make_golden_cfg5_sample.py

Stored: the raveled lag indices, the oracle's coefficients, the scene fingerprint (as headline_sample.npz).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from euispice_coreg_amd import synthetic  # noqa: E402
from oracle import coreg_oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.golden.make_golden_headline import fingerprint  # noqa: E402

LON, LAT, SHAPE = (200.0, 300.0), (-20.0, 20.0), (4096, 4096)


def axes():
    l1 = np.arange(-20.0, 21.0, 1.0)
    lc = np.round(np.arange(-2, 3) * 0.01, 10)
    lr = np.round(np.arange(-5, 6) * 0.1, 10)
    return l1, l1, lc, lc, lr


def main():
    small, hs, large, hl, truth = synthetic.make_scene()
    ax = axes()
    dims = tuple(len(a) for a in ax)
    st = H.oracle_state(small, hs, large, hl, ax, order=2, shape=list(SHAPE), lonlims=list(LON), latlims=list(LAT),
                        solar_r=(1.004,))
    rng = np.random.default_rng(20261005)
    pts = []
    for i3 in range(5):
        for i4 in range(5):
            # one near the injected shift (17, -9, +0.3), two anywhere
            pts.append((37 + int(rng.integers(-2, 3)), 11 + int(rng.integers(-2, 3)), i3, i4, 8 + int(rng.integers(-1, 2))))
            for _ in range(2):
                pts.append((int(rng.integers(0, 41)), int(rng.integers(0, 41)), i3, i4, int(rng.integers(0, 11))))
    idx = np.unique(np.ravel_multi_index(tuple(np.array(pts).T), dims))
    assert idx.size == 75
    # + the 38 lag-points the reference's own code evaluated (cdelt_intended_cfg5.npz): the oracle's value there is stored
    # beside them, so that the CPU suite can hold the two together without minutes of recomputation
    here = os.path.dirname(os.path.abspath(__file__))
    ref = np.load(os.path.join(here, "cdelt_intended_cfg5.npz"))
    assert np.array_equal(ref["fingerprint"], fingerprint(small, large))
    ridx = np.ravel_multi_index(tuple(ref["index"].T), dims)
    both = np.unique(np.concatenate([idx, ridx]))
    corr = O.find_best_header_parameters(st, "carrington", counts=os.cpu_count(), lag_subset=both).reshape(-1)
    vals = corr[idx]
    assert np.isfinite(vals).all()
    at_ref = corr[ridx]
    print("oracle vs the reference's own code on", ridx.size, "cfg5 lag-points: max |dcorr|", np.abs(at_ref - ref["corr"]).max())
    dst = os.path.join(here, "cfg5_sample.npz")
    np.savez(dst, index=idx, corr=vals, fingerprint=fingerprint(small, large), reference_index=ridx,
             oracle_at_reference=at_ref)
    print("wrote", dst, "max", vals.max(), "at", np.unravel_index(idx[np.argmax(vals)], dims))


if __name__ == "__main__":
    main()
