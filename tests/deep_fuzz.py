"""Long randomised GPU-vs-oracle parity run (the generator of tests/test_gpu_fuzz.py over many more seeds, plus
method='residus', CDELT2 lags, both CDELT semantics, degrees headers; round 4: the compile-time cubic kernel with
`orders` = 1,2,3, and -- Carrington cases with several (cdelt, crota) combinations -- the sweep stitched from two
runs of combinations, the one-shot "combo_begin" / "combo_end" option of a multi-GPU share).
usage: python tests/deep_fuzz.py [n] [seed0] [scale] [orders, e.g. 1,2,3]"""
import os
import sys
import time


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from euispice_coreg_amd import _lib
    from tests import helpers as H
    from tests.test_gpu_fuzz import _random_case
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # image / grid size multiplier
    orders = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2]
    import numpy as np
    n_split = 0
    h = _lib.CoregHandle(-1)
    bad = 0
    # tile visits of each case's LAST sweep launch; lag-points re-evaluated with centred sums (whole sweeps)
    kinds = {"visits": 0, "lds": 0, "interior": 0, "all_finite": 0, "refined_lag_points": 0, "flagged_not_refined": 0}
    t0 = time.time()
    for seed in range(seed0, seed0 + n):
        small, hs, large, hl, lags, rng = _random_case(seed, scale)
        order = int(rng.choice(orders))
        sem = str(rng.choice(["intended", "reference"]))
        lags = list(lags)
        if rng.integers(0, 3) == 0 and sem == "intended":  # the reference dies on a CDELT2 lag (quirk Q2)
            lags[3] = [0.0, -0.02]
        frame = "carrington" if seed % 2 == 0 else "helio"
        try:
            if frame == "carrington":
                lon0 = float(rng.choice([228.0, 200.0, 150.0]))
                lonlims = (lon0, lon0 + float(rng.choice([34.0, 100.0, 220.0])))
                latlims = (-12.0 - float(rng.choice([0, 60])), 22.0)
                shape = (int(rng.integers(20, 70)) * scale, int(rng.integers(20, 70)) * scale)
                solar_r = float(rng.choice([1.004, 1.0, 1.02]))
                want = H.oracle_carrington(small, hs, large, hl, lags, shape, lonlims, latlims, order=order,
                                           solar_r=(solar_r,), cdelt_semantics=sem)
                got = H.gpu_carrington(h, small, hs, large, hl, lags, shape, lonlims, latlims, order=order,
                                       solar_r=solar_r, cdelt_semantics=0 if sem == "intended" else 1)
                ls = _lib.LagSet(*lags)
                inner = ls.shape[2] * ls.shape[3] * ls.shape[4]
                if inner > 1 and len(orders) > 2:  # the same map from two runs of combinations (a multi-GPU share each)
                    cut = int(rng.integers(1, inner))
                    parts = []
                    for c_lo, c_hi in ((0, cut), (cut, inner)):
                        h.set_option("combo_begin", c_lo)
                        h.set_option("combo_end", c_hi)
                        parts.append(H.gpu_carrington(h, small, hs, large, hl, lags, shape, lonlims, latlims, order=order,
                                                      solar_r=solar_r, cdelt_semantics=0 if sem == "intended" else 1,
                                                      prepare=False, lag_end=ls.shape[0] * ls.shape[1] * (c_hi - c_lo)
                                                      ).reshape(ls.shape[0], ls.shape[1], c_hi - c_lo))
                    stitched = np.concatenate(parts, axis=2).reshape(got.shape)
                    assert np.array_equal(np.isnan(stitched), np.isnan(got)), f"seed={seed}: combo runs, NaN pattern"
                    if np.isfinite(got).any():
                        assert np.nanmax(np.abs(stitched - got)) <= 1e-12, f"seed={seed}: combo runs differ from the whole"
                    n_split += 1
                tol = 1e-9
            else:
                serial = bool(rng.integers(0, 2))
                want = H.oracle_helio(small, hs, large, hl, lags, order=order, parallelism=not serial,
                                      cdelt_semantics=sem)
                got = H.gpu_helio(h, small, hs, large, hl, lags, order=order, serial_semantics=serial,
                                  cdelt_semantics=0 if sem == "intended" else 1)
                tol = 1e-7
            for k, v in h.last_visit_counts().items():
                kinds[k] += v
            H.assert_corr_close(got, want, tol, f"seed={seed} {frame}")
        except AssertionError as e:
            bad += 1
            print(f"FAIL seed={seed} frame={frame} order={order} sem={sem}: {e}", flush=True)
        if (seed - seed0) % 20 == 19:
            print(f"[deep_fuzz] {seed - seed0 + 1}/{n} cases, {bad} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"[deep_fuzz] done: {n} cases (orders {orders}, {n_split} stitched from two combination runs), {bad} failures; "
          f"tile visits of the last launches: {kinds}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
