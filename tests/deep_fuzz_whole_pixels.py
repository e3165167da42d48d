"""Randomised run of the class tests/test_gpu_fuzz.py::test_lags_of_whole_pixels_under_an_unrotated_header pins on one
scene: unrotated headers and CRVAL lags that are whole multiples of CDELT (optionally on one axis only, optionally with a
CROTA lag on top) -- every coordinate near an integer, rows and columns on the bounds rule.  Random sizes, pixel scales
(either sign of CDELT1), NaN fractions, orders 1-3, sub-map and full-grid semantics, arcsec / degrees.
usage: python tests/deep_fuzz_whole_pixels.py [n] [seed0]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_case(seed):
    """(image to align, its header, reference image, its header, lags, spline order, full-grid semantics?, unit)"""
    import numpy as np
    from euispice_coreg_amd import synthetic
    rng = np.random.default_rng(seed)
    ny, nx = int(rng.integers(30, 80)), int(rng.integers(30, 80))
    unit = "deg" if rng.random() < 0.3 else "arcsec"
    small, hs, large, hl, _ = synthetic.make_scene(
        small_shape=(ny, nx), small_cdelt=(float(rng.uniform(10, 30)), float(rng.uniform(10, 30))),
        large_n=int(rng.integers(90, 130)), seed=seed, n_blobs=80, nan_frac=float(rng.choice([0.0, 0.01, 0.03])),
        pointing_error=(float(rng.uniform(-40, 40)), float(rng.uniform(-40, 40)), 0.0), small_unit=unit)
    hs = dict(hs)
    hs["CDELT1"] *= float(rng.choice([1.0, -1.0]))
    hs.update(CROTA=0.0, PC1_1=1.0, PC1_2=0.0, PC2_1=0.0, PC2_2=1.0)
    k1 = np.unique(rng.integers(-3, 4, int(rng.integers(1, 5)))).astype(float)
    k2 = np.unique(rng.integers(-3, 4, int(rng.integers(1, 5)))).astype(float)
    if rng.random() < 0.3:
        k1 = k1 + float(rng.uniform(-0.4, 0.4))  # whole pixels on one axis only
    l1, l2 = k1 * abs(hs["CDELT1"]), k2 * hs["CDELT2"]
    crot = None if rng.random() < 0.6 else [0.0, float(rng.choice([0.25, -0.5]))]
    order = int(rng.choice([1, 2, 3]))
    serial = bool(rng.integers(0, 2))
    return small, hs, large, hl, (l1, l2, None, None, crot), order, serial, unit


def main():
    import warnings
    import numpy as np
    from euispice_coreg_amd import _lib
    from tests import helpers as H
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    h = _lib.CoregHandle(-1)
    bad, t0 = 0, time.time()
    for seed in range(seed0, seed0 + n):
        small, hs, large, hl, lags, order, serial, unit = make_case(seed)
        try:
            with warnings.catch_warnings(), np.errstate(invalid="ignore", divide="ignore"):
                warnings.simplefilter("ignore")
                want = H.oracle_helio(small, hs, large, hl, lags, order=order, parallelism=not serial, unit_lag=unit)
                got = H.gpu_helio(h, small, hs, large, hl, lags, order=order, serial_semantics=serial)
            H.assert_corr_close(got, want, 1e-7, f"seed={seed}")
        except AssertionError as e:
            bad += 1
            print(f"FAIL seed={seed} order={order} serial={serial} unit={unit}: {str(e)[:200]}", flush=True)
        if (seed - seed0) % 50 == 49:
            print(f"[whole_pixels] {seed - seed0 + 1}/{n} cases, {bad} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"[whole_pixels] done: {n} cases, {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
