"""Long randomised GPU-vs-oracle parity run (the generator of tests/test_gpu_fuzz.py over many more seeds, plus
method='residus', CDELT2 lags, both CDELT semantics, degrees headers).  usage: python tests/deep_fuzz.py [n] [seed0] [scale]"""
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
    h = _lib.CoregHandle(-1)
    bad = 0
    # tile visits of each case's LAST sweep launch; lag-points re-evaluated with centred sums (whole sweeps)
    kinds = {"visits": 0, "lds": 0, "interior": 0, "all_finite": 0, "refined_lag_points": 0}
    t0 = time.time()
    for seed in range(seed0, seed0 + n):
        small, hs, large, hl, lags, rng = _random_case(seed, scale)
        order = int(rng.choice([1, 2]))
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
    print(f"[deep_fuzz] done: {n} cases, {bad} failures; tile visits of the last launches: {kinds}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
