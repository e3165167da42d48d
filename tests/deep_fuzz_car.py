"""Long randomised run of the two generators of tests/test_gpu_car.py -- pairs of Carrington maps
(`align_using_initial_carrington`: plate-carree WCS on both sides, wcslib's CAR chain restated) through the full-grid and
the sub-map semantics, the latter with lag axes THROUGH the identity lag -- over many more seeds than the test suite runs.
usage: python tests/deep_fuzz_car.py [n] [seed0] [order]   (order: force a spline order, e.g. 3, for the sub-map cases)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import warnings
    from euispice_coreg_amd import _lib
    from tests import test_gpu_car as T
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    force = int(sys.argv[3]) if len(sys.argv) > 3 else None
    h = _lib.CoregHandle(-1)
    bad, t0 = 0, time.time()

    def forced(hh, seed):
        from tests import helpers as H
        got, want, lags, order, _ = T._sub_map_case(hh, seed, force_order=force)
        H.assert_corr_close(got, want, 1e-7, f"CAR sub-map fuzz seed={seed} order={order}")

    for seed in range(seed0, seed0 + n):
        fns = (forced,) if force is not None else (T.test_fuzz_car, T.test_fuzz_initial_carrington_sub_map_semantics_with_the_identity_lag)
        for fn in fns:
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    fn(h, seed)
            except AssertionError as e:
                bad += 1
                print(f"FAIL seed={seed} {fn.__name__}: {str(e)[:300]}", flush=True)
            except _lib.CoregError as e:  # a drawn header the library refuses (explicit LONPOLE south of the equator)
                print(f"refused seed={seed} {fn.__name__}: {str(e)[:120]}", flush=True)
        if (seed - seed0) % 20 == 19:
            print(f"[deep_fuzz_car] {seed - seed0 + 1}/{n} seeds (two cases each), {bad} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"[deep_fuzz_car] done: {n} seeds x {1 if force is not None else 2} cases, {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
