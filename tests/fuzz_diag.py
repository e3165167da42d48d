"""Look into ONE case of tests/deep_fuzz.py (helioprojective frame): which lag-point deviates from the oracle, by how much,
and whether the deviation belongs to the image form (file upload) or to the sweep itself.
usage: python tests/fuzz_diag.py seed scale order sem(intended|reference) forms(0|1) [orders of the run, default 1,2,3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    from euispice_coreg_amd import _lib
    from tests import helpers as H
    from tests import deep_fuzz as DF
    seed, scale, order, sem, forms = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5] == "1"
    orders = [int(v) for v in sys.argv[6].split(",")] if len(sys.argv) > 6 else [1, 2, 3]
    case = DF.build_case(seed, scale, orders, forms)
    print({k: v for k, v in case.items() if k in ("order", "sem", "frame", "serial", "forms_used")})
    assert case["frame"] == "helio" and case["order"] == order and case["sem"] == sem
    h = _lib.CoregHandle(-1)
    small, hs, large, hl, lags = case["small"], case["hs"], case["large"], case["hl"], case["lags"]
    want = H.oracle_helio(small, hs, large, hl, lags, order=order, parallelism=not case["serial"], cdelt_semantics=sem)
    for name, (su, lu) in {"as uploaded": (case["small_up"], case["large_up"]), "decoded arrays": (small, large)}.items():
        got = H.gpu_helio(h, su, hs, lu, hl, lags, order=order, serial_semantics=case["serial"],
                          cdelt_semantics=0 if sem == "intended" else 1)
        d = np.abs(got - want)
        k = np.unravel_index(np.nanargmax(d), d.shape)
        ls = _lib.LagSet(*lags)
        print(f"{name}: max|dcorr| = {np.nanmax(d):.3e} at index {k}: lags = "
              f"{[float(a[i]) for a, i in zip(ls.arrays, k[:5])]}, got {got[k]:.12f} want {want[k]:.12f}; "
              f"second largest {np.sort(d[np.isfinite(d)])[-2]:.3e}; n > 1e-7: {(d > 1e-7).sum()} of {d.size}; "
              f"tap fix: {h.last_tap_fix()}")
    for o in (1, 2, 3):
        want_o = H.oracle_helio(small, hs, large, hl, lags, order=o, parallelism=not case["serial"], cdelt_semantics=sem)
        got_o = H.gpu_helio(h, small, hs, large, hl, lags, order=o, serial_semantics=case["serial"],
                            cdelt_semantics=0 if sem == "intended" else 1)
        print(f"order {o}: max|dcorr| = {np.nanmax(np.abs(got_o - want_o)):.3e}")
    print("header:", {k: hs[k] for k in ("CRPIX1", "CRPIX2", "CDELT1", "CDELT2", "CROTA", "CRVAL1", "CRVAL2", "NAXIS1", "NAXIS2")})


if __name__ == "__main__":
    main()
