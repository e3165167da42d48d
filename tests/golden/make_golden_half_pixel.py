#!/opt/conda/bin/python3.9
"""
Golden-vector generator: the REFERENCE's own `Alignment.align_using_helioprojective` (`hdrshift/alignment.py`) on lags of
HALF and WHOLE pixels under unrotated headers -- the lags that bring whole rows and columns of coordinates back within
wcslib's rounding noise of k + 1/2 (where an even spline order changes its footprint, floor(c + 0.5)) and of k (odd
orders, and the bounds rule at every order).  Four seeded scenes with 3 % NaN pixels x both branches x orders 1, 2, 3;
same file layout as `make_golden_alignment.py` (whose writer / runner it imports):

    tests/golden/half_pixel_golden.npz / .json

Run (build container only; seconds):
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_half_pixel.py
"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_alignment as M  # noqa: E402  (loads the reference through _reference_loader)

import numpy as np  # noqa: E402


def main():
    tmp = tempfile.mkdtemp(prefix="golden_half_pixel_")
    M.ARR.clear()
    M.META.update(scenes={}, cases={}, interpreter={})
    for k, seed in enumerate((11, 12, 13, 14)):
        small, hs, large, hl, _ = M.synthetic.make_scene(small_shape=(54, 46), small_cdelt=(20.0, 18.0), large_n=112,
                                                         seed=seed, n_blobs=90, nan_frac=0.03,
                                                         pointing_error=(40.0, -36.0, 0.0))
        hs = dict(hs)
        hs.update(CROTA=0.0, PC1_1=1.0, PC1_2=0.0, PC2_1=0.0, PC2_2=1.0)
        if k % 2:
            hs["CDELT1"] = -hs["CDELT1"]
        paths = M.write_pair(tmp, f"H{k}", small, hs, large, hl)
        l1 = [float(v) for v in np.array([-1.5, -0.5, 0.0, 0.5, 2.0]) * abs(hs["CDELT1"])]
        l2 = [float(v) for v in np.array([-0.5, 0.0, 1.0, 1.5]) * hs["CDELT2"]]
        for par in (True, False):
            for order in (1, 2, 3):
                e = M.run_case(f"H{k}_{'par' if par else 'ser'}_o{order}", f"H{k}", paths,
                               dict(lag_crval1=l1, lag_crval2=l2, lag_cdelt1=None, lag_cdelt2=None, lag_crota=None,
                                    parallelism=par, counts_cpu_max=3, reprojection_order=order), "helioprojective")
                assert "raises" not in e, e
    import astropy
    import scipy
    M.META["interpreter"] = {"python": sys.version.split()[0], "numpy": np.__version__, "scipy": scipy.__version__,
                             "astropy": astropy.__version__}
    dst = os.path.join(HERE, "half_pixel_golden.npz")
    np.savez_compressed(dst, **M.ARR)
    with open(os.path.join(HERE, "half_pixel_golden.json"), "w") as f:
        json.dump(M.META, f, indent=1, sort_keys=True)
    print("wrote", dst, os.path.getsize(dst), "bytes,", len(M.META["cases"]), "cases")


if __name__ == "__main__":
    main()
