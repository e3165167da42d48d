#!/opt/conda/bin/python3.9
"""
THE HEADLINE WORKLOAD AS THE REFERENCE ITSELF RUNS IT, on a sub-lattice of its lags: 2048 x 2048 image to align against
a 3072 x 3072 reference image, Carrington frame, grid 2048 x 2048 over lon (200, 300) x lat (-20, 20), solar_r 1.004,
order 2, `parallelism=True` -- the reference's own `Alignment(...).align_using_carrington(return_type='corr')`
(`hdrshift/alignment.py:144-250, 613-797`, `utils/rectify.py`) executed in the build container (side interpreter +
load-time shims: `_reference_loader.py`).  Lag-points are independent of one another, so a sub-lattice of the
60 x 60 lags arange(-30, 30, 1) gives entries of the headline map itself:

    lattice   lag_crval1 = lag_crval2 = arange(-30, 30, 8) + 2      ->  8 x 8 entries spread over the map
    peak      lag_crval1 = 15 .. 19, lag_crval2 = -11 .. -7         ->  5 x 5 entries around the injected shift (17, -9)

Two steps, so that the pixels are exactly those the tests regenerate (numpy 2.2 of the main interpreter):
    python tests/golden/make_golden_headline.py --dump-scene /tmp/headline_scene.npz
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_headline_reference.py /tmp/headline_scene.npz
-> tests/golden/headline_reference.npz (raveled indices into the 60 x 60 map, the reference's coefficients, the scene
fingerprint).  About a minute on 8 cores.
"""
import json
import os
import sys
import tempfile
import time
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _reference_loader  # noqa: E402

_reference_loader.load_reference()
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402
from astropy.io import fits  # noqa: E402

from euispice_coreg.hdrshift.alignment import Alignment  # noqa: E402


def main():
    sc = np.load(sys.argv[1], allow_pickle=False)
    small, large = sc["small"], sc["large"]  # float32, as a BITPIX = -32 file holds them
    hs, hl = json.loads(str(sc["hdr_small"])), json.loads(str(sc["hdr_large"]))
    tmp = tempfile.mkdtemp(prefix="golden_headline_")
    ps, pl = os.path.join(tmp, "small.fits"), os.path.join(tmp, "large.fits")
    for p, img, h in ((ps, small, hs), (pl, large, hl)):
        hdr = fits.Header()
        for k, v in h.items():
            if not k.startswith("NAXIS"):
                hdr[k] = v
        fits.HDUList([fits.PrimaryHDU(), fits.ImageHDU(data=img, header=hdr)]).writeto(p, overwrite=True)
    full = np.arange(-30.0, 30.0, 1.0)
    index, corr = [], []
    for name, l1, l2 in (("lattice", np.arange(-30.0, 30.0, 8.0) + 2.0, np.arange(-30.0, 30.0, 8.0) + 2.0),
                         ("peak", np.arange(15.0, 20.0, 1.0), np.arange(-11.0, -6.0, 1.0))):
        t = time.time()
        A = Alignment(large_fov_known_pointing=pl, small_fov_to_correct=ps, lag_crval1=l1, lag_crval2=l2, lag_cdelt1=None,
                      lag_cdelt2=None, lag_crota=None, parallelism=True, counts_cpu_max=8)
        c = A.align_using_carrington(lonlims=[200.0, 300.0], latlims=[-20.0, 20.0], shape=[2048, 2048],
                                     return_type="corr")[:, :, 0, 0, 0, 0]
        i1 = np.searchsorted(full, l1)
        i2 = np.searchsorted(full, l2)
        assert np.array_equal(full[i1], l1) and np.array_equal(full[i2], l2)
        index.append((i1[:, None] * 60 + i2[None, :]).ravel())
        corr.append(c.ravel())
        print(name, c.shape, "max", float(np.nanmax(c)), "at", np.unravel_index(np.nanargmax(c), c.shape),
              f"{time.time() - t:.1f} s", flush=True)
    index, corr = np.concatenate(index), np.concatenate(corr)
    assert np.unique(index).size == index.size and np.isfinite(corr).all()
    dst = os.path.join(HERE, "headline_reference.npz")
    np.savez(dst, index=index, corr=corr, fingerprint=sc["fingerprint"])
    print("wrote", dst, os.path.getsize(dst), "bytes,", index.size, "lag-points")


if __name__ == "__main__":
    main()
