#!/opt/conda/bin/python3.9
"""
BASELINE.json configs[0] AS THE REFERENCE ITSELF RUNS IT: synthetic 512 x 512 small-FOV image against a 1024 x 1024
large-FOV image, helioprojective, lag_crval1/2 in [-5, 5] step 1 arcsec, crota / cdelt fixed, `parallelism=False` -- the
reference's own `Alignment(...).align_using_helioprojective(return_type='corr')` executed in the build container
(side interpreter + load-time shims: `_reference_loader.py`), plus the same window through its parallel branch (sub-map
semantics; contains the zero lag, whose border pixels wcslib's rounding noise decides), the window centred on the
injected shift (17, -9) in both branches, and the Carrington frame on a 512 x 512 grid.

Two steps, so that the pixels are exactly those the tests regenerate (numpy 2.2 of the main interpreter):
    python tests/golden/make_golden_cfg1.py --dump-scene /tmp/cfg1_scene.npz
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_cfg1_reference.py /tmp/cfg1_scene.npz
-> tests/golden/cfg1_reference.npz (maps, headers as astropy read them back, scene fingerprint).  About four minutes.
"""
import json
import os
import sys
import tempfile
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
    tmp = tempfile.mkdtemp(prefix="golden_cfg1_")
    ps, pl = os.path.join(tmp, "small.fits"), os.path.join(tmp, "large.fits")
    for p, img, h in ((ps, small, hs), (pl, large, hl)):
        hdr = fits.Header()
        for k, v in h.items():
            if not k.startswith("NAXIS"):
                hdr[k] = v
        fits.HDUList([fits.PrimaryHDU(), fits.ImageHDU(data=img, header=hdr)]).writeto(p, overwrite=True)
    out = {"fingerprint": sc["fingerprint"]}
    for tag, path in (("hdr_small", ps), ("hdr_large", pl)):
        with fits.open(path) as f:
            h = f[-1].header
            out[tag] = np.array(json.dumps({k: (h[k] if isinstance(h[k], (str, bool, int)) else float(h[k]))
                                            for k in h.keys() if k not in ("XTENSION", "BITPIX", "PCOUNT", "GCOUNT", "")}))
    base = np.arange(-5.0, 6.0, 1.0)
    windows = {"0": (base, base), "": (17.0 + base, -9.0 + base)}
    for suffix, (l1, l2) in windows.items():
        for par in (False, True):
            A = Alignment(large_fov_known_pointing=pl, small_fov_to_correct=ps, lag_crval1=l1, lag_crval2=l2,
                          lag_cdelt1=None, lag_cdelt2=None, lag_crota=None, parallelism=par, counts_cpu_max=6)
            corr = A.align_using_helioprojective(return_type="corr")
            name = ("parallel" if par else "serial") + suffix
            out[name] = corr
            print(name, corr.shape, "max", float(np.nanmax(corr)), "argmax",
                  np.unravel_index(np.nanargmax(corr), corr.shape)[:2], flush=True)
        A = Alignment(large_fov_known_pointing=pl, small_fov_to_correct=ps, lag_crval1=l1, lag_crval2=l2, lag_cdelt1=None,
                      lag_cdelt2=None, lag_crota=None, parallelism=True, counts_cpu_max=6)
        corr = A.align_using_carrington(lonlims=[228.0, 262.0], latlims=[-12.0, 22.0], shape=[512, 512], return_type="corr")
        out["carrington" + suffix] = corr
        print("carrington" + suffix, corr.shape, "max", float(np.nanmax(corr)), flush=True)
    out["lag_base"] = base
    dst = os.path.join(HERE, "cfg1_reference.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
