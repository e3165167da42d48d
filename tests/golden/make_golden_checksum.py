#!/usr/bin/env python3
"""
A FITS file whose HDUs carry CHECKSUM / DATASUM cards written by astropy 4.3.1 (`writeto(..., checksum=True)`): a float32
image, an int16 image with BZERO, a tile-compressed image.  Pins the package's restatement of the FITS checksum (standard
4.0 appendix J; `utils/fits_io._sum32`, `_encode_checksum`) -- every HDU of the file must add up to -0 -- and lets the test
check that `write_corrected_fits` leaves valid checksums behind (the reference, like astropy's default, leaves stale ones).
Stage B, in the same run: the corrected file this package writes from it is verified by astropy (`verify_checksum() == 1`,
`verify_datasum() == 1` for every HDU).

Run (build container only):   python tests/golden/make_golden_checksum.py
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "checksum", "three_hdus_checksum.fits")
SIDE = "/opt/conda/bin/python3.9"

STAGE_A = r'''
import sys
import numpy as np
for _n, _v in [("asscalar", lambda a: a.item()), ("alen", len)]:
    if not hasattr(np, _n):
        setattr(np, _n, _v)
from astropy.io import fits
rng = np.random.default_rng(7)
def hdr(name):
    h = fits.Header()
    h["EXTNAME"] = name
    for k, v, c in [("CRVAL1", -310.0, "[arcsec] reference value"), ("CRVAL2", 420.0, "[arcsec]"), ("CDELT1", 0.492, ""),
                    ("CDELT2", 0.492, ""), ("CRPIX1", 16.5, ""), ("CRPIX2", 12.5, ""), ("CUNIT1", "arcsec", ""),
                    ("CUNIT2", "arcsec", ""), ("CTYPE1", "HPLN-TAN", ""), ("CTYPE2", "HPLT-TAN", ""), ("CROTA", 3.0, "[deg] roll"),
                    ("PC1_1", float(np.cos(np.deg2rad(3.0))), ""), ("PC1_2", float(-np.sin(np.deg2rad(3.0))), ""),
                    ("PC2_1", float(np.sin(np.deg2rad(3.0))), ""), ("PC2_2", float(np.cos(np.deg2rad(3.0))), "")]:
        h[k] = (v, c) if c else v
    h["HISTORY"] = "made for the checksum test"
    return h
img = (100 + 10 * rng.standard_normal((24, 32))).astype(np.float32)
hl = fits.HDUList([fits.PrimaryHDU(),
                   fits.ImageHDU(img, hdr("F32")),
                   fits.ImageHDU((img * 20).astype(np.uint16), hdr("U16")),
                   fits.CompImageHDU((img * 20).astype(np.int16), fits.ImageHDU((img * 20).astype(np.int16), hdr("RICE")).header,
                                     compression_type="RICE_1")])
hl.writeto(sys.argv[1], overwrite=True, checksum=True)
'''

STAGE_B = r'''
import sys, json
import numpy as np
for _n, _v in [("asscalar", lambda a: a.item()), ("alen", len)]:
    if not hasattr(np, _n):
        setattr(np, _n, _v)
from astropy.io import fits
out = {}
with fits.open(sys.argv[1], checksum=True, disable_image_compression=True) as hl:
    for h in hl:
        out[h.name] = [int(h.verify_checksum()), int(h.verify_datasum())]  # 1 verified, 0 failed, 2 no such card
print(json.dumps(out))
'''


def astropy_verdict(path):
    import json
    r = subprocess.run([SIDE, "-W", "ignore", "-c", STAGE_B, path], check=True, capture_output=True, text=True)
    return json.loads(r.stdout.strip().splitlines()[-1])


def main():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run([SIDE, "-W", "ignore", "-c", STAGE_A, OUT], check=True)
    before = astropy_verdict(OUT)
    # (astropy 4.3.1 does not verify the CHECKSUM / DATASUM it writes itself for a tile-compressed HDU: they do not
    # describe the bytes on disk.  The package keeps such cards exactly as consistent as it found them.)
    assert before["PRIMARY"] == before["F32"] == before["U16"] == [1, 1], before
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import numpy as np
    from euispice_coreg_amd.hdrshift import AlignmentResults
    corr = np.zeros((5, 5, 1, 1, 1, 1))
    corr[2, 3] = 1.0
    lag = np.arange(-2.0, 3.0)
    R = AlignmentResults(corr, lag, lag, None, [0], [0.5], "arcsec", image_to_align_path=OUT)
    tmp = OUT + ".corrected"
    R.write_corrected_fits([1, 2, 3], tmp)
    after = astropy_verdict(tmp)
    assert after == before, (before, after)
    os.remove(tmp)
    print("astropy verifies the corrected file as it verifies the input:", after)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
