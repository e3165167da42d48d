#!/usr/bin/env python3
"""
A tile-compressed 2048^2 float32 image (RICE_1, quantize level 16, SUBTRACTIVE_DITHER_1: the shape of an EUI level-2
file) through the pieces of this package, next to astropy's own decompression of the same file.
The file is written at run time by the side interpreter that has astropy 4.3.1 (/opt/conda/bin/python3.9; the box's
image has it) -- nothing here needs astropy at import time; without that interpreter the script says so and stops.
-> one JSON line
"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SIDE = "/opt/conda/bin/python3.9"

WRITER = r'''
import sys, time, json, numpy as np
for _n, _v in [("asscalar", lambda a: a.item()), ("alen", len)]:
    if not hasattr(np, _n):
        setattr(np, _n, _v)
from astropy.io import fits
data = np.load(sys.argv[1])
hdr = json.load(open(sys.argv[2]))
hdu = fits.CompImageHDU(data=data, compression_type="RICE_1", quantize_level=16.0, quantize_method=1, dither_seed=4242)
for k, v in hdr.items():
    if k not in ("NAXIS", "NAXIS1", "NAXIS2", "BITPIX"):
        hdu.header[k] = v
fits.HDUList([fits.PrimaryHDU(), hdu]).writeto(sys.argv[3], overwrite=True)
t = []
for _ in range(3):
    t0 = time.perf_counter()
    with fits.open(sys.argv[3]) as hl:
        d = np.array(hl[1].data)
    t.append(time.perf_counter() - t0)
np.save(sys.argv[4], d)
print(json.dumps({"astropy_open_and_decode_ms": 1e3 * min(t)}))
'''


def best(fn, n=5):
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = fn()
        t.append(time.perf_counter() - t0)
    return 1e3 * min(t), r


def main():
    if not os.path.exists(SIDE):
        print(json.dumps({"skipped": "no side interpreter with astropy to write the compressed file"}))
        return
    from euispice_coreg_amd import _lib, synthetic
    from euispice_coreg_amd.hdrshift import Alignment
    from euispice_coreg_amd.utils import fits_io
    d = tempfile.mkdtemp(prefix="coreg_comp_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    # (no NaN pixels: the OLD astropy that writes the file here mis-encodes NaNs of large quantized images -- its own
    # decode of its own file then holds -1.7e9 where the NaNs were; the decoders agree on that garbage bit for bit)
    small, hs, large, hl, truth = synthetic.make_scene(nan_frac=0.0)
    small32 = small.astype(np.float32)
    p_npy, p_hdr, p_comp, p_dec, p_plain, p_large = (os.path.join(d, n) for n in (
        "small.npy", "hdr.json", "hri_rice.fits", "astropy_decoded.npy", "hri_plain.fits", "fsi.fits"))
    np.save(p_npy, small32)
    json.dump({k: (v if not isinstance(v, (np.floating, np.integer)) else v.item()) for k, v in hs.items()}, open(p_hdr, "w"))
    r = subprocess.run([SIDE, "-W", "ignore", "-c", WRITER, p_npy, p_hdr, p_comp, p_dec], capture_output=True, text=True)
    if r.returncode != 0:
        print(json.dumps({"skipped": "the side interpreter could not write the file", "stderr": r.stderr[-400:]}))
        return
    out = json.loads(r.stdout.strip().splitlines()[-1])
    out["file_mib"] = os.path.getsize(p_comp) / 2**20
    fits_io.write_images(p_plain, [(None, {}), (small32, hs)])
    fits_io.write_images(p_large, [(None, {}), (large.astype(np.float32), hl)])
    want = np.load(p_dec)
    out["open_and_parse_ms"], ci = best(lambda: fits_io.open_compressed(p_comp, -1))
    out["host_decode_ms"], got = best(lambda: ci.decode(), 3)
    out["host_decode_equals_astropy"] = bool(np.array_equal(got, want, equal_nan=True))
    h = _lib.shared_handle(-1, 0)

    def up(img):
        h.set_small(img)
        h.synchronize()
    out["gpu_upload_and_decode_ms"], _ = best(lambda: up(ci))
    out["plain_float32_upload_ms"], _ = best(lambda: up(got))
    raw = fits_io.open_raw(p_plain, -1)
    out["raw_plain_file_upload_ms"], _ = best(lambda: up(raw))
    # the pixels the GPU decoded are the ones astropy decodes: the same sweep from either, bit for bit
    grid = _lib.Grid((200, 300), (-20, 20), (2048, 2048))
    h.prepare_reference_carrington(large.astype(np.float32), hl, grid, 1.004, 2)
    ls = _lib.LagSet(np.arange(12.0, 22.0), np.arange(-14.0, -4.0), None, None, None)
    h.set_small(ci)
    a = h.sweep_carrington(hs, grid, 1.004, ls)
    h.set_small(want)
    b_ = h.sweep_carrington(hs, grid, 1.004, ls)
    out["gpu_pixels_equal_astropy"] = bool(np.array_equal(a, b_, equal_nan=True) and np.isfinite(a).all())
    out["quantization_rms_error"] = float(np.sqrt(np.nanmean((want.astype(np.float64) - small32) ** 2)))
    lag = np.arange(-30, 30, 1.0)

    def call(path):
        A = Alignment(large_fov_known_pointing=p_large, small_fov_to_correct=path, lag_crval1=lag, lag_crval2=lag,
                      lag_cdelt1=[0], lag_cdelt2=[0], lag_crota=[0], parallelism=True)
        return A.align_using_carrington(lonlims=(200, 300), latlims=(-20, 20), shape=(2048, 2048))
    call(p_comp)
    out["alignment_call_compressed_file_ms"], res = best(lambda: call(p_comp))
    out["alignment_call_plain_file_ms"], res2 = best(lambda: call(p_plain))
    out["shift_compressed"], out["shift_plain"] = [float(v) for v in res.shift_arcsec[:2]], [float(v) for v in res2.shift_arcsec[:2]]
    t0 = time.perf_counter()
    res.write_corrected_fits([-1], os.path.join(d, "out.fits"))
    out["write_corrected_fits_compressed_ms"] = 1e3 * (time.perf_counter() - t0)
    print(json.dumps(out))
    for f in os.listdir(d):
        os.remove(os.path.join(d, f))
    os.rmdir(d)


if __name__ == "__main__":
    main()
