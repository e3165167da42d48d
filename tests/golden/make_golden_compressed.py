#!/opt/conda/bin/python3.9
"""
Golden files for the tile-compressed FITS images the reference reads through astropy.io.fits (EUI level-1 / level-2
files store their image in a compressed binary-table HDU; call sites alignment.py:191-208, :299-314; the writer
utils/Util.py:106-159 has a CompImageHDU branch).  The (de)compression itself is third-party code absent from
/root/reference: cfitsio's tiled-image convention (ricecomp.c, imcompress.c, quantize.c), bundled with astropy.

This script writes small compressed files with astropy 4.3.1 (its bundled cfitsio) and records what astropy decodes
from them, so that the library's own decoder (csrc/ricecomp.hpp) can be pinned bit for bit:
  tests/golden/compressed/<case>.fits    the files (data: inputs of the test)
  tests/golden/compressed_golden.npz     <case>/data = hdul[1].data as astropy returns it (expected output)

Run (build container only):   /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_compressed.py
"""
import os

import numpy as np

for _n, _v in [("asscalar", lambda a: a.item()), ("alen", len)]:
    if not hasattr(np, _n):
        setattr(np, _n, _v)

from astropy.io import fits  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "compressed")


def scene(rng, ny, nx):
    y, x = np.mgrid[0:ny, 0:nx]
    img = 100.0 + 900.0 * np.exp(-((x - 0.4 * nx) ** 2 + (y - 0.6 * ny) ** 2) / (0.05 * nx * ny))
    img += 300.0 * np.exp(-((x - 0.8 * nx) ** 2 + (y - 0.2 * ny) ** 2) / (0.01 * nx * ny))
    return img + np.sqrt(img) * rng.standard_normal(img.shape)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20240404)
    f = scene(rng, 70, 93).astype(np.float32)
    fn = f.copy()
    fn[5, 7] = np.nan
    fn[30:33, 40:60] = np.nan
    fz = f.copy()
    fz[10:20, :] = 0.0          # exact zeros: SUBTRACTIVE_DITHER_2 keeps them exact
    fconst = f.copy()
    fconst[12, :] = 250.0       # a constant row tile: cannot be quantized -> stored in GZIP_COMPRESSED_DATA
    i16 = np.clip(scene(rng, 64, 80) * 20.0 - 9000.0, -32768, 32767).astype(np.int16)
    u16 = np.clip(scene(rng, 64, 80) * 40.0, 0, 65535).astype(np.uint16)
    u8 = np.clip(scene(rng, 40, 50) / 5.0, 0, 255).astype(np.uint8)
    i32 = (scene(rng, 33, 47) * 70001.0).astype(np.int32)
    f64 = scene(rng, 50, 60)
    wide = scene(rng, 9, 300).astype(np.float32)  # rows longer than several Rice blocks
    big = scene(rng, 200, 260).astype(np.float32)  # several decoder threads / GPU workgroups
    big[50:60, 100:140] = np.nan
    cases = {
        "rice_i16": (i16, dict(compression_type="RICE_1")),
        "rice_u16": (u16, dict(compression_type="RICE_1")),                      # BZERO = 32768 convention
        "rice_u8": (u8, dict(compression_type="RICE_1")),
        "rice_i32": (i32, dict(compression_type="RICE_1")),
        "rice_f32_nodither": (f, dict(compression_type="RICE_1", quantize_level=16.0, quantize_method=-1)),
        "rice_f32_dither1": (f, dict(compression_type="RICE_1", quantize_level=16.0, quantize_method=1, dither_seed=1)),
        "rice_f32_dither1_seed9999": (f, dict(compression_type="RICE_1", quantize_level=8.0, quantize_method=1,
                                              dither_seed=9999)),
        "rice_f32_dither2_zeros": (fz, dict(compression_type="RICE_1", quantize_level=16.0, quantize_method=2,
                                            dither_seed=17)),
        "rice_f32_nan": (fn, dict(compression_type="RICE_1", quantize_level=16.0, quantize_method=1, dither_seed=5)),
        "rice_f32_tiles2d": (fn, dict(compression_type="RICE_1", quantize_level=16.0, quantize_method=1, dither_seed=3,
                                      tile_size=(16, 20))),
        "rice_f32_const_tile": (fconst, dict(compression_type="RICE_1", quantize_level=16.0, quantize_method=1,
                                             dither_seed=2)),
        "rice_f32_wide": (wide, dict(compression_type="RICE_1", quantize_level=4.0, quantize_method=1, dither_seed=400)),
        "rice_f32_big": (big, dict(compression_type="RICE_1", quantize_level=16.0, quantize_method=1, dither_seed=7321)),
        "rice_f64": (f64, dict(compression_type="RICE_1", quantize_level=16.0, quantize_method=1, dither_seed=11)),
        "gzip1_f32_lossless": (fn, dict(compression_type="GZIP_1", quantize_level=0.0)),
        "gzip2_f32_lossless": (fn, dict(compression_type="GZIP_2", quantize_level=0.0)),
        "gzip1_i16": (i16, dict(compression_type="GZIP_1")),
    }
    gold = {}
    hdr = fits.Header()
    for k, v in (("CRVAL1", -310.0), ("CRVAL2", 420.0), ("CDELT1", 0.492), ("CDELT2", 0.492), ("CRPIX1", 40.5),
                 ("CRPIX2", 32.5), ("CUNIT1", "arcsec"), ("CUNIT2", "arcsec"), ("CROTA", 3.0),
                 ("DATE-AVG", "2022-03-17T09:50:45.277"), ("EXTNAME", "IMAGE")):
        hdr[k] = v
    for name, (data, kw) in cases.items():
        path = os.path.join(OUT, name + ".fits")
        hdu = fits.CompImageHDU(data=data, **kw)
        for k in hdr:
            hdu.header[k] = hdr[k]
        fits.HDUList([fits.PrimaryHDU(), hdu]).writeto(path, overwrite=True)
        with fits.open(path) as hl:
            dec = np.array(hl[1].data)
            gold[name + "/data"] = dec
            ih = hl[1].header
            gold[name + "/header_keys"] = np.array(sorted(k for k in ih.keys() if k))
        with fits.open(path, disable_image_compression=True) as hl:
            th = hl[1].header
            print(f"{name:28s} {os.path.getsize(path):7d} B  decoded {dec.dtype} {dec.shape}  ZCMPTYPE={th['ZCMPTYPE']} "
                  f"ZQUANTIZ={th.get('ZQUANTIZ')} ZDITHER0={th.get('ZDITHER0')} ZTILE=({th['ZTILE1']},{th['ZTILE2']}) "
                  f"cols={[th['TTYPE%d' % (i + 1)] for i in range(th['TFIELDS'])]} "
                  f"max|dec-in|={np.nanmax(np.abs(dec.astype(np.float64) - data.astype(np.float64))):.3g}")
    np.savez_compressed(os.path.join(HERE, "compressed_golden.npz"), **gold)
    print("wrote", len(cases), "files +", os.path.join(HERE, "compressed_golden.npz"))


if __name__ == "__main__":
    main()
