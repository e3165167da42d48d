#!/usr/bin/env python3
"""
End-to-end example on synthetic data (no network, no astropy): write an HRIEUV-like image with a known pointing error
and an FSI-like reference as FITS files, align them exactly as with euispice_coreg (README.md:47-139 of the reference),
print the recovered shift and write the corrected FITS.  Needs an MI355X and the built library
(python -c "import __graft_entry__ as g; g.build()").

    python examples/align_synthetic.py [size] [--compressed]   # size of the small image, default 1024;
                                                               # --compressed: the image to align is written the way
                                                               # EUI files are (Rice-compressed tiles, decoded on the GPU)
"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from euispice_coreg_amd import synthetic  # noqa: E402
from euispice_coreg_amd.hdrshift import Alignment  # noqa: E402
from euispice_coreg_amd.utils import fits_io  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if args else 1024
    d = tempfile.mkdtemp(prefix="coreg_example_")
    small, hs, large, hl, truth = synthetic.make_scene(small_n=n, large_n=3 * n // 2)
    path_hri, path_fsi = os.path.join(d, "hri.fits"), os.path.join(d, "fsi.fits")
    if "--compressed" in sys.argv:
        info = fits_io.write_compressed_image(path_hri, small.astype(np.float32), hs, quantize="SUBTRACTIVE_DITHER_2")
        print(f"image to align: tile-compressed, {info['compressed_bytes'] / small.size:.2f} bytes per pixel")
    else:
        fits_io.write_images(path_hri, [(None, {}), (small.astype(np.float32), hs)])
    fits_io.write_images(path_fsi, [(None, {}), (large.astype(np.float32), hl)])
    print(f"injected pointing error: CRVAL ({truth['lag_crval1']}, {truth['lag_crval2']}) arcsec, "
          f"CROTA {truth['lag_crota']} deg")

    lag = np.arange(-30, 30, 1.0)
    for frame in ("carrington", "helioprojective"):
        A = Alignment(large_fov_known_pointing=path_fsi, small_fov_to_correct=path_hri, lag_crval1=lag, lag_crval2=lag,
                      lag_cdelt1=[0], lag_cdelt2=[0], lag_crota=[0.0, 0.3], parallelism=True, small_fov_value_max=2900.0)
        t0 = time.perf_counter()
        if frame == "carrington":
            res = A.align_using_carrington(lonlims=(228, 262), latlims=(-12, 22), shape=(n, n))
        else:
            res = A.align_using_helioprojective()
        dt = time.perf_counter() - t0
        st = A.last_stats
        print(f"[{frame}] {res.corr.size} lag-points in {dt * 1e3:.1f} ms (sweep kernel {st['sweep_kernel_ms']:.2f} ms, "
              f"{st['n_active_points']} active grid points){res}")
        out = os.path.join(d, f"hri_corrected_{frame}.fits")
        res.write_corrected_fits([-1], out)
        h = fits_io.read_header(out, -1)
        print(f"  corrected CRVAL = ({h['CRVAL1']:.3f}, {h['CRVAL2']:.3f}) arcsec, CROTA = {h['CROTA']:.3f} deg -> {out}")


if __name__ == "__main__":
    main()
