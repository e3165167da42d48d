"""GPU tests of the image hand-over: the reference-image crop (only the rectangle the target grid can touch crosses
PCIe) and the device-source entry points (replicas assembled by an all-gather) must give the very pixels and maps the
plain host uploads give."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

SHAPE = (72, 64)


@pytest.mark.parametrize("order", [1, 2, 3])
@pytest.mark.parametrize("f32", [True, False])
def test_reference_crop_is_bit_identical(gpu_handle, order, f32):
    """coreg_prepare_reference_*: the prepared reference with the source image cropped to the bounding box of the sample
    coordinates (default) equals, bit for bit, the one prepared from the whole image -- Carrington grid (the limb cuts
    it: NaN outside), helioprojective sub-map, and a grid that reaches the image border (mirrored taps)."""
    from euispice_coreg_amd import _lib
    small, hs, large, hl, _ = H.scene(small_n=96, large_n=200)
    large = large.astype(np.float32) if f32 else large + 1e-9 * np.arange(large.size).reshape(large.shape)
    grids = [_lib.Grid(H.CARR_LON, H.CARR_LAT, SHAPE), _lib.Grid((150.0, 350.0), (-60.0, 60.0), (90, 80)),
             _lib.Grid((244.0, 246.5), (3.0, 5.0), (40, 50))]
    gpu_handle.set_small(small)
    for grid in grids:
        out = []
        for crop in (1, 0):
            gpu_handle.set_option("crop_reference", crop)
            try:
                gpu_handle.prepare_reference_carrington(large, hl, grid, 1.004, order)
                out.append(gpu_handle.get_reference_on_grid((grid.n_lat, grid.n_lon), np.float64))
            finally:
                gpu_handle.set_option("crop_reference", 1)
        assert np.array_equal(out[0], out[1], equal_nan=True)
    assert np.isfinite(out[0]).any()
    # helioprojective sub-map (small header's grid inside the large image), and a target that sticks out of the image
    hs_out = dict(hs)
    hs_out["CRVAL1"] = hl["CRVAL1"] + 0.5 * hl["NAXIS1"] * hl["CDELT1"]  # centred on the edge of the large image
    for hdr in (hs, hs_out):
        out = []
        for crop in (1, 0):
            gpu_handle.set_option("crop_reference", crop)
            try:
                gpu_handle.prepare_reference_helioprojective(large, hl, hdr, order)
                out.append(gpu_handle.get_reference_on_grid((hdr["NAXIS2"], hdr["NAXIS1"]), np.float32))
            finally:
                gpu_handle.set_option("crop_reference", 1)
        assert np.array_equal(out[0], out[1], equal_nan=True)
    assert np.isnan(out[0]).any() and np.isfinite(out[0]).any()  # the shifted target does leave the image


def test_full_size_reference_crop_and_upload_sizes(gpu_handle):
    """Headline inputs: the 2048 x 2048 Carrington grid touches a small part of the 3072 x 3072 reference; cropped and
    whole uploads prepare the same reference, and the sweep over it finds the injected shift."""
    from euispice_coreg_amd import _lib, synthetic
    small, hs, large, hl, truth = synthetic.make_scene()
    large32 = large.astype(np.float32)
    grid = _lib.Grid((200.0, 300.0), (-20.0, 20.0), (2048, 2048))
    ref = []
    for crop in (1, 0):
        gpu_handle.set_option("crop_reference", crop)
        try:
            gpu_handle.prepare_reference_carrington(large32, hl, grid, 1.004, 2)
            ref.append(gpu_handle.get_reference_on_grid((2048, 2048), np.float64))
        finally:
            gpu_handle.set_option("crop_reference", 1)
    assert np.array_equal(ref[0], ref[1], equal_nan=True)
    gpu_handle.set_small(small.astype(np.float32))
    gpu_handle.prepare_reference_carrington(large32, hl, grid, 1.004, 2)
    lags = _lib.LagSet(np.arange(13.0, 22.0), np.arange(-13.0, -4.0), None, None, None)
    corr = gpu_handle.sweep_carrington(hs, grid, 1.004, lags).reshape(9, 9)
    assert np.unravel_index(np.argmax(corr), corr.shape) == (4, 4)


def test_device_source_uploads_equal_host_uploads(gpu_handle):
    """coreg_set_small_from_device / coreg_prepare_reference_*_from_device (the multi-GPU hand-over: replicas assembled
    on the device by an all-gather) against the host uploads: same resident pixels, same maps; float32 and float64."""
    import torch
    from euispice_coreg_amd import _lib
    small, hs, large, hl, _ = H.scene()
    grid = _lib.Grid(H.CARR_LON, H.CARR_LAT, SHAPE)
    lags = _lib.LagSet(17.0 + 2.0 * (np.arange(5) - 2), -9.0 + 2.0 * (np.arange(4) - 2), None, None, [0.0, 0.3])
    noisy = small + 1e-7 * np.arange(small.size).reshape(small.shape)  # not float32-exact: stays float64 on the device
    for s_img, l_img in ((small.astype(np.float32), large.astype(np.float32)), (noisy, large)):
        gpu_handle.set_small(s_img)
        gpu_handle.prepare_reference_carrington(l_img, hl, grid, 1.004, 2)
        want_c = gpu_handle.sweep_carrington(hs, grid, 1.004, lags)
        gpu_handle.prepare_reference_helioprojective(l_img, hl, hs, 2)
        want_h = gpu_handle.sweep_helioprojective(hs, hs, lags)
        ts, tl = torch.from_numpy(np.ascontiguousarray(s_img)).cuda(), torch.from_numpy(np.ascontiguousarray(l_img)).cuda()
        torch.cuda.synchronize()
        gpu_handle.set_small_from_device(ts.data_ptr(), ts.shape, s_img.dtype)
        gpu_handle.prepare_reference_carrington_from_device(tl.data_ptr(), tl.shape, l_img.dtype, hl, grid, 1.004, 2)
        got_c = gpu_handle.sweep_carrington(hs, grid, 1.004, lags)
        gpu_handle.prepare_reference_helioprojective_from_device(tl.data_ptr(), tl.shape, l_img.dtype, hl, hs, 2)
        got_h = gpu_handle.sweep_helioprojective(hs, hs, lags)
        gpu_handle.synchronize()
        assert np.array_equal(got_c, want_c, equal_nan=True) and np.array_equal(got_h, want_h, equal_nan=True)
        assert gpu_handle.last_stats()["small_is_f32"] == int(s_img.dtype == np.float32)


def _write_fits(path, data, hdr, bscale=None, bzero=None):
    """One image HDU behind an empty primary; BSCALE / BZERO renamed into place (write_images treats them as its own)."""
    from euispice_coreg_amd.utils import fits_io
    h = dict(hdr)
    if bscale is not None:
        h["BSCALX"], h["BZERX"] = float(bscale), float(bzero)
    fits_io.write_images(path, [(None, {}), (data, h)])
    if bscale is not None:
        blob = open(path, "rb").read()
        open(path, "wb").write(blob.replace(b"BSCALX  =", b"BSCALE  =").replace(b"BZERX   =", b"BZERO   ="))


RAW_CASES = [("f4", None), ("f4", (0.5, 10.0)), ("f8", None), ("i2", None), ("i2", (0.25, 100.0)), ("i4", None),
             ("i4", (1e-3, -7.0)), ("u1", None), ("i8", None)]


@pytest.mark.parametrize("kind,scale", RAW_CASES)
def test_raw_fits_uploads_are_bit_identical_to_decoded_uploads(gpu_handle, tmp_path, kind, scale):
    """coreg_set_small_fits / coreg_prepare_reference_*_fits (raw big-endian data units, decoded on the GPU) against the
    pixels utils/fits_io.py decodes on the host: same resident reference, same maps, every BITPIX, with and without
    BSCALE / BZERO, reference cropped and whole."""
    from euispice_coreg_amd import _lib
    from euispice_coreg_amd.utils import fits_io
    small, hs, large, hl, _ = H.scene(small_n=96, large_n=200)

    def cast(a):
        if kind == "f4":
            return a.astype(np.float32)
        if kind == "f8":
            return a + 1e-9 * np.arange(a.size).reshape(a.shape)  # not float32-exact
        b = np.nan_to_num(a, nan=0.0)
        if kind == "u1":
            return np.clip(b / 16.0, 0, 255).astype(np.uint8)
        if kind == "i2":
            return np.clip(b * 4.0 - 2000.0, -32768, 32767).astype(np.int16)
        if kind == "i4":
            return (b * 70001.0).astype(np.int32)       # beyond 2^24: not float32-exact
        return (b * 3.0e9).astype(np.int64)
    ps, pl = str(tmp_path / "s.fits"), str(tmp_path / "l.fits")
    _write_fits(ps, cast(small), hs, *(scale or (None, None)))
    _write_fits(pl, cast(large), hl, *(scale or (None, None)))
    raw_s, raw_l = fits_io.open_raw(ps, -1), fits_io.open_raw(pl, -1)
    dec_s, dec_l = fits_io.native_pixels(fits_io.read_image(ps, -1)[0]), fits_io.native_pixels(fits_io.read_image(pl, -1)[0])
    assert raw_s is not None and raw_l is not None and np.array_equal(np.asarray(raw_s), dec_s, equal_nan=True)
    grid = _lib.Grid(H.CARR_LON, H.CARR_LAT, SHAPE)
    lags = _lib.LagSet(17.0 + 2.0 * (np.arange(5) - 2), -9.0 + 2.0 * (np.arange(4) - 2), None, None, [0.0, 0.3])
    res = {}
    for name, s_img, l_img in (("decoded", dec_s, dec_l), ("raw", raw_s, raw_l)):
        gpu_handle.set_small(s_img)
        out = [None]
        for crop in (1, 0):
            gpu_handle.set_option("crop_reference", crop)
            try:
                gpu_handle.prepare_reference_carrington(l_img, hl, grid, 1.004, 2)
                out.append(gpu_handle.get_reference_on_grid((grid.n_lat, grid.n_lon), np.float64))
            finally:
                gpu_handle.set_option("crop_reference", 1)
        out.append(gpu_handle.sweep_carrington(hs, grid, 1.004, lags))
        gpu_handle.prepare_reference_helioprojective(l_img, hl, hs, 2)
        out.append(gpu_handle.get_reference_on_grid((hs["NAXIS2"], hs["NAXIS1"]), np.float32))
        out.append(gpu_handle.sweep_helioprojective(hs, hs, lags))
        out.append(gpu_handle.resample_helioprojective(hs, hs, order=2, dtype=np.float64))  # the resident image itself
        out[0] = gpu_handle.last_stats()["small_is_f32"]
        res[name] = out
    assert res["raw"][0] == res["decoded"][0]
    for a, b in zip(res["raw"][1:], res["decoded"][1:]):
        assert np.array_equal(a, b, equal_nan=True)
    assert np.isfinite(res["raw"][3]).any()
    # thresholds on the device see the same pixels
    gpu_handle.set_small(raw_s)
    n_raw = gpu_handle.threshold_small(None, float(np.nanpercentile(np.abs(dec_s), 90)))
    gpu_handle.set_small(dec_s)
    assert n_raw == gpu_handle.threshold_small(None, float(np.nanpercentile(np.abs(dec_s), 90)))


def test_alignment_raw_fits_path_equals_host_decode(tmp_path, monkeypatch):
    """`Alignment` on float32 FITS files: memory-mapped raw upload (default) and host decode (COREG_RAW_FITS=0) give the
    same correlation array, in both frames; the image to align is never decoded on the host in the default path."""
    from euispice_coreg_amd.hdrshift import Alignment
    from euispice_coreg_amd.utils import fits_io
    small, hs, large, hl, _ = H.scene()
    ps, pl = str(tmp_path / "s.fits"), str(tmp_path / "l.fits")
    _write_fits(ps, small.astype(np.float32), hs)
    _write_fits(pl, large.astype(np.float32), hl)
    lag1, lag2 = 17.0 + 2.0 * (np.arange(5) - 2), -9.0 + 2.0 * (np.arange(4) - 2)
    out = {}
    for raw in ("1", "0"):
        monkeypatch.setenv("COREG_RAW_FITS", raw)
        A = Alignment(pl, ps, lag1, lag2, [0], [0], [0], parallelism=True, small_fov_value_max=2500.0)
        c = A.align_using_carrington(lonlims=H.CARR_LON, latlims=H.CARR_LAT, shape=SHAPE, return_type="corr")
        assert isinstance(A.data_small, fits_io.RawImage) == (raw == "1")
        B = Alignment(pl, ps, lag1, lag2, [0], [0], [0], parallelism=True)
        out[raw] = (c, B.align_using_helioprojective(return_type="corr"))
    assert np.array_equal(out["1"][0], out["0"][0], equal_nan=True)
    assert np.array_equal(out["1"][1], out["0"][1], equal_nan=True)
    assert np.isfinite(out["1"][0]).all()


def _compressed_files():
    import glob
    import os
    from tests.conftest import GOLDEN
    return sorted(glob.glob(os.path.join(GOLDEN, "compressed", "rice_*.fits")))


@pytest.mark.parametrize("path", _compressed_files(), ids=lambda p: p.split("/")[-1][:-5])
def test_tile_compressed_images_decoded_on_the_gpu_equal_the_host_decode(gpu_handle, path):
    """coreg_set_small_tiled / coreg_prepare_reference_*_tiled: the COMPRESSED bytes of a tile-compressed image (EUI files)
    go up, one GPU thread per tile runs cfitsio's RICE_1 codec + dequantization (csrc/ricecomp.hpp) -- against the same
    image decoded on the host (which tests/test_fits_compressed_cpu.py pins bit for bit to astropy's output): identical
    resident pixels, identical prepared references, identical sweeps."""
    from euispice_coreg_amd import _lib
    from euispice_coreg_amd.utils import fits_io
    ci = fits_io.open_compressed(path, -1)
    dec = fits_io.native_pixels(np.asarray(ci))
    hdr = dict(ci.header)
    hdr.update(PC1_1=1.0, PC1_2=0.0, PC2_1=0.0, PC2_2=1.0, CROTA=0.0, CTYPE1="HPLN-TAN", CTYPE2="HPLT-TAN")
    small, hs, large, hl, _ = H.scene()
    lags = _lib.LagSet([-3.0, 0.5, 4.0], [-2.0, 1.25], None, None, [0.0, 0.4])
    res = {}
    for name, img in (("host", dec), ("gpu", ci)):
        if name == "gpu" and not ci.on_gpu:   # (a tile stored gzipped: the binding decodes on the host, same numbers)
            assert not _lib._is_tiled(ci)
        out = []
        # as the image to align: the resident pixels read back through an order-1 resample on their own grid
        gpu_handle.set_small(img)
        out.append(None)
        out.append(gpu_handle.resample_helioprojective(hdr, hdr, order=1, dtype=np.float64))
        gpu_handle.prepare_reference_helioprojective(large, hl, hdr, 2)
        out.append(gpu_handle.sweep_helioprojective(hdr, hdr, lags))
        # as the reference image: sub-map on the scene's small grid, and on a Carrington grid
        gpu_handle.set_small(small)
        href = dict(hdr, CDELT1=4.44, CDELT2=4.44, CRVAL1=hs["CRVAL1"], CRVAL2=hs["CRVAL2"],
                    DSUN_OBS=hl["DSUN_OBS"], CRLN_OBS=hl["CRLN_OBS"], CRLT_OBS=hl["CRLT_OBS"])
        gpu_handle.prepare_reference_helioprojective(img, href, hs, 2)
        out.append(gpu_handle.get_reference_on_grid((hs["NAXIS2"], hs["NAXIS1"]), np.float32))
        grid = _lib.Grid(H.CARR_LON, H.CARR_LAT, SHAPE)
        gpu_handle.prepare_reference_carrington(img, href, grid, 1.004, 2)
        out.append(gpu_handle.get_reference_on_grid((grid.n_lat, grid.n_lon), np.float64))
        res[name] = out[1:]
    for a, b in zip(res["host"], res["gpu"]):
        assert np.array_equal(a, b, equal_nan=True)
    assert np.isfinite(res["gpu"][0]).any() and np.isfinite(res["gpu"][2]).any()
    # interior pixels of the order-1 read-back ARE the decoded pixels (weights 1, 0)
    rb, d64 = res["gpu"][0], np.asarray(dec, dtype=np.float64)
    m = np.isfinite(rb)
    assert m.sum() > 0.5 * rb.size and np.array_equal(rb[m], d64[m])


def test_alignment_on_a_tile_compressed_file_equals_the_decoded_image(tmp_path, monkeypatch):
    """`Alignment` with a tile-compressed FITS file as the image to align (the EUI case): compressed upload + GPU decode
    (default) and host decode (COREG_RAW_FITS=0) give the same correlation array; `write_corrected_fits` then keeps the
    compressed stream."""
    import os
    from tests.conftest import GOLDEN
    from euispice_coreg_amd.hdrshift import Alignment
    from euispice_coreg_amd.utils import fits_io
    src = os.path.join(GOLDEN, "compressed", "rice_f32_big.fits")
    _, _, large, hl, _ = H.scene()
    hl = dict(hl, CRVAL1=-310.0, CRVAL2=420.0)   # the reference looks where the compressed image's header points
    lag = np.arange(-4.0, 4.5, 2.0)
    out = {}
    for raw in ("1", "0"):
        monkeypatch.setenv("COREG_RAW_FITS", raw)
        A = Alignment((large, hl), src, lag, lag, [0], [0], [0], parallelism=True, force_crota_0=False)
        out[raw] = A.align_using_helioprojective(return_type="corr")
        assert isinstance(A.data_small, fits_io.CompressedImage) == (raw == "1")
    assert np.array_equal(out["1"], out["0"], equal_nan=True) and np.isfinite(out["1"]).any()
    monkeypatch.setenv("COREG_RAW_FITS", "1")
    res = Alignment((large, hl), src, lag, lag, [0], [0], [0], parallelism=True).align_using_helioprojective()
    dst = str(tmp_path / "corrected.fits")
    res.write_corrected_fits([-1], dst)
    d, h = fits_io.read_image(dst, -1)
    assert np.array_equal(d, np.asarray(fits_io.open_compressed(src, -1)), equal_nan=True)
    assert h["CRVAL1"] == pytest.approx(-310.0 + res.shift_arcsec[0])


@pytest.mark.parametrize("kind", ["float32 dithered (level 2)", "uint16 (level 1)"])
def test_full_size_compressed_image_round_trip_through_the_gpu(gpu_handle, tmp_path, kind):
    """A 2048 x 2048 image in the format EUI files have, written by this package (csrc/riceenc.hpp), decoded on the GPU:
    the size-independent properties of a codec at full size -- encode -> GPU decode gives the input back (integers exactly,
    quantized floats to half a step, NaN kept), the GPU's pixels are the host decoder's bit for bit, and a sweep on the
    compressed image equals the sweep on the decoded pixels."""
    from euispice_coreg_amd import _lib, synthetic
    from euispice_coreg_amd.utils import fits_io
    small, hs, large, hl, _ = synthetic.make_scene()
    assert small.shape == (2048, 2048)
    if kind.startswith("float32"):
        img = small.astype(np.float32)
        assert np.isnan(img).any()
        kw = dict(quantize="SUBTRACTIVE_DITHER_2", dither0=4321)
    else:
        img = np.clip(np.nan_to_num(small, nan=0.0) * 8.0, 0, 65535).astype(np.uint16)
        kw = {}
    p = str(tmp_path / "full.fits")
    info = fits_io.write_compressed_image(p, img, hs, **kw)
    ci = fits_io.open_compressed(p, -1)
    assert ci.on_gpu and ci.n_tiles == 2048 and info["compressed_bytes"] < 0.75 * img.nbytes
    host = ci.decode()
    # read the resident pixels back: an order-1 resample onto the image's own grid has weights (1, 0) on interior pixels
    hdr = dict(hs)
    gpu_handle.set_small(ci)
    rb = gpu_handle.resample_helioprojective(hdr, hdr, order=1, dtype=np.float64)
    m = np.isfinite(rb)
    assert m.mean() > 0.9
    # (at this size the pixel coordinates of the read-back are integers to 1e-13 only: the neighbour's weight is not an
    # exact zero.  A wrong decode is off by a quantization step, 2e-1, or by thousands.)
    d = np.abs(rb[m] - host.astype(np.float64)[m])
    assert d.max() <= 1e-8, f"GPU decode differs from the host decode by up to {d.max()} on {(d > 1e-8).sum()} pixels"
    if img.dtype.kind == "f":
        ok = np.isfinite(img)
        assert np.array_equal(np.isnan(host), ~ok)
        assert np.abs(host[ok].astype(np.float64) - img[ok]).max() <= 0.5 * info["scale"] + np.spacing(np.float32(img[ok].max()))
    else:
        assert np.array_equal(host, img.astype(np.float64))
    # sweeps: compressed upload against the decoded pixels
    grid = _lib.Grid(H.CARR_LON, H.CARR_LAT, (512, 512))
    gpu_handle.prepare_reference_carrington(large, hl, grid, 1.004, 2)
    lags = _lib.LagSet(np.arange(-6.0, 7.0, 3.0), np.arange(-6.0, 7.0, 3.0), None, None, None)
    a = gpu_handle.sweep_carrington(hs, grid, 1.004, lags)
    gpu_handle.set_small(fits_io.native_pixels(host))
    b = gpu_handle.sweep_carrington(hs, grid, 1.004, lags)
    assert np.array_equal(a, b, equal_nan=True) and np.isfinite(a).all()


def test_corrupt_compressed_streams_are_refused_by_the_gpu_decoder(gpu_handle):
    """What the sanitizer harness proves of the shared decoder on the host (tests/test_rice_sanitizers_cpu.py) seen from
    the GPU side: a truncated tile, a tile table pointing outside the heap and bit-flipped streams end in COREG_EINVAL or
    in exactly the pixels the host decoder produces from the same bytes -- never in a fault -- and the handle keeps
    working afterwards."""
    import copy
    import os
    from tests.conftest import GOLDEN
    from euispice_coreg_amd import _lib
    from euispice_coreg_amd.utils import fits_io
    path = os.path.join(GOLDEN, "compressed", "rice_f32_big.fits")
    good = fits_io.open_compressed(path, -1)
    hdr = dict(good.header, PC1_1=1.0, PC1_2=0.0, PC2_1=0.0, PC2_2=1.0, CROTA=0.0, CTYPE1="HPLN-TAN", CTYPE2="HPLT-TAN")

    def variant():
        ci = copy.copy(good)
        ci.tile_nbytes, ci.tile_offset = good.tile_nbytes.copy(), good.tile_offset.copy()
        ci._heap = np.array(good._heap)  # (a private, writable copy of the heap)
        return ci

    ci = variant()
    ci.tile_nbytes[17] = 5
    with pytest.raises(_lib.CoregError, match="truncated or corrupt"):
        gpu_handle.set_small(ci)
    ci = variant()
    ci.tile_offset[3] = ci._heap.size - 2
    with pytest.raises(_lib.CoregError, match="truncated or corrupt"):
        gpu_handle.set_small(ci)
    rng = np.random.default_rng(99)
    flagged = same = 0
    for _ in range(24):
        ci = variant()
        for _k in range(int(rng.integers(1, 6))):
            ci._heap[int(rng.integers(0, ci._heap.size))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        out = np.empty(ci.shape, dtype=np.float32)
        status = _lib.decode_tiled_host(ci, out)
        try:
            gpu_handle.set_small(ci)
        except _lib.CoregError:
            assert status.any(), "the GPU flagged a stream the host decoder accepts"
            flagged += 1
            continue
        assert not status.any(), "the host decoder flagged a stream the GPU accepts"
        rb = gpu_handle.resample_helioprojective(hdr, hdr, order=1, dtype=np.float64)
        m = np.isfinite(rb) & np.isfinite(out)
        assert np.array_equal(rb[m], out.astype(np.float64)[m])
        same += 1
    assert flagged + same == 24 and same > 0
    # the handle is intact
    gpu_handle.set_small(good)
    rb = gpu_handle.resample_helioprojective(hdr, hdr, order=1, dtype=np.float64)
    m = np.isfinite(rb)
    assert np.array_equal(rb[m], np.asarray(good).astype(np.float64)[m])


@pytest.mark.parametrize("tile", [(150, 40), (300, 200), (7, 3)])
def test_compressed_tiles_of_any_size_decode_on_the_gpu(gpu_handle, tmp_path, tile):
    """Tiles beyond what the kernel stages in LDS (4096 pixels, 16 KB of stream) take its direct path -- one lane decodes
    straight from the heap into the image -- and tiny ragged tiles the staged one: same pixels as the host decoder."""
    from euispice_coreg_amd.utils import fits_io
    small, hs, _, _, _ = H.scene(small_n=96)
    img = np.resize(small.astype(np.float32), (200, 300))
    hdr = dict(hs, NAXIS1=300, NAXIS2=200, PC1_1=1.0, PC1_2=0.0, PC2_1=0.0, PC2_2=1.0, CROTA=0.0)
    p = str(tmp_path / "t.fits")
    fits_io.write_compressed_image(p, img, hdr, tile=tile, quantize="SUBTRACTIVE_DITHER_1", dither0=9000)
    ci = fits_io.open_compressed(p, -1)
    assert ci.on_gpu and ci.ztile == tile
    host = ci.decode()
    gpu_handle.set_small(ci)
    rb = gpu_handle.resample_helioprojective(hdr, hdr, order=1, dtype=np.float64)
    m = np.isfinite(rb)
    assert m.mean() > 0.8 and np.abs(rb[m] - host.astype(np.float64)[m]).max() <= 1e-9


def test_upload_stream_and_upload_thread_give_the_same_map(gpu_handle):
    """Round 5: the image to align goes up on the handle's upload stream (option "overlap_upload", default on) and,
    opt-in, through the handle's upload THREAD ("async_upload": the call returns once the upload is queued; the pixel
    buffer must outlive the next reader).  Same maps as the one-stream hand-over -- float32 arrays, raw BITPIX = -32 data
    units -- also when a new image replaces one that an earlier sweep has just read, and when the reference preparation
    sits between the upload and the sweep."""
    import tempfile, os
    from euispice_coreg_amd import _lib
    from euispice_coreg_amd.utils import fits_io
    small, hs, large, hl, _ = H.scene(small_n=160, large_n=224)
    s32 = small.astype(np.float32)
    other = (small * 1.03 + 2.0).astype(np.float32)
    grid = _lib.Grid(H.CARR_LON, H.CARR_LAT, (96, 80))
    ls = _lib.LagSet(17.0 + 2.0 * (np.arange(9) - 4), -9.0 + 2.0 * (np.arange(8) - 4), None, None, None)
    d = tempfile.mkdtemp(prefix="coreg_async_")
    p = os.path.join(d, "s.fits")
    fits_io.write_images(p, [(None, {}), (other, hs)])
    raw = fits_io.open_raw(p, -1)

    def call(img):
        gpu_handle.set_small(img)
        gpu_handle.prepare_reference_carrington(large, hl, grid, 1.004, 2)
        return gpu_handle.sweep_carrington(hs, grid, 1.004, ls)

    try:
        gpu_handle.set_option("overlap_upload", 0)
        want = [call(s32), call(other)]
        gpu_handle.set_option("overlap_upload", 1)
        for async_upload in (0, 1):
            gpu_handle.set_option("async_upload", async_upload)
            for _ in range(3):
                assert np.array_equal(call(s32), want[0], equal_nan=True)
                assert np.array_equal(call(other), want[1], equal_nan=True)
                assert np.array_equal(call(raw), want[1], equal_nan=True)
            # readers other than the sweep join too: threshold, pivots, synchronize
            gpu_handle.set_small(s32)
            assert gpu_handle.threshold_small(None, None) == int(np.isfinite(s32).sum())
            gpu_handle.set_small(other)
            assert np.isfinite(gpu_handle.get_pivots()).all()
            gpu_handle.set_small(s32)
            gpu_handle.synchronize()
    finally:
        gpu_handle.set_option("async_upload", 0)
        gpu_handle.set_option("overlap_upload", 1)
        raw.close()
