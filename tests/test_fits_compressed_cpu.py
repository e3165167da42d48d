"""
Tile-compressed FITS images (what EUI level-1 / level-2 files hold; the reference reads them through astropy's
CompImageHDU) without astropy: utils/fits_io.py parses the table, csrc/ricecomp.hpp -- cfitsio's RICE_1 codec and float
dequantization restated -- decodes it.  Pinned against what astropy 4.3.1 (its bundled cfitsio) decodes from the same
files: tests/golden/compressed/*.fits and compressed_golden.npz, written by tests/golden/make_golden_compressed.py.
The host decoder tested here is the very function the GPU runs one thread per tile of (tests/test_gpu_upload.py).
"""
import glob
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN

FILES = sorted(glob.glob(os.path.join(GOLDEN, "compressed", "*.fits")))
NAMES = [os.path.basename(p)[:-5] for p in FILES]


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "compressed_golden.npz"))


def test_fixture_set_is_complete():
    assert len(FILES) == 17 and "rice_f32_dither2_zeros" in NAMES and "gzip2_f32_lossless" in NAMES


@pytest.mark.parametrize("name", NAMES)
def test_decoded_pixels_equal_astropy_bit_for_bit(gold, name):
    from euispice_coreg_amd.utils import fits_io
    path = os.path.join(GOLDEN, "compressed", name + ".fits")
    ci = fits_io.open_compressed(path, -1)
    assert ci is not None and fits_io.open_raw(path, -1) is None and fits_io.open_compressed(path, 0) is None
    want = gold[name + "/data"]
    got = np.asarray(ci)
    assert got.shape == want.shape == ci.shape
    # float images: identical bits (NaN where astropy has NaN); integer images: identical values -- with BZERO (the
    # unsigned-16 convention) this reader returns float64(stored) * BSCALE + BZERO where astropy returns uint16
    assert np.array_equal(got.astype(np.float64), want.astype(np.float64), equal_nan=True)
    if want.dtype.kind == "f":
        assert got.dtype == want.dtype and np.array_equal(np.signbit(got), np.signbit(want))
    elif name != "rice_u16":
        assert got.dtype == want.dtype
    # the same through the general readers
    data, hdr = fits_io.read_image(path, -1)
    assert np.array_equal(data, got, equal_nan=True)
    hdr2 = fits_io.read_header(path, "IMAGE")
    assert hdr == hdr2 == ci.header
    # the header is the IMAGE's (as astropy presents it), not the table's
    assert hdr["NAXIS1"] == want.shape[1] and hdr["NAXIS2"] == want.shape[0] and hdr["BITPIX"] == ci.zbitpix
    assert hdr["CRVAL1"] == -310.0 and hdr["CUNIT1"] == "arcsec" and hdr["DATE-AVG"] == "2022-03-17T09:50:45.277"
    assert not any(k.startswith(("TFORM", "TTYPE", "ZNAXIS", "ZTILE", "ZNAME", "ZVAL")) or k in ("ZIMAGE", "ZCMPTYPE", "PCOUNT")
                   for k in hdr)
    astropy_keys = set(gold[name + "/header_keys"].tolist()) - {"COMMENT", "HISTORY", ""}
    assert {k for k in astropy_keys - set(hdr) if not k.startswith("Z")} <= {"PCOUNT", "GCOUNT", "XTENSION", "EXTEND",
                                                                              "SIMPLE", "CHECKSUM", "DATASUM"}
    # which of them the GPU can take as they are
    expect_gpu = name.startswith("rice_") and bool((ci.tile_nbytes > 0).all())
    assert ci.on_gpu == expect_gpu
    lo = fits_io.load_for_upload(path, -1)
    assert isinstance(lo[0], fits_io.CompressedImage) and fits_io.native_pixels(lo[0]) is lo[0]


def test_tiles_stored_gzipped_are_found_and_patched(gold):
    """cfitsio stores a tile it cannot quantize (a constant row) gzipped in a second column: the Rice decoder reports the
    tile as not Rice-coded and the reader fills it in from there."""
    from euispice_coreg_amd import _lib
    from euispice_coreg_amd.utils import fits_io
    ci = fits_io.open_compressed(os.path.join(GOLDEN, "compressed", "rice_f32_const_tile.fits"), -1)
    assert (ci.tile_nbytes == 0).sum() == 1 and ci.gzip_nbytes[12] > 0 and not ci.on_gpu
    out = np.full(ci.shape, -1.0, dtype=np.float32)
    status = _lib.decode_tiled_host(ci, out)
    assert status.tolist() == [0] * 12 + [2] + [0] * 57 and (out[12] == -1.0).all()  # left untouched
    assert np.array_equal(np.asarray(ci)[12], np.full(93, 250.0, dtype=np.float32))
    assert np.array_equal(np.asarray(ci), gold["rice_f32_const_tile/data"], equal_nan=True)


def test_truncated_and_corrupt_streams_are_reported_not_read_past(tmp_path, gold):
    from euispice_coreg_amd import _lib
    from euispice_coreg_amd.utils import fits_io
    ci = fits_io.open_compressed(os.path.join(GOLDEN, "compressed", "rice_i16.fits"), -1)
    good = np.asarray(ci)
    ci.tile_nbytes = ci.tile_nbytes.copy()
    ci.tile_nbytes[5] = 7  # the stream of tile 5 ends early
    with pytest.raises(IOError):
        ci.decode()
    out = np.empty(ci.shape, dtype=np.float64)
    status = _lib.decode_tiled_host(ci, out)
    assert status[5] == 1 and status.sum() == 1 and np.array_equal(np.delete(out, 5, axis=0), np.delete(good, 5, axis=0))
    ci.tile_offset = ci.tile_offset.copy()
    ci.tile_offset[9] = ci._heap.size - 3  # points past the heap
    assert _lib.decode_tiled_host(ci, out)[9] == 1
    # inconsistent parameters are refused before anything is decoded
    ci.bytepix = 3
    with pytest.raises(_lib.CoregError):
        _lib.decode_tiled_host(ci, out)


def test_write_corrected_fits_keeps_the_compressed_stream_and_patches_the_header(tmp_path, gold):
    """utils/Util.py:106-159 on a tile-compressed input: the corrected pointing keywords go into the table's header, the
    table and its heap (the compressed pixels) are copied as they are, every other HDU byte for byte."""
    from euispice_coreg_amd.hdrshift import AlignmentResults
    from euispice_coreg_amd.utils import fits_io
    from tests.test_oracle_golden import REF_CORR
    src = os.path.join(GOLDEN, "compressed", "rice_f32_nan.fits")
    R = AlignmentResults(REF_CORR, np.arange(15, 26, 1), np.arange(5, 11, 1), None, [0], [0.75], "arcsec",
                         image_to_align_path=src)
    out = str(tmp_path / "corrected.fits")
    R.write_corrected_fits([-1], out)
    a, b = open(src, "rb").read(), open(out, "rb").read()
    assert a[:2880] == b[:2880]                                    # primary HDU
    (h_in, sp_in), (h_out, sp_out) = fits_io._scan(src), fits_io._scan(out)
    n = sp_in[1][1]
    assert sp_out[1][1] == n and a[sp_in[1][0]:sp_in[1][0] + n] == b[sp_out[1][0]:sp_out[1][0] + n]   # table + heap
    d_out, hdr = fits_io.read_image(out, -1)
    assert np.array_equal(d_out, gold["rice_f32_nan/data"], equal_nan=True)
    assert hdr["CRVAL1"] == pytest.approx(-310.0 + R.shift_arcsec[0], rel=1e-14)
    assert hdr["CRVAL2"] == pytest.approx(420.0 + R.shift_arcsec[1], rel=1e-14)
    assert hdr["CROTA"] == pytest.approx(3.0 + 0.75, rel=1e-14) and hdr["PC1_1"] == pytest.approx(np.cos(np.deg2rad(3.75)))
    # structure and compression keywords, comments included, are the input's
    for k in ("ZCMPTYPE", "ZTILE1", "TFORM1", "ZDITHER0", "NAXIS1", "PCOUNT"):
        assert h_out[1][k] == h_in[1][k]
    assert b"/ compression algorithm" in b[2880:sp_out[1][0]]
    # the corrected cards keep their own comments too (astropy's header update does)
    for card in (a[2880 + i:2880 + i + 80] for i in range(0, sp_in[1][0] - 2880, 80)):
        if card[:8].rstrip() in (b"CRVAL1", b"CRVAL2") and b" /" in card[10:]:
            comment = card[10:].split(b"/", 1)[1].rstrip()
            assert any(o[:8] == card[:8] and o.rstrip().endswith(comment)
                       for o in (b[2880 + i:2880 + i + 80] for i in range(0, sp_out[1][0] - 2880, 80)))
    assert R.return_corrected_header(-1)["CRVAL1"] == hdr["CRVAL1"]


def test_rice_one_is_the_old_name_of_rice_1(tmp_path):
    """ZCMPTYPE = 'RICE_ONE' (what early cfitsio versions wrote; cfitsio and astropy read both names)."""
    from euispice_coreg_amd.utils import fits_io
    src = os.path.join(GOLDEN, "compressed", "rice_i16.fits")
    b = open(src, "rb").read()
    assert b.count(b"'RICE_1  '") == 1
    p = str(tmp_path / "old_name.fits")
    open(p, "wb").write(b.replace(b"'RICE_1  '", b"'RICE_ONE'"))
    ci = fits_io.open_compressed(p, -1)
    assert ci.cmptype == "RICE_1" and ci.on_gpu
    assert np.array_equal(np.asarray(ci), np.asarray(fits_io.open_compressed(src, -1)))
