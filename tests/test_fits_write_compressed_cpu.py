"""
The writing side of the tile-compressed FITS support: utils/fits_io.write_compressed_image -> csrc/riceenc.hpp (cfitsio's
RICE_1 encoder `fits_rcomp*` and float quantization restated; the reference writes such files through astropy's
CompImageHDU, utils/Util.py:137-138).  Pinned three ways:
  * golden: tests/golden/written/*.fits were written by this package and decoded by astropy 4.3.1 / its cfitsio
    (tests/golden/make_golden_written.py) -- a third party reads them as intended; re-writing them gives the same bytes;
  * round trip, the size-independent property of a codec: decode(encode(q)) == q for every pixel width (differences that
    wrap, verbatim blocks, all-zero blocks, ragged tiles), |dequantize(quantize(v)) - v| <= ZSCALE / 2, NaN and (dither 2)
    exact zeros kept;
  * the encoder never writes past the capacity it was given.
Host code only (no GPU); the GPU decodes what this writes in tests/test_gpu_upload.py.
"""
import glob
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN

FILES = sorted(glob.glob(os.path.join(GOLDEN, "written", "*.fits")))
NAMES = [os.path.basename(p)[:-5] for p in FILES]


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "written_golden.npz"))


def _cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_written", os.path.join(GOLDEN, "make_golden_written.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.cases(), mod.HEADER


def test_fixture_set_is_complete():
    assert len(FILES) == 13 and "w_i32_noise" in NAMES and "w_f32_nan_tiles2d" in NAMES


@pytest.mark.parametrize("name", NAMES)
def test_written_files_are_what_astropy_reads(gold, name, tmp_path):
    from euispice_coreg_amd.utils import fits_io
    cases, header = _cases()
    a, kw = cases[name]
    assert np.array_equal(a, gold[name + "/input"], equal_nan=True)  # (the generator's inputs are the committed ones)
    # the encoder is deterministic: the same input gives the committed file, byte for byte
    p = str(tmp_path / (name + ".fits"))
    info = fits_io.write_compressed_image(p, a, header, **kw)
    committed = os.path.join(GOLDEN, "written", name + ".fits")
    assert open(p, "rb").read() == open(committed, "rb").read()
    # what astropy / cfitsio decoded from the committed file is what this package decodes from it ...
    ci = fits_io.open_compressed(committed, -1)
    got = np.asarray(ci)
    want = gold[name + "/astropy"]
    finite = np.isfinite(a) if a.dtype.kind == "f" else np.ones(a.shape, bool)
    assert np.array_equal(got[finite].astype(np.float64), want[finite].astype(np.float64))
    # ... except at undefined pixels, where astropy 4.3.1 does not look for nulls (NULL_VALUE * ZSCALE + ZZERO)
    assert np.isnan(got[~finite]).all() and (want[~finite] < -1e8).all()
    # and it is the input: integers exactly, floats to half a quantization step
    if a.dtype.kind == "f":
        assert got.dtype == a.dtype
        slack = np.spacing(np.abs(a[finite]).max().astype(a.dtype)) if a.dtype == np.float32 else 0.0
        assert np.abs(got[finite].astype(np.float64) - a[finite]).max() <= 0.5 * info["scale"] * (1 + 1e-9) + slack
        if kw.get("quantize") == "SUBTRACTIVE_DITHER_2":
            assert (a == 0).any() and (got[a == 0] == 0).all()
    else:
        assert np.array_equal(got, a)
    assert ci.on_gpu and ci.header["CRVAL1"] == -310.0 and ci.header["CTYPE1"] == "HPLN-TAN"
    assert ci.header["BITPIX"] == {"f4": -32, "f8": -64, "i2": 16, "u2": 16, "u1": 8, "i4": 32}[a.dtype.str[1:]]


@pytest.mark.parametrize("bytepix, lo, hi", [(1, 0, 256), (2, -32768, 32768), (4, -2 ** 31, 2 ** 31)])
@pytest.mark.parametrize("kind", ["noise", "smooth", "spikes", "constant"])
def test_integers_round_trip_for_every_pixel_width(bytepix, lo, hi, kind):
    from euispice_coreg_amd import _lib
    from euispice_coreg_amd.utils import fits_io
    rng = np.random.default_rng(bytepix * 10 + len(kind))
    ny, nx = 37, 211  # ragged against 32-pixel blocks and against the tiles below
    if kind == "noise":
        q = rng.integers(lo, hi, size=(ny, nx))
    elif kind == "smooth":
        x = np.arange(nx)[None, :] + np.arange(ny)[:, None]
        q = lo + ((hi - lo) * (0.5 + 0.4 * np.sin(x / 17.0))).astype(np.int64) + rng.integers(-3, 4, size=(ny, nx))
    elif kind == "spikes":
        q = np.full((ny, nx), lo + (hi - lo) // 2)
        q[rng.random((ny, nx)) < 0.03] = hi - 1
        q[rng.random((ny, nx)) < 0.03] = lo
    else:
        q = np.full((ny, nx), hi - 1)
    q = np.clip(q, lo, hi - 1).astype(np.int32)
    for tile in ((nx, 1), (64, 8), (nx, ny), (7, 5)):
        heap, nbytes, offs, zs, zz = _lib.encode_tiled_host(q, tile, bytepix)
        assert zs is None and (nbytes > 0).all() and offs[0] == 0 and np.array_equal(offs[1:], np.cumsum(nbytes)[:-1])

        class Img:  # (the attributes _lib._fits_tiled reads from a CompressedImage)
            pass
        ci = Img()
        ci.zbitpix, ci.shape, ci.ztile, ci.blocksize, ci.bytepix = 8 * bytepix, (ny, nx), tile, 32, bytepix
        ci.quantize, ci.dither0, ci.has_blank, ci.blank, ci.bscale, ci.bzero = 0, 1, False, 0, 1, 0
        ci._heap, ci.tile_offset, ci.tile_nbytes, ci.n_tiles = heap, offs, nbytes, len(nbytes)
        ci.zscale = ci.zzero = None
        ci.zscale0, ci.zzero0 = 1.0, 0.0
        out = np.empty((ny, nx), dtype=np.float64)
        status = _lib.decode_tiled_host(ci, out)
        assert not status.any()
        assert np.array_equal(out, q.astype(np.float64))
        if kind == "constant" and tile == (nx, ny):
            assert nbytes.sum() < q.size // 8  # all-zero blocks cost FSBITS bits each
    del fits_io


@pytest.mark.parametrize("quantize", ["NO_DITHER", "SUBTRACTIVE_DITHER_1", "SUBTRACTIVE_DITHER_2"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_floats_round_trip_within_half_a_step(tmp_path, quantize, dtype):
    from euispice_coreg_amd.utils import fits_io
    rng = np.random.default_rng(5)
    a = (1000.0 * rng.random((61, 150)) ** 3 - 20.0).astype(dtype)
    a[rng.random(a.shape) < 0.02] = np.nan
    a[7, :] = np.nan            # a tile without a finite pixel
    a[20:23, 10:90] = 0.0
    p = str(tmp_path / "f.fits")
    for tile, seed in (((150, 1), 1), ((40, 13), 9990)):  # (seed 9990: the walk through the random sequence wraps)
        info = fits_io.write_compressed_image(p, a, {"CRVAL1": 1.0}, tile=tile, quantize=quantize, scale=0.37, dither0=seed)
        ci = fits_io.open_compressed(p, -1)
        got = ci.decode()
        assert got.dtype == dtype and np.array_equal(np.isnan(got), np.isnan(a))
        ok = np.isfinite(a)
        slack = np.spacing(np.float32(1000.0)) if dtype == np.float32 else 1e-12
        assert np.abs(got[ok].astype(np.float64) - a[ok].astype(np.float64)).max() <= 0.5 * 0.37 + slack
        if quantize == "SUBTRACTIVE_DITHER_2":
            assert (got[20:23, 10:90] == 0).all()
        assert info["scale"] == 0.37 and ci.dither0 == seed and ci.quantize == fits_io._QUANTIZE[quantize]
    # the default scale: the image's noise / quantize_level
    noisy = (100.0 + 5.0 * rng.standard_normal((64, 64))).astype(dtype)
    info = fits_io.write_compressed_image(p, noisy, {}, quantize=quantize)
    assert 0.8 * 5.0 / 16 < info["scale"] < 1.25 * 5.0 / 16


def test_encoder_respects_its_capacity_and_rejects_bad_input():
    import ctypes as C
    from euispice_coreg_amd import _lib
    lib = _lib.load_library()
    q = np.random.default_rng(0).integers(-2 ** 31, 2 ** 31, size=(8, 64)).astype(np.int32)
    nt = 8
    nbytes, offs = np.zeros(nt, np.int32), np.zeros(nt, np.int64)
    used = C.c_longlong(0)
    guard = 77
    for cap in (0, 3, 100, 1000):
        heap = np.full(cap + 16, guard, dtype=np.uint8)
        rc = lib.coreg_encode_tiled_host(q.ctypes.data, 2, 8, 64, 64, 1, 4, 32, 0, 1, 1.0, heap.ctypes.data, cap, nbytes.ctypes.data,
                                         offs.ctypes.data, None, None, C.byref(used))
        assert rc == _lib.COREG_ENOMEM and (heap[cap:] == guard).all()
    with pytest.raises(_lib.CoregError):  # a range that does not fit 32-bit integers at this scale
        _lib.encode_tiled_host(np.array([[0.0, 1e12]], dtype=np.float64), (2, 1), 4, 32, 2, 1, 1.0)
    with pytest.raises(TypeError):
        _lib.encode_tiled_host(np.zeros((2, 2), np.int16), (2, 1))
    with pytest.raises(_lib.CoregError):
        _lib.encode_tiled_host(np.zeros((2, 2), np.int32), (2, 1), bytepix=3)
