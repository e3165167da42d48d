"""
ctypes binding of libcoreg_hip.so (C ABI: include/coreg_hip.h).

The library is the product path: there is NO CPU fallback.  If the shared object has not been
built (`python -c "import __graft_entry__ as g; g.build()"` or `make -C euispice_coreg_amd/csrc`)
importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COREG_HIP_LIB", os.path.join(_HERE, "libcoreg_hip.so"))  # env override: kernel experiments

COREG_OK = 0
COREG_EINVAL, COREG_EHIP, COREG_ESTATE, COREG_ENOTIMPL, COREG_ENOMEM = -1, -2, -3, -4, -5
COREG_F32, COREG_F64 = 0, 1
PROJ_TAN, PROJ_CAR = 0, 1
METHOD_CORRELATION, METHOD_RESIDUS = 0, 1
CDELT_INTENDED, CDELT_REFERENCE = 0, 1


class CoregError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libcoreg_hip error {code}: {msg}")
        self.code = code


class Wcs2d(C.Structure):
    _fields_ = [("naxis1", C.c_int32), ("naxis2", C.c_int32),
                ("crpix1", C.c_double), ("crpix2", C.c_double),
                ("crval1", C.c_double), ("crval2", C.c_double),
                ("cdelt1", C.c_double), ("cdelt2", C.c_double),
                ("pc1_1", C.c_double), ("pc1_2", C.c_double), ("pc2_1", C.c_double), ("pc2_2", C.c_double),
                ("crota", C.c_double), ("unit_to_deg", C.c_double), ("lonpole", C.c_double),
                ("dsun_obs", C.c_double), ("crln_obs", C.c_double), ("crlt_obs", C.c_double),
                ("latpole", C.c_double), ("proj", C.c_int32), ("reserved", C.c_int32)]


class Lags(C.Structure):
    _fields_ = [("crval1", C.POINTER(C.c_double)), ("n_crval1", C.c_int32),
                ("crval2", C.POINTER(C.c_double)), ("n_crval2", C.c_int32),
                ("cdelt1", C.POINTER(C.c_double)), ("n_cdelt1", C.c_int32),
                ("cdelt2", C.POINTER(C.c_double)), ("n_cdelt2", C.c_int32),
                ("crota", C.POINTER(C.c_double)), ("n_crota", C.c_int32)]


class CarrGrid(C.Structure):
    _fields_ = [("lon0", C.c_double), ("lon1", C.c_double), ("n_lon", C.c_int32),
                ("lat0", C.c_double), ("lat1", C.c_double), ("n_lat", C.c_int32),
                ("lat_cos", C.POINTER(C.c_float)), ("lat_sin", C.POINTER(C.c_float))]


class Stats(C.Structure):
    _fields_ = [("sweep_kernel_ms", C.c_double), ("precompute_ms", C.c_double), ("total_gpu_ms", C.c_double),
                ("n_lags", C.c_int64), ("n_grid_points", C.c_int64), ("n_active_points", C.c_int64),
                ("n_sweep_launches", C.c_int64), ("small_is_f32", C.c_int32), ("used_lds", C.c_int32)]


class FitsPixels(C.Structure):
    _fields_ = [("data", C.c_void_p), ("bitpix", C.c_int32), ("reserved", C.c_int32), ("bscale", C.c_double),
                ("bzero", C.c_double)]


class FitsTiled(C.Structure):
    _fields_ = [("heap", C.c_void_p), ("heap_bytes", C.c_int64), ("tile_offset", C.c_void_p), ("tile_nbytes", C.c_void_p),
                ("n_tiles", C.c_int32), ("zbitpix", C.c_int32), ("naxis1", C.c_int32), ("naxis2", C.c_int32),
                ("ztile1", C.c_int32), ("ztile2", C.c_int32), ("blocksize", C.c_int32), ("bytepix", C.c_int32),
                ("quantize", C.c_int32), ("dither0", C.c_int32), ("has_blank", C.c_int32), ("blank", C.c_int32),
                ("zscale", C.c_void_p), ("zzero", C.c_void_p), ("zscale0", C.c_double), ("zzero0", C.c_double),
                ("bscale", C.c_double), ("bzero", C.c_double)]


def _is_tiled(img):
    """utils.fits_io.CompressedImage that the GPU can decode as it is (duck-typed)."""
    return hasattr(img, "tile_nbytes") and hasattr(img, "ztile") and getattr(img, "on_gpu", False)


def _fits_tiled(ci):
    """(struct, arrays that must stay alive during the call) of a CompressedImage."""
    off = np.ascontiguousarray(ci.tile_offset, dtype=np.int64)
    nb = np.ascontiguousarray(ci.tile_nbytes, dtype=np.int32)
    keep = [off, nb, ci._heap]
    zs = zz = None
    if ci.zscale is not None and ci.zzero is not None:
        zs = np.ascontiguousarray(ci.zscale, dtype=np.float64)
        zz = np.ascontiguousarray(ci.zzero, dtype=np.float64)
        keep += [zs, zz]
    t = FitsTiled(ci._heap.ctypes.data, int(ci._heap.size), off.ctypes.data, nb.ctypes.data, int(ci.n_tiles), int(ci.zbitpix),
                  int(ci.shape[1]), int(ci.shape[0]), int(ci.ztile[0]), int(ci.ztile[1]), int(ci.blocksize), int(ci.bytepix),
                  int(ci.quantize), int(ci.dither0), int(bool(ci.has_blank)), int(ci.blank),
                  zs.ctypes.data if zs is not None else None, zz.ctypes.data if zz is not None else None,
                  float(ci.zscale0), float(ci.zzero0), float(ci.bscale), float(ci.bzero))
    return t, keep


def decode_tiled_host(ci, out):
    """Rice-decode a CompressedImage into `out` ([ny, nx] float32 for ZBITPIX = -32, else float64) on the host (the code
    the GPU runs, a few threads over the tiles).  Returns the per-tile status (0 ok, 1 corrupt, 2 not Rice-coded)."""
    t, keep = _fits_tiled(ci)
    status = np.zeros(ci.n_tiles, dtype=np.int32)
    rc = load_library().coreg_decode_tiled_host(C.byref(t), out.ctypes.data, COREG_F32 if out.dtype == np.float32 else COREG_F64,
                                                status.ctypes.data)
    if rc != COREG_OK:
        raise CoregError(rc, "coreg_decode_tiled_host: bad arguments (unsupported tiling / codec parameters)")
    del keep
    return status


def encode_tiled_host(pixels, tile, bytepix=4, blocksize=32, quantize=0, dither0=1, scale=1.0):
    """Rice-encode an image tile by tile on the host (cfitsio's encoder restated, csrc/riceenc.hpp): `pixels` float32 /
    float64 (quantized first: `quantize` 1 NO_DITHER, 2 / 3 SUBTRACTIVE_DITHER_1 / _2, ZSCALE = `scale`, ZZERO = each tile's
    minimum) or int32 (the stored integers of an integer image, BYTEPIX 1 / 2 / 4).
    Returns (heap bytes, tile_nbytes, tile_offset, zscale or None, zzero or None)."""
    a = np.ascontiguousarray(pixels)
    if a.ndim != 2:
        raise ValueError("image must be 2-D")
    ny, nx = a.shape
    tx, ty = int(tile[0]), int(tile[1])
    nt = (-(-nx // tx)) * (-(-ny // ty))
    if a.dtype == np.float32:
        dt = COREG_F32
    elif a.dtype == np.float64:
        dt = COREG_F64
    elif a.dtype == np.int32:
        dt = 2  # COREG_I32
    else:
        raise TypeError("pixels must be float32, float64 or int32 (stored integers)")
    is_float = dt != 2
    n_blocks = nt * (-(-(tx * ty) // int(blocksize)) + 1)
    cap = a.size * int(bytepix) + n_blocks + 8 * nt + 64
    heap = np.empty(cap, dtype=np.uint8)
    nbytes = np.zeros(nt, dtype=np.int32)
    offs = np.zeros(nt, dtype=np.int64)
    zs = np.zeros(nt, dtype=np.float64) if is_float else None
    zz = np.zeros(nt, dtype=np.float64) if is_float else None
    used = C.c_longlong(0)
    rc = load_library().coreg_encode_tiled_host(a.ctypes.data, dt, ny, nx, tx, ty, int(bytepix), int(blocksize), int(quantize),
                                                int(dither0), float(scale), heap.ctypes.data, cap, nbytes.ctypes.data,
                                                offs.ctypes.data, zs.ctypes.data if is_float else None,
                                                zz.ctypes.data if is_float else None, C.byref(used))
    if rc != COREG_OK:
        raise CoregError(rc, "coreg_encode_tiled_host: bad arguments, or a tile's range does not fit 32-bit integers at "
                             "this scale")
    return heap[:used.value].copy(), nbytes, offs, zs, zz


def _is_raw(img):
    """utils.fits_io.RawImage (duck-typed: the binding does not import the FITS reader)."""
    return hasattr(img, "bitpix") and hasattr(img, "ptr")


def _fits_pixels(raw) -> FitsPixels:
    if len(raw.shape) != 2:
        raise ValueError("image must be 2-D")
    return FitsPixels(raw.ptr, int(raw.bitpix), 0, float(raw.bscale), float(raw.bzero))


# every symbol include/coreg_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
_WP = C.POINTER(Wcs2d)
SYMBOLS = [
    ("coreg_version", C.c_char_p, []),
    ("coreg_create", C.c_int, [C.POINTER(_P), C.c_int]),
    ("coreg_destroy", None, [_P]),
    ("coreg_last_error", C.c_char_p, [_P]),
    ("coreg_set_stream", C.c_int, [_P, _P]),
    ("coreg_synchronize", C.c_int, [_P]),
    ("coreg_set_small", C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    ("coreg_set_small_f32", C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    ("coreg_set_small_fits", C.c_int, [_P, C.POINTER(FitsPixels), C.c_int32, C.c_int32]),
    ("coreg_prepare_reference_carrington_fits", C.c_int,
     [_P, C.POINTER(FitsPixels), C.c_int32, C.c_int32, _WP, C.POINTER(CarrGrid), C.c_double, C.c_int]),
    ("coreg_prepare_reference_helioprojective_fits", C.c_int,
     [_P, C.POINTER(FitsPixels), C.c_int32, C.c_int32, _WP, _WP, C.c_int]),
    ("coreg_set_small_tiled", C.c_int, [_P, C.POINTER(FitsTiled)]),
    ("coreg_prepare_reference_carrington_tiled", C.c_int,
     [_P, C.POINTER(FitsTiled), _WP, C.POINTER(CarrGrid), C.c_double, C.c_int]),
    ("coreg_prepare_reference_helioprojective_tiled", C.c_int, [_P, C.POINTER(FitsTiled), _WP, _WP, C.c_int]),
    ("coreg_decode_tiled_host", C.c_int, [C.POINTER(FitsTiled), _P, C.c_int, _P]),
    ("coreg_encode_tiled_host", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_double, _P, C.c_longlong, _P, _P, _P, _P, C.POINTER(C.c_longlong)]),
    ("coreg_threshold_small", C.c_int, [_P, C.c_int, C.c_double, C.c_int, C.c_double, C.POINTER(C.c_longlong)]),
    ("coreg_set_reference_on_grid", C.c_int, [_P, _P, C.c_int, C.c_int32, C.c_int32]),
    ("coreg_prepare_reference_carrington", C.c_int,
     [_P, _P, C.c_int32, C.c_int32, _WP, C.POINTER(CarrGrid), C.c_double, C.c_int]),
    ("coreg_prepare_reference_helioprojective", C.c_int, [_P, _P, C.c_int32, C.c_int32, _WP, _WP, C.c_int]),
    ("coreg_prepare_reference_carrington_f32", C.c_int,
     [_P, _P, C.c_int32, C.c_int32, _WP, C.POINTER(CarrGrid), C.c_double, C.c_int]),
    ("coreg_prepare_reference_helioprojective_f32", C.c_int, [_P, _P, C.c_int32, C.c_int32, _WP, _WP, C.c_int]),
    ("coreg_set_small_from_device", C.c_int, [_P, _P, C.c_int, C.c_int32, C.c_int32]),
    ("coreg_prepare_reference_carrington_from_device", C.c_int,
     [_P, _P, C.c_int, C.c_int32, C.c_int32, _WP, C.POINTER(CarrGrid), C.c_double, C.c_int]),
    ("coreg_prepare_reference_helioprojective_from_device", C.c_int,
     [_P, _P, C.c_int, C.c_int32, C.c_int32, _WP, _WP, C.c_int]),
    ("coreg_get_reference_on_grid", C.c_int, [_P, _P, C.c_int]),
    ("coreg_resample_carrington", C.c_int, [_P, _WP, C.POINTER(CarrGrid), C.c_double, C.c_int, _P]),
    ("coreg_resample_helioprojective", C.c_int, [_P, _WP, _WP, C.c_int, _P]),
    ("coreg_resample_helioprojective_f64", C.c_int, [_P, _WP, _WP, C.c_int, _P]),
    ("coreg_sweep_carrington", C.c_int,
     [_P, _WP, C.POINTER(CarrGrid), C.c_double, C.POINTER(Lags), C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, _P,
      C.c_int]),
    ("coreg_sweep_helioprojective", C.c_int,
     [_P, _WP, _WP, C.POINTER(Lags), C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, _P, C.c_int]),
    ("coreg_sums_size", C.c_int, [_P, C.POINTER(C.c_int64)]),
    ("coreg_copy_sums", C.c_int, [_P, _P, C.c_int]),
    ("coreg_finalize_sums", C.c_int, [_P, _P, C.c_int, _P, C.c_int]),
    ("coreg_get_pivots", C.c_int, [_P, C.POINTER(C.c_double)]),
    ("coreg_set_pivots", C.c_int, [_P, C.POINTER(C.c_double)]),
    ("coreg_last_stats", C.c_int, [_P, C.POINTER(Stats)]),
    ("coreg_last_visit_counts", C.c_int, [_P, C.POINTER(C.c_int64)]),
    ("coreg_last_tap_fix", C.c_int, [_P, C.POINTER(C.c_int64)]),
    ("coreg_set_option", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("coreg_shift_header", C.c_int,
     [_WP, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, _WP]),
    ("coreg_homography", C.c_int, [_WP, _WP, C.POINTER(C.c_double)]),
    ("coreg_lag_homography", C.c_int,
     [_WP, _WP, C.POINTER(Lags), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_double)]),
    ("coreg_carrington_origin", C.c_int, [_WP, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("coreg_car_map", C.c_int, [_WP, _WP, C.c_int64, _P, _P, _P, _P]),
    ("coreg_wcslib_pixel_to_pixel", C.c_int, [_WP, _WP, C.c_int64, _P, _P, _P, _P, _P, _P]),
    ("coreg_car_tile_margin", C.c_int, [_WP, _WP, C.c_int32, C.c_double, C.POINTER(C.c_double)]),
    ("coreg_nansum_planes_be", C.c_int, [_P, C.c_int32, C.c_int64, _P, C.c_int32, _P]),
    ("coreg_fit_gaussian2d", C.c_int,
     [C.c_int32, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32, _P,
      C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    # all GPUs of the node from one process
    ("coreg_device_count", C.c_int, []),
    ("coreg_physical_device_count", C.c_int, []),
    ("coreg_multi_create", C.c_int, [C.POINTER(_P), C.c_int, C.POINTER(C.c_int)]),
    ("coreg_multi_destroy", None, [_P]),
    ("coreg_multi_size", C.c_int, [_P]),
    ("coreg_multi_handle", _P, [_P, C.c_int]),
    ("coreg_multi_last_error", C.c_char_p, [_P]),
    ("coreg_multi_collective", C.c_char_p, [_P]),
    ("coreg_multi_rccl_status", C.c_char_p, [_P]),
    ("coreg_multi_last_mode", C.c_int, [_P]),
    ("coreg_multi_set_option", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("coreg_multi_set_small", C.c_int, [_P, _P, C.c_int, C.c_int32, C.c_int32]),
    ("coreg_multi_set_small_fits", C.c_int, [_P, C.POINTER(FitsPixels), C.c_int32, C.c_int32]),
    ("coreg_multi_prepare_reference_carrington_fits", C.c_int,
     [_P, C.POINTER(FitsPixels), C.c_int32, C.c_int32, _WP, C.POINTER(CarrGrid), C.c_double, C.c_int]),
    ("coreg_multi_prepare_reference_helioprojective_fits", C.c_int,
     [_P, C.POINTER(FitsPixels), C.c_int32, C.c_int32, _WP, _WP, C.c_int]),
    ("coreg_multi_set_small_tiled", C.c_int, [_P, C.POINTER(FitsTiled)]),
    ("coreg_multi_prepare_reference_carrington_tiled", C.c_int,
     [_P, C.POINTER(FitsTiled), _WP, C.POINTER(CarrGrid), C.c_double, C.c_int]),
    ("coreg_multi_prepare_reference_helioprojective_tiled", C.c_int, [_P, C.POINTER(FitsTiled), _WP, _WP, C.c_int]),
    ("coreg_multi_threshold_small", C.c_int, [_P, C.c_int, C.c_double, C.c_int, C.c_double, C.POINTER(C.c_longlong)]),
    ("coreg_multi_set_reference_on_grid", C.c_int, [_P, _P, C.c_int, C.c_int32, C.c_int32]),
    ("coreg_multi_prepare_reference_carrington", C.c_int,
     [_P, _P, C.c_int, C.c_int32, C.c_int32, _WP, C.POINTER(CarrGrid), C.c_double, C.c_int]),
    ("coreg_multi_prepare_reference_helioprojective", C.c_int, [_P, _P, C.c_int, C.c_int32, C.c_int32, _WP, _WP, C.c_int]),
    ("coreg_multi_sweep_carrington", C.c_int,
     [_P, _WP, C.POINTER(CarrGrid), C.c_double, C.POINTER(Lags), C.c_int, C.c_int, C.c_int, _P]),
    ("coreg_multi_sweep_helioprojective", C.c_int, [_P, _WP, _WP, C.POINTER(Lags), C.c_int, C.c_int, C.c_int, _P]),
    ("coreg_multi_last_stats", C.c_int, [_P, C.c_int, C.POINTER(Stats)]),
    ("coreg_multi_plan", C.c_int,
     [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
      C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
]

_lib = None


def _preload_torch_hip_runtime():
    """PyTorch's ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, same SONAME as the system one).  A
    process that loads the SYSTEM runtime first (through this library) and imports torch afterwards ends up with a torch
    that reports "No HIP GPUs are available"; the other order works, both then share torch's copy.  So when torch is
    installed but not imported yet, its runtime is loaded first -- without importing torch (that costs seconds).
    COREG_NO_TORCH_PRELOAD=1 switches this off."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("COREG_NO_TORCH_PRELOAD", "0") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    cand = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            return  # the system runtime will serve; torch, if imported later, may not see the GPU
        # ... and the RCCL that goes with that runtime is the one the in-library multi-GPU driver should dlopen
        rccl = os.path.join(libdir, "librccl.so")
        if os.path.exists(rccl):
            os.environ.setdefault("COREG_RCCL_LIB", rccl)


def load_library():
    """dlopen libcoreg_hip.so and declare every prototype.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    _preload_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` at the repo root "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


_UNIT_TO_DEG = {"deg": 1.0, "arcsec": 1.0 / 3600.0, "arcmin": 1.0 / 60.0, "rad": 180.0 / np.pi, "mas": 1.0 / 3.6e6}


def unit_to_deg(unit) -> float:
    u = str(unit).strip()
    if u not in _UNIT_TO_DEG:
        raise ValueError(f"unsupported CUNIT {unit!r}")
    return _UNIT_TO_DEG[u]


def wcs_from_header(hdr, carrington=False) -> Wcs2d:
    """Fill the C struct from a header mapping (after alignment.py:580-611: PCi_j and CROTA present)."""
    w = Wcs2d()
    if "ZNAXIS1" in hdr:
        w.naxis1, w.naxis2 = int(hdr["ZNAXIS1"]), int(hdr["ZNAXIS2"])
    else:
        w.naxis1, w.naxis2 = int(hdr.get("NAXIS1", 0)), int(hdr.get("NAXIS2", 0))
    w.crpix1, w.crpix2 = float(hdr["CRPIX1"]), float(hdr["CRPIX2"])
    w.crval1, w.crval2 = float(hdr["CRVAL1"]), float(hdr["CRVAL2"])
    w.cdelt1, w.cdelt2 = float(hdr["CDELT1"]), float(hdr["CDELT2"])
    w.pc1_1, w.pc1_2 = float(hdr.get("PC1_1", 1.0)), float(hdr.get("PC1_2", 0.0))
    w.pc2_1, w.pc2_2 = float(hdr.get("PC2_1", 0.0)), float(hdr.get("PC2_2", 1.0))
    w.crota = float(hdr["CROTA"] if "CROTA" in hdr else hdr.get("CROTA2", 0.0))
    u1 = hdr.get("CUNIT1", "deg")
    if str(u1).strip() != str(hdr.get("CUNIT2", u1)).strip():
        raise ValueError("CUNIT1 and CUNIT2 must be equal")  # alignment.py:839-840
    w.unit_to_deg = unit_to_deg(u1)
    # projection: gnomonic (HPLN-TAN / HPLT-TAN) unless the axes say plate carree (CRLN-CAR / CRLT-CAR maps,
    # alignment.py:365-366); a CAR header without LONPOLE / LATPOLE gets the FITS defaults inside the library
    if str(hdr.get("CTYPE1", "")).strip().upper().endswith("-CAR"):
        w.proj = PROJ_CAR
        w.lonpole = float(hdr["LONPOLE"]) if "LONPOLE" in hdr else float("nan")
    else:
        w.proj = PROJ_TAN
        w.lonpole = float(hdr.get("LONPOLE", 180.0))
    w.latpole = float(hdr["LATPOLE"]) if "LATPOLE" in hdr else float("nan")
    if carrington:
        w.dsun_obs = float(hdr["DSUN_OBS"])
        w.crln_obs = float(hdr["CRLN_OBS"])
        w.crlt_obs = float(hdr["CRLT_OBS"])
    return w


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class LagSet:
    """Owns contiguous float64 copies of the five lag axes and the C struct pointing at them."""

    def __init__(self, crval1, crval2, cdelt1, cdelt2, crota):
        def arr(v):
            return np.ascontiguousarray(np.atleast_1d(np.asarray([0.0] if v is None else v, dtype=np.float64)))

        self.arrays = [arr(crval1), arr(crval2), arr(cdelt1), arr(cdelt2), arr(crota)]
        a = self.arrays
        self.c = Lags(_dptr(a[0]), len(a[0]), _dptr(a[1]), len(a[1]), _dptr(a[2]), len(a[2]), _dptr(a[3]), len(a[3]),
                      _dptr(a[4]), len(a[4]))
        self.shape = tuple(len(x) for x in a)
        self.size = int(np.prod(self.shape))


class Grid:
    """Carrington grid (utils/rectify.py:875-878) as the C struct; `numpy_lat_trig=True` passes NumPy's own
    float32 cos/sin(radians(lat)) tables so that the latitude trig equals the reference's bit for bit."""

    def __init__(self, lonlims, latlims, shape, numpy_lat_trig=True):
        self.n_lon, self.n_lat = int(shape[0]), int(shape[1])
        self._cos = self._sin = None
        cp = sp = None
        if numpy_lat_trig:
            lat = np.radians(np.linspace(latlims[0], latlims[1], self.n_lat, dtype=np.float32))
            self._cos = np.ascontiguousarray(np.cos(lat), dtype=np.float32)
            self._sin = np.ascontiguousarray(np.sin(lat), dtype=np.float32)
            cp = self._cos.ctypes.data_as(C.POINTER(C.c_float))
            sp = self._sin.ctypes.data_as(C.POINTER(C.c_float))
        self.c = CarrGrid(float(lonlims[0]), float(lonlims[1]), self.n_lon, float(latlims[0]), float(latlims[1]),
                          self.n_lat, cp, sp)


class CoregHandle:
    """One GPU context of libcoreg_hip (RAII over coreg_create / coreg_destroy)."""

    def __init__(self, device=-1):
        self._lib = load_library()
        self._h = _P()
        rc = self._lib.coreg_create(C.byref(self._h), int(device))
        if rc != COREG_OK:
            self._h = None
            raise CoregError(rc, "coreg_create failed (no HIP device visible?)")
        # caller-defined identity of the prepared reference now resident on the device (None = unknown): lets a session
        # that aligns many images to one reference skip decoding and re-preparing it
        self.reference_tag = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.coreg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc):
        if rc != COREG_OK:
            raise CoregError(rc, self._lib.coreg_last_error(self._h).decode("utf-8", "replace"))

    # -- configuration
    def set_option(self, name, value):
        self._chk(self._lib.coreg_set_option(self._h, name.encode(), int(value)))

    def set_stream(self, stream_ptr):
        self._chk(self._lib.coreg_set_stream(self._h, _P(stream_ptr)))

    def synchronize(self):
        self._chk(self._lib.coreg_synchronize(self._h))

    # -- images
    def set_small(self, img):
        """Image to align.  A RawImage (utils/fits_io.py) goes up as the file stores it and is decoded on the GPU;
        float32 arrays (FITS BITPIX=-32 pixels) go up as they are; anything else as float64."""
        if _is_tiled(img):  # a tile-compressed image: the compressed bytes go up, the GPU decodes them
            t, keep = _fits_tiled(img)
            self._chk(self._lib.coreg_set_small_tiled(self._h, C.byref(t)))
            return
        # (option "async_upload": the library's upload thread reads the pixels after this call has returned -- the object
        # that owns them is kept until the next upload replaces it; the caller must not modify them meanwhile)
        # (the PREVIOUS buffer stays referenced until the library call has returned -- that call joins the
        # upload job that may still be reading it -- and only then is it let go)
        previous = getattr(self, "_small_keepalive", None)
        if _is_raw(img):
            px = _fits_pixels(img)
            self._chk(self._lib.coreg_set_small_fits(self._h, C.byref(px), img.shape[0], img.shape[1]))
            self._small_keepalive = (img, px)
            del previous
            return
        img = np.asarray(img)
        if img.ndim != 2:
            raise ValueError("small image must be 2-D")
        if img.dtype == np.float32:
            img = np.ascontiguousarray(img)
            self._chk(self._lib.coreg_set_small_f32(self._h, img.ctypes.data, img.shape[0], img.shape[1]))
            self._small_keepalive = img
            del previous
        else:
            img = np.ascontiguousarray(img, dtype=np.float64)
            self._chk(self._lib.coreg_set_small(self._h, img.ctypes.data, img.shape[0], img.shape[1]))

    def drop_small_keepalive(self):
        """The caller knows that the image has been read (a sweep has returned): let go of the pixel buffer held for the
        upload thread, so that it is freed where the caller frees it and not inside the next upload."""
        self._small_keepalive = None

    def threshold_small(self, vmin=None, vmax=None) -> int:
        """|v| < vmin or |v| > vmax -> NaN on the resident image to align (alignment.py:876-887); returns the number of
        finite pixels left."""
        n = C.c_longlong(0)
        self._chk(self._lib.coreg_threshold_small(self._h, int(vmin is not None), float(vmin or 0.0),
                                                  int(vmax is not None), float(vmax or 0.0), C.byref(n)))
        return int(n.value)

    def set_reference_on_grid(self, ref):
        ref = np.ascontiguousarray(ref)
        if ref.ndim != 2 or ref.dtype not in (np.float32, np.float64):
            raise ValueError("reference on grid must be a 2-D float32/float64 array")
        dt = COREG_F32 if ref.dtype == np.float32 else COREG_F64
        self.reference_tag = None
        self._chk(self._lib.coreg_set_reference_on_grid(self._h, ref.ctypes.data, dt, ref.shape[0], ref.shape[1]))

    @staticmethod
    def _reference_pixels(large):
        """float32 arrays (FITS BITPIX=-32 pixels) go up as they are, anything else as float64."""
        large = np.asarray(large)
        if large.ndim != 2:
            raise ValueError("reference image must be 2-D")
        if large.dtype == np.float32:
            return np.ascontiguousarray(large), True
        return np.ascontiguousarray(large, dtype=np.float64), False

    def prepare_reference_carrington(self, large, hdr_large, grid: Grid, solar_r, order=2):
        w = wcs_from_header(hdr_large, carrington=True)
        self.reference_tag = None
        if _is_tiled(large):
            t, keep = _fits_tiled(large)
            self._chk(self._lib.coreg_prepare_reference_carrington_tiled(self._h, C.byref(t), C.byref(w), C.byref(grid.c),
                                                                         float(solar_r), int(order)))
            return
        if _is_raw(large):
            px = _fits_pixels(large)
            self._chk(self._lib.coreg_prepare_reference_carrington_fits(
                self._h, C.byref(px), large.shape[0], large.shape[1], C.byref(w), C.byref(grid.c), float(solar_r),
                int(order)))
            return
        large, f32 = self._reference_pixels(large)
        fn = self._lib.coreg_prepare_reference_carrington_f32 if f32 else self._lib.coreg_prepare_reference_carrington
        self._chk(fn(self._h, large.ctypes.data, large.shape[0], large.shape[1], C.byref(w), C.byref(grid.c),
                     float(solar_r), int(order)))

    def prepare_reference_helioprojective(self, large, hdr_large, hdr_small, order=2):
        wl, ws = wcs_from_header(hdr_large), wcs_from_header(hdr_small)
        self.reference_tag = None
        if _is_tiled(large):
            t, keep = _fits_tiled(large)
            self._chk(self._lib.coreg_prepare_reference_helioprojective_tiled(self._h, C.byref(t), C.byref(wl), C.byref(ws),
                                                                              int(order)))
            return
        if _is_raw(large):
            px = _fits_pixels(large)
            self._chk(self._lib.coreg_prepare_reference_helioprojective_fits(
                self._h, C.byref(px), large.shape[0], large.shape[1], C.byref(wl), C.byref(ws), int(order)))
            return
        large, f32 = self._reference_pixels(large)
        fn = self._lib.coreg_prepare_reference_helioprojective_f32 if f32 else \
            self._lib.coreg_prepare_reference_helioprojective
        self._chk(fn(self._h, large.ctypes.data, large.shape[0], large.shape[1], C.byref(wl), C.byref(ws), int(order)))

    # -- the same with the pixels already on this GPU (device pointer + shape + numpy dtype float32 / float64)
    @staticmethod
    def _dev_dtype(dtype):
        dt = np.dtype(dtype)
        if dt == np.float32:
            return COREG_F32
        if dt == np.float64:
            return COREG_F64
        raise ValueError("device images must be float32 or float64")

    def set_small_from_device(self, dev_ptr, shape, dtype=np.float32):
        self._chk(self._lib.coreg_set_small_from_device(self._h, _P(int(dev_ptr)), self._dev_dtype(dtype),
                                                        int(shape[0]), int(shape[1])))

    def prepare_reference_carrington_from_device(self, dev_ptr, shape, dtype, hdr_large, grid: Grid, solar_r, order=2):
        w = wcs_from_header(hdr_large, carrington=True)
        self.reference_tag = None
        self._chk(self._lib.coreg_prepare_reference_carrington_from_device(
            self._h, _P(int(dev_ptr)), self._dev_dtype(dtype), int(shape[0]), int(shape[1]), C.byref(w),
            C.byref(grid.c), float(solar_r), int(order)))

    def prepare_reference_helioprojective_from_device(self, dev_ptr, shape, dtype, hdr_large, hdr_small, order=2):
        wl, ws = wcs_from_header(hdr_large), wcs_from_header(hdr_small)
        self.reference_tag = None
        self._chk(self._lib.coreg_prepare_reference_helioprojective_from_device(
            self._h, _P(int(dev_ptr)), self._dev_dtype(dtype), int(shape[0]), int(shape[1]), C.byref(wl), C.byref(ws),
            int(order)))

    def get_reference_on_grid(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        dt = COREG_F32 if out.dtype == np.float32 else COREG_F64
        self._chk(self._lib.coreg_get_reference_on_grid(self._h, out.ctypes.data, dt))
        return out

    # -- single resamples
    def resample_carrington(self, hdr, grid: Grid, solar_r, order=2):
        w = wcs_from_header(hdr, carrington=True)
        out = np.empty((grid.n_lat, grid.n_lon), dtype=np.float64)
        self._chk(self._lib.coreg_resample_carrington(self._h, C.byref(w), C.byref(grid.c), float(solar_r), int(order),
                                                      out.ctypes.data))
        return out

    def resample_helioprojective(self, hdr_target, hdr, order=2, dtype=np.float32):
        wt, w = wcs_from_header(hdr_target), wcs_from_header(hdr)
        out = np.empty((wt.naxis2, wt.naxis1), dtype=dtype)
        fn = self._lib.coreg_resample_helioprojective if out.dtype == np.float32 else \
            self._lib.coreg_resample_helioprojective_f64
        self._chk(fn(self._h, C.byref(wt), C.byref(w), int(order), out.ctypes.data))
        return out

    # -- sweeps
    def sweep_carrington(self, hdr_small, grid: Grid, solar_r, lags: LagSet, order=2, method=METHOD_CORRELATION,
                         cdelt_semantics=CDELT_INTENDED, lag_begin=0, lag_end=None, out_dev_ptr=None):
        lag_end = lags.size if lag_end is None else int(lag_end)
        w = wcs_from_header(hdr_small, carrington=True)
        if out_dev_ptr is None:
            out = np.empty(lag_end - lag_begin, dtype=np.float64)
            ptr, on_dev = out.ctypes.data, 0
        else:
            out, ptr, on_dev = None, int(out_dev_ptr), 1
        self._chk(self._lib.coreg_sweep_carrington(self._h, C.byref(w), C.byref(grid.c), float(solar_r),
                                                   C.byref(lags.c), int(order), int(method), int(cdelt_semantics),
                                                   int(lag_begin), lag_end, _P(ptr), on_dev))
        return out

    def sweep_helioprojective(self, hdr_target, hdr_small, lags: LagSet, order=2, method=METHOD_CORRELATION,
                              cdelt_semantics=CDELT_INTENDED, lag_begin=0, lag_end=None, out_dev_ptr=None):
        lag_end = lags.size if lag_end is None else int(lag_end)
        wt, w = wcs_from_header(hdr_target), wcs_from_header(hdr_small)
        if out_dev_ptr is None:
            out = np.empty(lag_end - lag_begin, dtype=np.float64)
            ptr, on_dev = out.ctypes.data, 0
        else:
            out, ptr, on_dev = None, int(out_dev_ptr), 1
        self._chk(self._lib.coreg_sweep_helioprojective(self._h, C.byref(wt), C.byref(w), C.byref(lags.c), int(order),
                                                        int(method), int(cdelt_semantics), int(lag_begin), lag_end,
                                                        _P(ptr), on_dev))
        return out

    # -- multi-GPU point sharding (see include/coreg_hip.h, coreg_finalize_sums)
    def set_point_shard(self, rank, world):
        self.set_option("shard_world", int(world))
        self.set_option("shard_rank", int(rank))

    def sums_size(self) -> int:
        n = C.c_int64(0)
        self._chk(self._lib.coreg_sums_size(self._h, C.byref(n)))
        return int(n.value)

    def copy_sums(self, dst_ptr=None):
        """The six sums per lag slot of the pending point-sharded sweep: into a device buffer (`dst_ptr`), or returned
        as a host array."""
        if dst_ptr is not None:
            self._chk(self._lib.coreg_copy_sums(self._h, _P(int(dst_ptr)), 1))
            return None
        out = np.empty(self.sums_size(), dtype=np.float64)
        self._chk(self._lib.coreg_copy_sums(self._h, out.ctypes.data, 0))
        return out

    def finalize_sums(self, sums, n_out, out_dev_ptr=None):
        """Reduced sums (host array, or an int device pointer) -> coefficients of the pending sweep's lag slice."""
        if isinstance(sums, np.ndarray):
            sums = np.ascontiguousarray(sums, dtype=np.float64)
            sptr, s_dev = sums.ctypes.data, 0
        else:
            sptr, s_dev = int(sums), 1
        if out_dev_ptr is None:
            out = np.empty(int(n_out), dtype=np.float64)
            self._chk(self._lib.coreg_finalize_sums(self._h, _P(sptr), s_dev, out.ctypes.data, 0))
            return out
        self._chk(self._lib.coreg_finalize_sums(self._h, _P(sptr), s_dev, _P(int(out_dev_ptr)), 1))
        return None

    def get_pivots(self):
        """(mean of the finite reference values on the grid, mean of the finite pixels of the image to align)."""
        p = (C.c_double * 2)()
        self._chk(self._lib.coreg_get_pivots(self._h, p))
        return float(p[0]), float(p[1])

    def set_pivots(self, pivot_ref, pivot_small):
        p = (C.c_double * 2)(float(pivot_ref), float(pivot_small))
        self._chk(self._lib.coreg_set_pivots(self._h, p))

    def last_stats(self) -> dict:
        s = Stats()
        self._chk(self._lib.coreg_last_stats(self._h, C.byref(s)))
        return {f: getattr(s, f) for f, _ in Stats._fields_}

    def last_visit_counts(self) -> dict:
        """(tile, lag batch) visits of the sweep kernel's last launch by kind (diagnostics; waits for the stream)."""
        c = (C.c_int64 * 6)()
        self._chk(self._lib.coreg_last_visit_counts(self._h, c))
        return {"visits": c[0], "lds": c[1], "interior": c[2], "all_finite": c[3], "refined_lag_points": c[4],
                "flagged_not_refined": c[5]}


    def last_tap_fix(self) -> dict:
        """Odd spline orders, helioprojective frame: samples of the last sweep re-evaluated with wcslib's arithmetic."""
        c = (C.c_int64 * 3)()
        self._chk(self._lib.coreg_last_tap_fix(self._h, c))
        return {"samples": c[0], "lag_points": c[1], "overflow": bool(c[2])}


class _HandleView(CoregHandle):
    """A CoregHandle that does not own its context (device k of a MultiHandle): single-device utilities and options."""

    def __init__(self, lib, ptr):  # noqa: super().__init__ would create a context
        self._lib = lib
        self._h = _P(ptr)
        self.reference_tag = None

    def close(self):
        self._h = None


MULTI_MODES = {0: "none", 1: "blocks", 2: "slices", 3: "points", 4: "combos"}


def device_count() -> int:
    """GPUs the library sees (COREG_VIRTUAL_DEVICES overrides it); 0 when there is none.  Does not touch torch."""
    return max(0, int(load_library().coreg_device_count()))


def physical_device_count() -> int:
    """GPUs actually present (coreg_device_count() minus the COREG_VIRTUAL_DEVICES mapping)."""
    return max(0, int(load_library().coreg_physical_device_count()))


def multi_plan(n_crval1, n_crval2, n_inner, world, per_combo_launch=True):
    """(mode, g_combo, g1, g2) the in-library multi-GPU driver gives a lag set (host-only helper; mirrors
    parallel.lag_plan)."""
    mode, gc, g1, g2 = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    rc = load_library().coreg_multi_plan(int(n_crval1), int(n_crval2), int(n_inner), int(world), int(per_combo_launch),
                                         C.byref(mode), C.byref(gc), C.byref(g1), C.byref(g2))
    if rc != COREG_OK:
        raise CoregError(rc, "coreg_multi_plan: bad arguments")
    return MULTI_MODES[mode.value], gc.value, g1.value, g2.value


class MultiHandle:
    """Every GPU of the node from this one process (include/coreg_hip.h, coreg_multi): one host thread + library context
    per device inside the library, full image replicas, lag-plane blocks, ONE RCCL all-gather of the per-lag
    coefficients.  Same calling surface as CoregHandle for what `hdrshift.Alignment` needs; sweeps always return the
    whole map (lag_begin / lag_end must span it)."""

    def __init__(self, n_devices=0, device_ids=None):
        self._lib = load_library()
        self._m = _P()
        ids = None
        if device_ids is not None:
            ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
            n_devices = len(device_ids)
        rc = self._lib.coreg_multi_create(C.byref(self._m), int(n_devices), ids)
        if rc != COREG_OK:
            self._m = None
            raise CoregError(rc, "coreg_multi_create failed (no HIP device visible?)")
        self.reference_tag = None
        self.size = int(self._lib.coreg_multi_size(self._m))
        self.primary = _HandleView(self._lib, self._lib.coreg_multi_handle(self._m, 0))

    def close(self):
        if getattr(self, "_m", None):
            self.primary.close()
            self._lib.coreg_multi_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc):
        if rc != COREG_OK:
            raise CoregError(rc, self._lib.coreg_multi_last_error(self._m).decode("utf-8", "replace"))

    @property
    def collective(self) -> str:
        return self._lib.coreg_multi_collective(self._m).decode()

    @property
    def rccl_status(self) -> str:
        return self._lib.coreg_multi_rccl_status(self._m).decode()

    @property
    def last_mode(self) -> str:
        return MULTI_MODES[int(self._lib.coreg_multi_last_mode(self._m))]

    def handle(self, k) -> CoregHandle:
        ptr = self._lib.coreg_multi_handle(self._m, int(k))
        if not ptr:
            raise IndexError(k)
        return _HandleView(self._lib, ptr)

    def set_option(self, name, value):
        self._chk(self._lib.coreg_multi_set_option(self._m, name.encode(), int(value)))

    def synchronize(self):
        for k in range(self.size):
            self.handle(k).synchronize()

    @staticmethod
    def _pixels(img):
        img = np.asarray(img)
        if img.ndim != 2:
            raise ValueError("image must be 2-D")
        if img.dtype == np.float32:
            return np.ascontiguousarray(img), COREG_F32
        return np.ascontiguousarray(img, dtype=np.float64), COREG_F64

    def set_small(self, img):
        if _is_tiled(img):
            t, keep = _fits_tiled(img)
            self._chk(self._lib.coreg_multi_set_small_tiled(self._m, C.byref(t)))
            return
        if _is_raw(img):
            px = _fits_pixels(img)
            self._chk(self._lib.coreg_multi_set_small_fits(self._m, C.byref(px), img.shape[0], img.shape[1]))
            return
        img, dt = self._pixels(img)
        self._chk(self._lib.coreg_multi_set_small(self._m, img.ctypes.data, dt, img.shape[0], img.shape[1]))

    def threshold_small(self, vmin=None, vmax=None) -> int:
        n = C.c_longlong(0)
        self._chk(self._lib.coreg_multi_threshold_small(self._m, int(vmin is not None), float(vmin or 0.0),
                                                        int(vmax is not None), float(vmax or 0.0), C.byref(n)))
        return int(n.value)

    def set_reference_on_grid(self, ref):
        ref = np.ascontiguousarray(ref)
        if ref.ndim != 2 or ref.dtype not in (np.float32, np.float64):
            raise ValueError("reference on grid must be a 2-D float32/float64 array")
        self.reference_tag = None
        self._chk(self._lib.coreg_multi_set_reference_on_grid(
            self._m, ref.ctypes.data, COREG_F32 if ref.dtype == np.float32 else COREG_F64, ref.shape[0], ref.shape[1]))

    def prepare_reference_carrington(self, large, hdr_large, grid: Grid, solar_r, order=2):
        w = wcs_from_header(hdr_large, carrington=True)
        self.reference_tag = None
        if _is_tiled(large):
            t, keep = _fits_tiled(large)
            self._chk(self._lib.coreg_multi_prepare_reference_carrington_tiled(self._m, C.byref(t), C.byref(w),
                                                                               C.byref(grid.c), float(solar_r), int(order)))
            return
        if _is_raw(large):
            px = _fits_pixels(large)
            self._chk(self._lib.coreg_multi_prepare_reference_carrington_fits(
                self._m, C.byref(px), large.shape[0], large.shape[1], C.byref(w), C.byref(grid.c), float(solar_r),
                int(order)))
            return
        large, dt = self._pixels(large)
        self._chk(self._lib.coreg_multi_prepare_reference_carrington(
            self._m, large.ctypes.data, dt, large.shape[0], large.shape[1], C.byref(w), C.byref(grid.c), float(solar_r),
            int(order)))

    def prepare_reference_helioprojective(self, large, hdr_large, hdr_small, order=2):
        wl, ws = wcs_from_header(hdr_large), wcs_from_header(hdr_small)
        self.reference_tag = None
        if _is_tiled(large):
            t, keep = _fits_tiled(large)
            self._chk(self._lib.coreg_multi_prepare_reference_helioprojective_tiled(self._m, C.byref(t), C.byref(wl),
                                                                                    C.byref(ws), int(order)))
            return
        if _is_raw(large):
            px = _fits_pixels(large)
            self._chk(self._lib.coreg_multi_prepare_reference_helioprojective_fits(
                self._m, C.byref(px), large.shape[0], large.shape[1], C.byref(wl), C.byref(ws), int(order)))
            return
        large, dt = self._pixels(large)
        self._chk(self._lib.coreg_multi_prepare_reference_helioprojective(
            self._m, large.ctypes.data, dt, large.shape[0], large.shape[1], C.byref(wl), C.byref(ws), int(order)))

    def _whole(self, lags, lag_begin, lag_end):
        if int(lag_begin) != 0 or (lag_end is not None and int(lag_end) != lags.size):
            raise ValueError("a multi-GPU sweep covers the whole lag set (the library cuts it itself)")

    def sweep_carrington(self, hdr_small, grid: Grid, solar_r, lags: LagSet, order=2, method=METHOD_CORRELATION,
                         cdelt_semantics=CDELT_INTENDED, lag_begin=0, lag_end=None):
        self._whole(lags, lag_begin, lag_end)
        w = wcs_from_header(hdr_small, carrington=True)
        out = np.empty(lags.size, dtype=np.float64)
        self._chk(self._lib.coreg_multi_sweep_carrington(self._m, C.byref(w), C.byref(grid.c), float(solar_r),
                                                         C.byref(lags.c), int(order), int(method), int(cdelt_semantics),
                                                         out.ctypes.data))
        return out

    def sweep_helioprojective(self, hdr_target, hdr_small, lags: LagSet, order=2, method=METHOD_CORRELATION,
                              cdelt_semantics=CDELT_INTENDED, lag_begin=0, lag_end=None):
        self._whole(lags, lag_begin, lag_end)
        wt, w = wcs_from_header(hdr_target), wcs_from_header(hdr_small)
        out = np.empty(lags.size, dtype=np.float64)
        self._chk(self._lib.coreg_multi_sweep_helioprojective(self._m, C.byref(wt), C.byref(w), C.byref(lags.c),
                                                              int(order), int(method), int(cdelt_semantics),
                                                              out.ctypes.data))
        return out

    def resample_helioprojective(self, hdr_target, hdr, order=2, dtype=np.float32):
        """Single resample of the resident image to align: device 0's copy (every device holds the same one)."""
        return self.primary.resample_helioprojective(hdr_target, hdr, order=order, dtype=dtype)

    def get_reference_on_grid(self, shape, dtype):
        return self.primary.get_reference_on_grid(shape, dtype)

    def last_stats(self, k=None):
        """Stats of device k's share of the last sweep; k = None: device 0's with the kernel times of all devices."""
        def one(i):
            s = Stats()
            self._chk(self._lib.coreg_multi_last_stats(self._m, int(i), C.byref(s)))
            return {f: getattr(s, f) for f, _ in Stats._fields_}
        if k is not None:
            return one(k)
        per = [one(i) for i in range(self.size)]
        out = dict(per[0])
        out["per_device_sweep_kernel_ms"] = [p["sweep_kernel_ms"] for p in per]
        out["n_lags"] = sum(p["n_lags"] for p in per) if self.last_mode != "points" else per[0]["n_lags"]
        out["n_devices"] = self.size
        out["collective"] = self.collective
        out["lag_sharding"] = self.last_mode
        return out


_SHARED = {}


def shared_handle(device=-1, slot=0) -> CoregHandle:
    """One long-lived handle per device for callers that run many sweeps back to back (jitter correction runs one sweep
    per image pair, hdrshift caller jitter_correction/jitter_correction.py:101-138): device buffers, pinned staging and
    the stream are allocated once and re-used.  Closed at interpreter exit.  `slot`: independent contexts on the same
    device (own stream and buffers), one per host thread that drives sweeps concurrently."""
    key = (int(device), int(slot))
    h = _SHARED.get(key)
    if h is None or getattr(h, "_h", None) is None:
        h = CoregHandle(device)
        _SHARED[key] = h
    return h


def shared_multi_handle() -> MultiHandle:
    """One long-lived MultiHandle over all visible GPUs for this process (created on first use)."""
    h = _SHARED.get("multi")
    if h is None or getattr(h, "_m", None) is None:
        h = MultiHandle(0)
        _SHARED["multi"] = h
    return h


def _close_shared():
    for h in list(_SHARED.values()):
        try:
            h.close()
        except Exception:
            pass
    _SHARED.clear()


import atexit  # noqa: E402

atexit.register(_close_shared)


# host-only helpers (no GPU): used by CPU tests of the header logic
def shift_header(hdr, d_crval1, d_crval2, d_cdelt1, d_cdelt2, d_crota, cdelt_semantics=CDELT_INTENDED):
    lib = load_library()
    w = wcs_from_header(hdr)
    out = Wcs2d()
    rc = lib.coreg_shift_header(C.byref(w), d_crval1, d_crval2, d_cdelt1, d_cdelt2, d_crota, cdelt_semantics,
                                C.byref(out))
    if rc < 0:
        raise CoregError(rc, "coreg_shift_header")
    return rc, out


def homography(hdr_from, hdr_to):
    lib = load_library()
    a, b = wcs_from_header(hdr_from), wcs_from_header(hdr_to)
    h = (C.c_double * 9)()
    rc = lib.coreg_homography(C.byref(a), C.byref(b), h)
    if rc != COREG_OK:
        raise CoregError(rc, "coreg_homography")
    return np.array(h[:]).reshape(3, 3)


def lag_homography(hdr_target, hdr_small, lags: "LagSet", idx, cdelt_semantics=CDELT_INTENDED):
    lib = load_library()
    a, b = wcs_from_header(hdr_target), wcs_from_header(hdr_small)
    ii = (C.c_int32 * 5)(*[int(i) for i in idx])
    h = (C.c_double * 9)()
    rc = lib.coreg_lag_homography(C.byref(a), C.byref(b), C.byref(lags.c), ii, cdelt_semantics, h)
    if rc < 0:
        raise CoregError(rc, "coreg_lag_homography")
    return rc, np.array(h[:]).reshape(3, 3)


def car_map(hdr_from, hdr_to, px, py):
    """Pixels of a CAR header -> pixels of another CAR header (host, no GPU).  None when a header has no valid pole."""
    lib = load_library()
    px = np.ascontiguousarray(px, dtype=np.float64).ravel()
    py = np.ascontiguousarray(py, dtype=np.float64).ravel()
    ox, oy = np.empty_like(px), np.empty_like(py)
    wf, wt = wcs_from_header(hdr_from), wcs_from_header(hdr_to)
    rc = lib.coreg_car_map(C.byref(wf), C.byref(wt), px.size, px.ctypes.data, py.ctypes.data, ox.ctypes.data,
                           oy.ctypes.data)
    if rc == 1:
        return None
    if rc != COREG_OK:
        raise CoregError(rc, "coreg_car_map: bad arguments")
    return ox, oy


def car_tile_margin(hdr_target, hdr_shifted, tile_w, tile_abs_lat_rad):
    """Pixels by which k_sweep widens a CAR tile's corner box (host, no GPU); None when a header has no valid pole."""
    lib = load_library()
    wf, wt = wcs_from_header(hdr_target), wcs_from_header(hdr_shifted)
    m = C.c_double()
    rc = lib.coreg_car_tile_margin(C.byref(wf), C.byref(wt), int(tile_w), float(tile_abs_lat_rad), C.byref(m))
    if rc == 1:
        return None
    if rc != COREG_OK:
        raise CoregError(rc, "coreg_car_tile_margin: bad arguments")
    return m.value


def wcslib_pixel_to_pixel(hdr_from, hdr_to, px, py):
    """pixel -> sky -> ang2pipi -> pixel through the library's restatement of wcslib (host, no GPU).
    Returns (x, y, lng_deg, lat_deg)."""
    lib = load_library()
    px = np.ascontiguousarray(px, dtype=np.float64).ravel()
    py = np.ascontiguousarray(py, dtype=np.float64).ravel()
    out = [np.empty_like(px) for _ in range(4)]
    wf, wt = wcs_from_header(hdr_from), wcs_from_header(hdr_to)
    rc = lib.coreg_wcslib_pixel_to_pixel(C.byref(wf), C.byref(wt), px.size, px.ctypes.data, py.ctypes.data,
                                         *[o.ctypes.data for o in out])
    if rc != COREG_OK:
        raise CoregError(rc, "coreg_wcslib_pixel_to_pixel")
    return tuple(out)


def nansum_planes_be(cube_be, plane_index):
    """np.nansum(float64(cube_be)[plane_index], axis=0) of a big-endian float32 / float64 array [n_planes, ...] (a view
    of a FITS data unit) by the library's threaded host routine; None when the array is not of that kind."""
    a = cube_be
    if a.dtype.kind != "f" or a.dtype.byteorder != ">" or a.dtype.itemsize not in (4, 8) or not a.flags.c_contiguous:
        return None
    idx = np.ascontiguousarray(plane_index, dtype=np.int64)
    if idx.size and (idx.min() < 0 or idx.max() >= a.shape[0]):
        raise IndexError("plane index out of range")
    n_pix = int(np.prod(a.shape[1:]))
    out = np.empty(a.shape[1:], dtype=np.float64)
    rc = load_library().coreg_nansum_planes_be(a.ctypes.data, -8 * a.dtype.itemsize, n_pix, idx.ctypes.data, idx.size,
                                               out.ctypes.data)
    if rc != COREG_OK:
        raise CoregError(rc, "coreg_nansum_planes_be: bad arguments")
    return out


def fit_gaussian2d(x, y, z, p0, lb, ub, jac="2-point", ftol=1e-8, xtol=1e-8, gtol=1e-8, max_nfev=0):
    """Bounded least-squares fit of the 2-D Gaussian of AlignmentResults.py:12-21 by the library's restatement of scipy's
    `curve_fit(..., bounds=...)` (host, no GPU).  Returns (popt, status, nfev): status as scipy's least_squares (0 =
    max_nfev reached, -1 = non-finite input / residuals)."""
    lib = load_library()
    arrs = [np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel()) for a in (x, y, z, p0, lb, ub)]
    if not (arrs[0].size == arrs[1].size == arrs[2].size) or any(a.size != 6 for a in arrs[3:]):
        raise ValueError("fit_gaussian2d: x, y, z must have one length; p0, lb, ub six entries")
    if jac not in ("2-point", "analytic"):
        raise ValueError("jac must be '2-point' or 'analytic'")
    popt = np.empty(6)
    nfev, status = C.c_int32(0), C.c_int32(0)
    rc = lib.coreg_fit_gaussian2d(arrs[0].size, *[a.ctypes.data for a in arrs], int(jac == "analytic"), float(ftol),
                                  float(xtol), float(gtol), int(max_nfev), popt.ctypes.data, C.byref(nfev),
                                  C.byref(status))
    if rc != COREG_OK:
        raise CoregError(rc, "coreg_fit_gaussian2d: bad arguments (sizes, bounds, or p0 outside the bounds)")
    return popt, int(status.value), int(nfev.value)


def carrington_origin(hdr):
    lib = load_library()
    w = wcs_from_header(hdr, carrington=True)
    x0, y0 = C.c_double(), C.c_double()
    rc = lib.coreg_carrington_origin(C.byref(w), C.byref(x0), C.byref(y0))
    if rc != COREG_OK:
        raise CoregError(rc, "coreg_carrington_origin")
    return x0.value, y0.value
