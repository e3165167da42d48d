/*
 * coreg_hip.h -- C ABI of libcoreg_hip.so: the MI355X (gfx950) implementation of the euispice_coreg
 * `hdrshift.Alignment` correlation sweep (one lag-point = resample the small-FOV image through a shifted
 * header onto the target grid + masked Pearson coefficient against the reference image on that grid).
 *
 * The reference (adolliou/euispice_coreg, pure Python) has no FFI; the seam this library replaces is
 *     Alignment._find_best_header_parameters()            euispice_coreg/hdrshift/alignment.py:613-797
 * i.e. everything below the public `align_using_helioprojective()` / `align_using_carrington()` calls.
 * Each entry point cites the reference code it stands in for.  All pointers are plain host pointers owned by
 * the caller unless a parameter says "device"; the library copies what it needs and never frees caller memory.
 * Every function returns COREG_OK (0) or a negative COREG_E* code; coreg_last_error() gives the text.
 * A handle is bound to one GPU and is not thread-safe; use one handle per thread / per process (one process
 * per GPU is the intended multi-GPU model: shard the raveled lag range with lag_begin/lag_end).
 */
#ifndef COREG_HIP_H
#define COREG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COREG_OK 0
#define COREG_EINVAL (-1)   /* bad argument */
#define COREG_EHIP (-2)     /* HIP runtime error */
#define COREG_ESTATE (-3)   /* call order: image / reference not set */
#define COREG_ENOTIMPL (-4) /* valid request the library does not implement */
#define COREG_ENOMEM (-5)

#define COREG_F32 0
#define COREG_F64 1
#define COREG_I32 2 /* coreg_encode_tiled_host only: the stored integers of an integer image */

#define COREG_PROJ_TAN 0
#define COREG_PROJ_CAR 1

#define COREG_METHOD_CORRELATION 0 /* alignment.py:522-542 */
#define COREG_METHOD_RESIDUS 1     /* alignment.py:544-547 (no NaN mask, quirk Q8) */

/* CDELT lag semantics */
#define COREG_CDELT_INTENDED 0  /* CDELTi += d, PC rebuilt with the new lambda (utils/Util.py:199-215) */
#define COREG_CDELT_REFERENCE 1 /* alignment.py:420-440 as written: d_cdelt1 only forces the PC rebuild; a
                                   non-zero d_cdelt2 kills the worker -> that lag-point is reported as NaN */

typedef struct coreg_handle coreg_handle;

/* The FITS keywords of one 2-D image header that the path reads (after alignment.py:580-611 has made sure
 * PCi_j and CROTA exist).  Angles are in the header's own unit; unit_to_deg converts them to degrees
 * (1/3600 for CUNIT='arcsec', 1 for 'deg') exactly as wcslib's unit fix does inside astropy.wcs.WCS. */
typedef struct coreg_wcs2d {
    int32_t naxis1, naxis2;
    double crpix1, crpix2;
    double crval1, crval2;
    double cdelt1, cdelt2;
    double pc1_1, pc1_2, pc2_1, pc2_2;
    double crota;       /* degrees: hdr['CROTA'] (else 'CROTA2'); roll for the Carrington transform          */
    double unit_to_deg; /* CUNIT1 == CUNIT2 scale                                                            */
    double lonpole;     /* degrees; FITS default for TAN is 180                                              */
    double dsun_obs;    /* metres          (Carrington only, utils/rectify.py:405)                           */
    double crln_obs;    /* degrees         (Carrington only, utils/rectify.py:406)                           */
    double crlt_obs;    /* degrees         (Carrington only, utils/rectify.py:407)                           */
    double latpole;     /* degrees; NaN = FITS default (90).  Only the CAR projection reads it                   */
    int32_t proj;       /* COREG_PROJ_TAN (HPLN-TAN / HPLT-TAN) or COREG_PROJ_CAR (CRLN-CAR / CRLT-CAR inputs of
                           align_using_initial_carrington, alignment.py:344-399).  For CAR a NaN lonpole means the
                           FITS default: 0 deg when CRVAL2 >= 0, else 180 deg                                    */
    int32_t reserved;
} coreg_wcs2d;

/* The five lag axes of Alignment.__init__ (alignment.py:47-55), already in header units
 * (alignment.py:819-837).  The sweep covers meshgrid(crval1, crval2, cdelt1, cdelt2, crota, indexing='ij')
 * and results are indexed in that C order (alignment.py:667-674). */
typedef struct coreg_lags {
    const double* crval1; int32_t n_crval1;
    const double* crval2; int32_t n_crval2;
    const double* cdelt1; int32_t n_cdelt1;
    const double* cdelt2; int32_t n_cdelt2;
    const double* crota;  int32_t n_crota; /* degrees */
} coreg_lags;

/* Carrington target grid of utils/rectify.py:875-878: lon = linspace(lon0, lon1, n_lon, float32),
 * lat = linspace(lat0, lat1, n_lat, float32); images on the grid are [n_lat][n_lon] row-major.
 * lat_cos/lat_sin (optional, may be NULL): caller-supplied float32 cos/sin of radians(lat) -- NumPy evaluates
 * these in float32 with a not-correctly-rounded SIMD routine (quirk Q6); a NumPy caller that passes its own
 * tables reproduces the reference's latitude trig bit for bit.  NULL = correctly rounded float32. */
typedef struct coreg_carr_grid {
    double lon0, lon1; int32_t n_lon;
    double lat0, lat1; int32_t n_lat;
    const float* lat_cos; /* [n_lat] or NULL */
    const float* lat_sin; /* [n_lat] or NULL */
} coreg_carr_grid;

typedef struct coreg_stats {
    double sweep_kernel_ms;    /* sum of HIP-event durations of the sweep kernel launches of the last sweep   */
    double precompute_ms;      /* grid-coordinate / compaction kernels of the last sweep                      */
    double total_gpu_ms;       /* first launch -> last launch of the last sweep, on the handle's stream       */
    int64_t n_lags;            /* lag-points computed by the last sweep (this handle's slice)                 */
    int64_t n_grid_points;     /* G                                                                           */
    int64_t n_active_points;   /* grid points that can overlap the small image for some lag (after culling)   */
    int64_t n_sweep_launches;
    int32_t small_is_f32;      /* 1 when the small image was stored as float32 (every value float32-exact)    */
    int32_t used_lds;          /* 1 when the LDS-staged gather path ran                                       */
} coreg_stats;

const char* coreg_version(void);

/* device < 0: current HIP device */
int coreg_create(coreg_handle** h, int device);
void coreg_destroy(coreg_handle* h);
const char* coreg_last_error(const coreg_handle* h);

/* Use an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) instead of the handle's own. */
int coreg_set_stream(coreg_handle* h, void* hip_stream);
int coreg_synchronize(coreg_handle* h);

/* Image to align, float64 [ny][nx], NaN = masked (alignment.py:314, after :844-887 thresholds).
 * Stored on the device as float32 when every finite value is exactly representable in float32 (true for
 * BITPIX=-32 / integer FITS data cast to float64), else as float64: arithmetic is float64 either way. */
int coreg_set_small(coreg_handle* h, const double* img, int32_t ny, int32_t nx);

/* Same, for callers that still hold the FITS BITPIX=-32 pixels as float32 (the float64 cast of alignment.py:314 is
 * exact, so the results are identical): no float64 copy, no exactness scan, pinned staging.  Used by the
 * jitter-correction session (jitter_correction/jitter_correction.py:101-138), which uploads one image per sweep. */
int coreg_set_small_f32(coreg_handle* h, const float* img, int32_t ny, int32_t nx);

/* Pixels exactly as a FITS data unit stores them: big-endian, BITPIX 8 (unsigned) / 16 / 32 / 64 (two's complement) /
 * -32 / -64 (IEEE), physical value = bscale * stored + bzero.  The reference decodes them on the host through
 * astropy.io.fits and casts to float64 (alignment.py:299-314, :191-208); here the RAW bytes cross PCIe (`data` may be a
 * read-only mmap of the file: it is copied into page-locked staging by a thread pool and is free again on return) and
 * the byte swap, BSCALE / BZERO and the float conversion run on the GPU.  Results are bit-identical to handing the
 * decoded float32 / float64 pixels to coreg_set_small[_f32] / coreg_prepare_reference_*[_f32]. */
typedef struct coreg_fits_pixels {
    const void* data; /* first byte of the data unit (ny * nx elements of |bitpix| / 8 bytes, row-major) */
    int32_t bitpix;
    int32_t reserved;
    double bscale, bzero; /* 1, 0 when the header has neither */
} coreg_fits_pixels;
int coreg_set_small_fits(coreg_handle* h, const coreg_fits_pixels* px, int32_t ny, int32_t nx);
int coreg_prepare_reference_carrington_fits(coreg_handle* h, const coreg_fits_pixels* px, int32_t ny, int32_t nx,
                                            const coreg_wcs2d* hdr_large, const coreg_carr_grid* grid, double solar_r,
                                            int order);
int coreg_prepare_reference_helioprojective_fits(coreg_handle* h, const coreg_fits_pixels* px, int32_t ny, int32_t nx,
                                                 const coreg_wcs2d* hdr_large, const coreg_wcs2d* hdr_small, int order);

/* Tile-compressed FITS images (the image lives in a binary table, one row per tile, the compressed bytes in the
 * table's heap): what EUI level-1 / level-2 files hold and the reference reads through astropy's CompImageHDU
 * (alignment.py:191-208, :299-314).  The COMPRESSED bytes cross PCIe and are decoded on the GPU, one thread per tile
 * (csrc/ricecomp.hpp: cfitsio's RICE_1 codec and float dequantization restated; pinned bit for bit against astropy /
 * cfitsio output, tests/golden/compressed_golden.npz).  The caller parses the table (utils/fits_io.py: open_compressed)
 * and hands over:
 *   heap / heap_bytes            the table's heap (host; may be a read-only mmap)
 *   tile_offset / tile_nbytes    [n_tiles] where each tile's Rice stream lies in the heap (COMPRESSED_DATA descriptors);
 *                                every tile must be Rice-coded (tiles cfitsio stored in GZIP_COMPRESSED_DATA: decode on
 *                                the host, coreg_decode_tiled_host + zlib, and upload the pixels instead)
 *   zbitpix, naxis1/2, ztile1/2  the image and its tiling (tiles in row-major order, edge tiles clipped)
 *   blocksize, bytepix           ZVALn of BLOCKSIZE / BYTEPIX
 *   quantize                     0 integer image; 1 NO_DITHER, 2 SUBTRACTIVE_DITHER_1, 3 SUBTRACTIVE_DITHER_2 (ZQUANTIZ)
 *   dither0                      ZDITHER0
 *   has_blank / blank            ZBLANK (quantized floats: -2147483647 when the file names none)
 *   zscale / zzero               [n_tiles] columns, or NULL with the scalars zscale0 / zzero0 (keywords)
 *   bscale / bzero               integer images: physical = float64(stored) * bscale + bzero when either differs from 1 / 0
 * Resident pixels: float32 for ZBITPIX = -32, else float64 tested for float32-exactness, exactly as the other uploads. */
typedef struct coreg_fits_tiled {
    const void* heap;
    int64_t heap_bytes;
    const int64_t* tile_offset;
    const int32_t* tile_nbytes;
    int32_t n_tiles;
    int32_t zbitpix;
    int32_t naxis1, naxis2, ztile1, ztile2;
    int32_t blocksize, bytepix;
    int32_t quantize;
    int32_t dither0;
    int32_t has_blank, blank;
    const double* zscale;
    const double* zzero;
    double zscale0, zzero0;
    double bscale, bzero;
} coreg_fits_tiled;
int coreg_set_small_tiled(coreg_handle* h, const coreg_fits_tiled* t);
int coreg_prepare_reference_carrington_tiled(coreg_handle* h, const coreg_fits_tiled* t, const coreg_wcs2d* hdr_large,
                                             const coreg_carr_grid* grid, double solar_r, int order);
int coreg_prepare_reference_helioprojective_tiled(coreg_handle* h, const coreg_fits_tiled* t, const coreg_wcs2d* hdr_large,
                                                  const coreg_wcs2d* hdr_small, int order);
/* The same decode on the host (no GPU; a few threads over the tiles): out = [naxis2][naxis1] float32 (dtype COREG_F32,
 * ZBITPIX = -32 only) or float64.  tile_status (optional, [n_tiles]): 0 decoded, 1 corrupt stream, 2 not Rice-coded
 * (left untouched in `out`).  Returns COREG_OK unless an argument is bad. */
int coreg_decode_tiled_host(const coreg_fits_tiled* t, void* out, int dtype, int32_t* tile_status);
/* The writing side (host, no GPU): what astropy's CompImageHDU does when the reference writes a corrected file whose
 * input was tile-compressed (utils/Util.py:137-138) -- cfitsio's RICE_1 encoder (`fits_rcomp*`) restated, and for
 * floating-point pixels (dtype COREG_F32 / COREG_F64, BYTEPIX 4) its quantization q = NINT((v - ZZERO) / ZSCALE + r - 0.5)
 * with ZZERO = the tile's smallest finite value and ZSCALE = `scale` for every tile (quantize 1 / 2 / 3 as above; NaN ->
 * -2147483647).  Integer images: dtype COREG_I32, the STORED integers (after BZERO), BYTEPIX 1 / 2 / 4.  Tiles in
 * row-major order; heap (capacity heap_cap; <= pixels * BYTEPIX + n_blocks + 8 bytes per tile always suffices),
 * tile_nbytes / tile_offset [n_tiles] and, for floats, zscale / zzero [n_tiles] are filled, *heap_used set.
 * utils/fits_io.write_compressed_image builds the table around it.  COREG_ENOMEM: heap_cap too small; COREG_EINVAL also
 * when a tile's range does not fit 32-bit integers at this scale. */
int coreg_encode_tiled_host(const void* pixels, int dtype, int ny, int nx, int tile_x, int tile_y, int bytepix,
                            int blocksize, int quantize, int dither0, double scale, unsigned char* heap,
                            long long heap_cap, int32_t* tile_nbytes, int64_t* tile_offset, double* zscale, double* zzero,
                            long long* heap_used);

/* Alignment._set_threshold_minmax_to_nan (alignment.py:876-887) on the resident image to align:
 * |v| < vmin -> NaN when has_min, |v| > vmax -> NaN when has_max.  *n_finite (optional) receives the number of finite
 * pixels left: 0 is the reference's "minimum or maximum value have set all small FOV to nan" error (alignment.py:655). */
int coreg_threshold_small(coreg_handle* h, int has_min, double vmin, int has_max, double vmax, long long* n_finite);

/* Reference image already resampled on the target grid (what alignment.py:646-651 leaves in data_large),
 * [gy][gx], dtype COREG_F32 (helioprojective sub-map, alignment.py:995) or COREG_F64 (Carrington). */
int coreg_set_reference_on_grid(coreg_handle* h, const void* ref, int dtype, int32_t gy, int32_t gx);

/* `order` everywhere below: the spline order of scipy.ndimage.map_coordinates(..., prefilter=False) (utils/Util.py:98-102,
 * Alignment(reprojection_order=...), alignment.py:54), 0..5; 2 (the default of the reference) and 1 run on the tuned
 * kernels.
 *
 * Once-per-sweep reference preparation on the GPU.
 * carrington:      alignment.py:646-648 -> :889-901  (large image -> Carrington grid, float64)
 * helioprojective: alignment.py:649-651 -> :987-1000 (large image -> small header's pixel grid, float32).  Both headers
 *                  TAN, or both CAR: two Carrington maps, align_using_initial_carrington -- BOTH branches of the
 *                  reference build this sub-map for that frame (alignment.py:649-651 and :765-767). */
int coreg_prepare_reference_carrington(coreg_handle* h, const double* large, int32_t ny, int32_t nx,
                                       const coreg_wcs2d* hdr_large, const coreg_carr_grid* grid, double solar_r,
                                       int order);
int coreg_prepare_reference_helioprojective(coreg_handle* h, const double* large, int32_t ny, int32_t nx,
                                            const coreg_wcs2d* hdr_large, const coreg_wcs2d* hdr_small, int order);
/* The same with the reference image handed over as the float32 pixels a BITPIX=-32 FITS file holds (the reference
 * casts them to float64, alignment.py:191 / :301 -- exactly, so results are identical): half the bytes over PCIe. */
int coreg_prepare_reference_carrington_f32(coreg_handle* h, const float* large, int32_t ny, int32_t nx,
                                           const coreg_wcs2d* hdr_large, const coreg_carr_grid* grid, double solar_r,
                                           int order);
int coreg_prepare_reference_helioprojective_f32(coreg_handle* h, const float* large, int32_t ny, int32_t nx,
                                                const coreg_wcs2d* hdr_large, const coreg_wcs2d* hdr_small, int order);

/* The same three uploads with the pixels ALREADY IN DEVICE MEMORY of the handle's GPU (dtype COREG_F32 or COREG_F64):
 * the multi-GPU form of the hand-over, where every rank sends 1/N of each image over its own PCIe link and one RCCL
 * all-gather over xGMI assembles the replicas (euispice_coreg_amd/parallel.py: replicate_image; the reference hands its
 * workers the images through shared memory, alignment.py:692-720).  The buffer is read by work enqueued on the handle's
 * stream: whatever produced it must be ordered before that stream (same stream, or synchronised), and it must stay valid
 * until that work has run (coreg_synchronize).  The image to align is copied; the reference source is only read. */
int coreg_set_small_from_device(coreg_handle* h, const void* dev_img, int dtype, int32_t ny, int32_t nx);
int coreg_prepare_reference_carrington_from_device(coreg_handle* h, const void* dev_large, int dtype, int32_t ny,
                                                   int32_t nx, const coreg_wcs2d* hdr_large,
                                                   const coreg_carr_grid* grid, double solar_r, int order);
int coreg_prepare_reference_helioprojective_from_device(coreg_handle* h, const void* dev_large, int dtype, int32_t ny,
                                                        int32_t nx, const coreg_wcs2d* hdr_large,
                                                        const coreg_wcs2d* hdr_small, int order);

/* Copy the resident reference-on-grid back (tests / figures). out: [gy][gx] of `dtype` (must match). */
int coreg_get_reference_on_grid(coreg_handle* h, void* out, int dtype);

/* One resample of the small image through ONE header (= function_to_apply of one lag-point):
 * carrington:      alignment.py:889-901   out float64 [n_lat][n_lon], NaN outside
 * helioprojective: alignment.py:1018-1029 out float32 [hdr_target.naxis2][hdr_target.naxis1] */
int coreg_resample_carrington(coreg_handle* h, const coreg_wcs2d* hdr, const coreg_carr_grid* grid, double solar_r,
                              int order, double* out);
int coreg_resample_helioprojective(coreg_handle* h, const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr,
                                   int order, float* out);
/* same map, float64 destination: the once-only re-grid of the small image onto a regular sub-FOV grid
 * (alignment.py:1120-1126, dst = zeros_like(xg)) */
int coreg_resample_helioprojective_f64(coreg_handle* h, const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr,
                                       int order, double* out);

/* The sweep: replaces the per-lag loop alignment.py:470-578 + :613-797 for one lag_solar_r value.
 * hdr_small is the UNSHIFTED header of the image to align; its CRVAL/CDELT/CROTA are the *_ref values of
 * alignment.py:799-814 and each lag-point applies _shift_header (alignment.py:401-468) to it.
 * [lag_begin, lag_end) selects a contiguous slice of the C-order raveled lag index (np.array_split-style
 * sharding, alignment.py:677-687); corr_out receives lag_end - lag_begin float64 values
 * (host memory, or device memory when out_on_device != 0).  With a host buffer the call returns when the values are
 * there; with a device buffer it only enqueues the work on the handle's stream (coreg_set_stream) and returns:
 * anything ordered after it on that stream -- an RCCL all-gather, a copy -- sees the results.
 * Headers, grids and lags that cannot give finite pixel coordinates -- a NaN / infinite CRPIX / CRVAL / CDELT / CROTA /
 * PCi_j card (or one beyond 1e12 in magnitude), CDELT = 0 (or below 1e-30), a singular PCi_j, unit_to_deg <= 0, a non-finite LONPOLE on a TAN header, DSUN_OBS <= 0 or a
 * non-finite CRLN_OBS / CRLT_OBS (Carrington transform), a non-finite grid limit or lag, solar_r <= 0, an image or grid
 * of more than 2^31 - 1 pixels / points -- are refused with
 * COREG_EINVAL by every sweep, resample and reference preparation before anything is planned or launched (the reference
 * hands such a header to astropy, which raises, or returns NaN everywhere).  A lag-point whose OWN shifted header
 * degenerates (CDELT + d_cdelt = 0) stays NaN, like one whose worker dies in the reference (coreg_shift_header
 * returns 2 for it). */
int coreg_sweep_carrington(coreg_handle* h, const coreg_wcs2d* hdr_small, const coreg_carr_grid* grid,
                           double solar_r, const coreg_lags* lags, int order, int method, int cdelt_semantics,
                           int64_t lag_begin, int64_t lag_end, double* corr_out, int out_on_device);
/* hdr_target: header whose pixel grid the reference image lives on (the unshifted small header in the
 * parallelism=True path, alignment.py:1000; the large header in the serial path, quirk Q1).  Both headers TAN
 * (helioprojective), or both CAR: Carrington maps, align_using_initial_carrington (alignment.py:344-399), hdr_target =
 * the reference map's header. */
int coreg_sweep_helioprojective(coreg_handle* h, const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr_small,
                                const coreg_lags* lags, int order, int method, int cdelt_semantics,
                                int64_t lag_begin, int64_t lag_end, double* corr_out, int out_on_device);

/* Multi-GPU sharding of the GRID instead of the lag range, for sweeps with few lag-points per GPU (SURVEY 8e fallback):
 * with coreg_set_option(h, "shard_world", W) and ("shard_rank", r) a sweep call covers ALL its lag-points over rank
 * r's share of the target grid's points (tile groups r*G/W .. (r+1)*G/W of k_sweep's work partition, identical on
 * every rank) and, instead of coefficients, leaves the six sums n, Sa, Sb, Saa, Sbb, Sab of every lag slot on the
 * device (corr_out of the sweep call is only NaN-filled).  The caller adds the ranks' sums element-wise -- one
 * all-reduce(SUM) of coreg_sums_size() doubles, the data-path collective of this mode (the reference's counterpart
 * is the lock + shared-memory scatter of alignment.py:502-506) -- and hands the result back:
 *   coreg_sums_size      number of doubles of the pending sums
 *   coreg_copy_sums      copy them out (device or host destination)
 *   coreg_finalize_sums  reduced sums in (device or host) -> Pearson coefficients / residus, C-order lag slice out */
int coreg_sums_size(coreg_handle* h, int64_t* n_doubles);
int coreg_copy_sums(coreg_handle* h, double* dst, int dst_on_device);
int coreg_finalize_sums(coreg_handle* h, const double* sums, int sums_on_device, double* corr_out, int out_on_device);

/* The two pivots the Pearson moments are taken about: [0] mean of the finite reference values on the grid, [1] mean
 * of the finite pixels of the image to align (set by the upload / preparation calls).  Any pivot gives the same
 * coefficient up to rounding, but the six sums of DIFFERENT ranks only add up if every rank used the same two values:
 * a point-sharded multi-GPU sweep reads rank 0's pivots and sets them on every rank before sweeping (the pivots are
 * device_mean results, identical on identical GPUs, but nothing else enforces it). */
int coreg_get_pivots(coreg_handle* h, double* pivots2);
int coreg_set_pivots(coreg_handle* h, const double* pivots2);

/* Waits for an in-flight device-output sweep (those return without synchronising the stream). */
int coreg_last_stats(coreg_handle* h, coreg_stats* out);

/* Diagnostics.  counts6[0..3]: (tile, lag batch) visits of the sweep kernel's workgroups in the LAST launch of the last
 * sweep -- all; gathered from an LDS window; of those, "interior" (every sample inside the image: no bounds rule); of
 * those, all-finite windows (no sample mask either).  counts6[4]: lag-points of the WHOLE last sweep whose six sums
 * were too ill-conditioned for the one-pass Pearson formula and were re-evaluated about their own means ("refine");
 * counts6[5]: lag-points that were flagged but kept their one-pass value -- 0: there is no cap on the re-evaluations,
 * and a lag-point whose noise-decided samples were taken out of (put into) its sums by the extra slab of a
 * helioprojective launch (the zero lag, single samples of odd orders) is re-evaluated with those samples taken about its
 * own pivots as well.  (The field counts such lag-points where that second slab is not available; it stays for ABI
 * stability.)  Waits for the stream. */
int coreg_last_visit_counts(coreg_handle* h, int64_t* counts6);
/* Odd spline orders in the helioprojective frame ("tap_fix", below): counts3[0] samples of the last sweep whose mapped
 * coordinate lay within 1e-8 px of an integer and were re-evaluated with wcslib's own arithmetic, counts3[1] lag-points
 * concerned, counts3[2] = 1 when the list exceeded "tap_cap" and nothing was applied. */
int coreg_last_tap_fix(coreg_handle* h, int64_t* counts3);

/* Options (name -> integer value), in three groups.  A caller of the drop-in API needs NONE of them: every default is
 * the behaviour that reproduces the reference; euispice_coreg_amd/hdrshift sets only "async_upload" (when a reference
 * preparation follows the upload) and the sharding options of group B.
 *
 * A. USER-FACING -- change what is computed or how inputs are handed over:
 *   "refine"       1 (default) lag-points whose sums are ill-conditioned (sum xx / (n var) > 1e5: a handful of samples, an
 *                  overlap inside a flat region) are re-evaluated with sums centred on the lag-point's own means and the
 *                  corrected two-pass formula -- the accuracy of c_correlate.py:39-72's means-first evaluation -- by
 *                  kernels of their own, EVERY flagged lag-point, also in grid-sharded multi-GPU sweeps
 *                  (coreg_finalize_sums flags from the reduced sums); 0: the one-pass formula everywhere
 *   "border_fix"   1 (default) lag-points whose map leaves an image axis invariant (the zero lag, CDELT-only lags) have
 *                  their border pixels -- and, for odd spline orders, every pixel's tap set -- decided as the reference's
 *                  wcslib round trip decides them; 0: the exact map decides
 *   "tap_fix"      1 (default): helioprojective / plate-carree sweeps re-evaluate, with wcslib's own arithmetic on the
 *                  host, the single samples whose fate hangs on the sign of the rounding noise of the reference's round
 *                  trip (alignment.py:1038-1069).  Odd spline orders: every sample whose mapped coordinate comes back
 *                  within 1e-8 px of an integer -- the sign picks the taps, hence which neighbour's NaN poisons the
 *                  sample.  Even orders: only those within 1e-8 px of a BOUND of the image -- the sign decides the bounds
 *                  rule c < 0 or c > n - 1.  One scan kernel before the sweep (0.07 ms on a 2048^2 grid x 3721 lags,
 *                  0.4 ms on a 192 x 832 raster x 78 141 lags), one correction kernel after it when anything was listed;
 *                  0: the exact map's coordinate decides
 *   "tap_cap"      2^24 (default): most samples listed per sweep; beyond it nothing is applied (coreg_last_tap_fix)
 *   "crop_reference" 1 (default) coreg_prepare_reference_* upload only the rectangle of the reference image the target
 *                  grid can touch (bounding box of the sample coordinates, computed on the GPU; identical results),
 *                  0 the whole image
 *   "overlap_upload"  1 (default) coreg_set_small_f32 / coreg_set_small_fits (BITPIX -32) put the image to align on an
 *                  upload stream of the handle's own: a coreg_prepare_reference_* called next does not queue behind the
 *                  image's DMA; the first call that reads the image joins the streams.  0: everything on one stream
 *   "async_upload" 0 (default).  1: those two calls return as soon as the upload is QUEUED on the handle's upload thread
 *                  (staging copies + DMA run there while the caller prepares the reference and plans the sweep).  The
 *                  caller's pixel buffer must then stay valid and unchanged until the next call on this handle that reads
 *                  the image (a sweep, coreg_threshold_small, coreg_synchronize, ...) has returned
 *
 * B. MULTI-GPU PLUMBING -- set by the host side that shards a sweep (parallel.py, csrc/multi.hpp), not by users:
 *   "shard_world", "shard_rank"  grid shares: above (coreg_copy_sums / coreg_finalize_sums)
 *   "combo_begin", "combo_end"  sharding by (cdelt1, cdelt2, crota) combination: the NEXT sweep call covers
 *                  only the combinations [begin, end) of the inner C-order index (i_cdelt1 * n_cdelt2 + i_cdelt2) * n_crota
 *                  + i_crota; its output (and lag_begin / lag_end) is the C-order array [n_crval1][n_crval2][end - begin].
 *                  One-shot: taken off the handle at the very top of that call, before any validation -- it leaves
 *                  "all combinations" behind whether it succeeds, fails late or fails at once.
 *
 * C. TEST / TUNING KNOBS -- identical results (to summation rounding) at any setting; they exist so that the tests can
 *    hold one code path against another and the profiles can sweep a parameter.  Not part of the drop-in surface:
 *   "use_lds"      1 (default) stage the gather window in LDS, 0 gather from global memory
 *   "clean_path"   1 (default) interior visits whose LDS window holds only finite values skip the per-sample mask and
 *                  take the count and the reference moments from per-chunk sums; 0: always the masked arithmetic
 *   "h_series"     1 (default) helioprojective maps with |projective term| < 4e-6 invert 1 + eps as 1 - eps + eps^2
 *                  (exact to float64 there) instead of dividing; 0 always divide
 *   "h_incr"       1 (default) order-2 homography sweeps fold the LDS window offset into each lane's map once per tile
 *                  visit and advance the map's affine terms by additions along runs of a grid row (round 6; maps equal
 *                  to 1e-11); 0: every sample evaluates the map from the pixel
 *   "refine_cond_log10"  5 (default): log10 of the "refine" threshold; -1 re-evaluates every lag-point
 *   "refine_max"   accepted and ignored (round 4 capped the re-evaluations per block of lag slots; no cap any more)
 *   "tap_nan_filter"  2 (default) the odd-order pass ("tap_fix") lists only the near-integer samples that can change the
 *                  result: on the bounds rule, or -- one axis near an integer -- with the taps the two footprints share
 *                  all finite and exactly one of the two end lines not, or -- both axes -- with a non-finite pixel in the
 *                  union of the footprints; 1: the union test everywhere; 0: all near-integer samples
 *   "tile_skip"    1 (default) Carrington precompute drops whole tiles that provably miss the cull box; 0 evaluates them
 *   "tile_w"       0 (default, auto) or a power of two in [4, 256]: grid-tile width in points (tile = 1024 pts)
 *   "n_groups"     0 (default, auto): tile groups (partial-sum slabs) per lag batch
 *   "lds_bytes"    dynamic LDS per workgroup for the float64 gather window (default and max 159 KiB)
 *   "pitch"        -1 (default, auto): compile-time LDS row pitch of the order-2 / order-3 Carrington kernels; 0 per-visit
 *   "skew"         accepted and ignored (a row skew against bank conflicts, measured slower in round 2)
 *   "patch_w"      0 (default, auto) or the maximum width, in CRVAL1 lags, of a workgroup's lag patch
 *   "taper_frac"   -1 (default, automatic) / 0 equal shares of the grid points per tile group / n: the last n/1024 of the
 *                  groups get linearly smaller shares (down to "taper_min"/1024, default 128) and the others more;
 *                  automatic = 512 for launches of at least "taper_rounds" (default 6) rounds of workgroups and at least
 *                  128 tile groups
 * coreg_multi_set_option: "image_shares" (A: 1 = row shares + one all-gather of the image to align when RCCL is in use,
 * 0 = every device copies the whole image), "force_mode" (C: one partition on any lag set).
 * Returns COREG_EINVAL for unknown names. */
int coreg_set_option(coreg_handle* h, const char* name, int64_t value);

/* Host-only helpers (no GPU work; usable where no device exists): the header arithmetic the sweep applies per
 * lag-point, exported so that it can be checked against the reference's on CPU.
 *   coreg_shift_header      alignment.py:401-468 (_shift_header); returns 1 when COREG_CDELT_REFERENCE would kill
 *                           the worker (d_cdelt2 != 0), 2 when the shifted header has CDELT = 0 or a non-finite
 *                           CDELT / CROTA / PCi_j (no header left to evaluate), else 0
 *   coreg_homography        0-based pixels of `from` -> 0-based pixels of `to` through the sky, row-major 3x3 with
 *                           h[8] = 1: WCS(to).world_to_pixel(WCS(from).pixel_to_world(p)), alignment.py:1041-1065
 *   coreg_lag_homography    the map the helioprojective sweep gives lag-point idx = (i_crval1, i_crval2, i_cdelt1,
 *                           i_cdelt2, i_crota): pixels of hdr_target -> pixels of _shift_header(hdr_small, lag)
 *                           (factored evaluation used by the sweep); returns 1 for a lag the reference cannot evaluate
 *   coreg_carrington_origin X0, Y0 of utils/rectify.py:399-404 */
int coreg_shift_header(const coreg_wcs2d* ref, double d_crval1, double d_crval2, double d_cdelt1, double d_cdelt2,
                       double d_crota, int cdelt_semantics, coreg_wcs2d* out);
int coreg_homography(const coreg_wcs2d* from, const coreg_wcs2d* to, double* h9);
int coreg_lag_homography(const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr_small, const coreg_lags* lags,
                         const int32_t idx[5], int cdelt_semantics, double* h9);
/* alignment.py:1038-1069 + utils/Util.py:282-312 for n pixels of header `from`, through wcslib's own arithmetic
 * (restated operation by operation, csrc/geometry.hpp WcslibTan; pinned bit for bit by tests/golden/border_golden.npz):
 * pixel -> sky (`from`) -> ang2pipi -> pixel (`to`).  lng / lat (optional): the sky coordinates in degrees.  This is
 * what decides the border pixels of the zero lag of a helioprojective sweep; every other lag uses the exact
 * homography (coreg_lag_homography). */
int coreg_wcslib_pixel_to_pixel(const coreg_wcs2d* from, const coreg_wcs2d* to, int64_t n, const double* px,
                                const double* py, double* ox, double* oy, double* lng, double* lat);
int coreg_carrington_origin(const coreg_wcs2d* hdr, double* x0, double* y0);
/* CAR -> CAR map of one lag-point on the host: 0-based pixels of `from` -> 0-based pixels of `to` through the common
 * sphere, WCS(to).world_to_pixel(WCS(from).pixel_to_world(p)) for CRLN-CAR / CRLT-CAR headers (alignment.py:1038-1069
 * on the align_using_initial_carrington path).  Returns 1 when either header has no valid native pole (astropy raises
 * InvalidTransformError). */
int coreg_car_map(const coreg_wcs2d* from, const coreg_wcs2d* to, int64_t n, const double* px, const double* py,
                  double* ox, double* oy);
/* The allowance, in pixels of the shifted map, by which the sweep kernel widens the bounding box of a tile's four mapped
 * corners on the CAR -> CAR path (the map is not projective): tile = tile_w x (1024 / tile_w) target pixels whose
 * native latitudes reach tile_abs_lat_rad in absolute value.  +inf = no box is trusted for that tile (polar tiles: the
 * kernel samples them point by point from global memory).  Exported so that the bound can be checked against
 * coreg_car_map on the host.  Returns 1 when a header has no valid native pole. */
int coreg_car_tile_margin(const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr_shifted, int32_t tile_w,
                          double tile_abs_lat_rad, double* margin_px);


/* SPICE preparation (host only, no GPU): np.nansum(float64(cube)[0, selected], axis=0) of
 * AlignmentSpice._prepare_spice_from_l2 (hdrshift/alignment_spice.py:250-323) straight from the FITS data unit: `cube`
 * points at big-endian elements (BITPIX -32 or -64), plane k of `n_sel` selected planes starts at element
 * plane_index[k] * n_pixels.  Every pixel is accumulated in float64 over the planes IN THE ORDER GIVEN, NaN counted as
 * +0.0 -- the sequential reduction NumPy performs over an outer axis, hence the same sums to the bit -- by a few
 * threads that split the pixels.  out: n_pixels doubles (all +0.0 when n_sel is 0). */
int coreg_nansum_planes_be(const void* cube, int32_t bitpix, int64_t n_pixels, const int64_t* plane_index, int32_t n_sel,
                           double* out);

/* Sub-lag refinement of the correlation peak (host only, no GPU): the bounded least-squares fit of
 *     g(x, y) = offset + amplitude * exp(-((x - xo)^2 / (2 sigma_x^2) + (y - yo)^2 / (2 sigma_y^2)))
 * that AlignmentResults._compute_shift (hdrshift/AlignmentResults.py:12-21, :218-341) hands to
 * scipy.optimize.curve_fit(f=twoD_Gaussian, xdata, ydata, p0, bounds) -- i.e. least_squares(method='trf',
 * jac='2-point', x_scale=1, ftol = xtol = gtol = 1e-8, max_nfev = 600).  csrc/fit.hpp restates that algorithm
 * (third-party scipy; trust-region reflective, exact SVD trust-region solver, scipy's forward-difference step rule and
 * termination tests) so that the fit stops where the reference's call stops.
 *   m points (x, y, z), m <= 64; p0 / lb / ub / popt: (amplitude, xo, yo, sigma_x, sigma_y, offset);
 *   jac: 0 = forward differences as scipy's default (the reference's call), 1 = analytic Jacobian;
 *   ftol / xtol / gtol <= 0 and max_nfev <= 0 select scipy's defaults.
 *   *status: scipy's termination status (1 gtol, 2 ftol, 3 xtol, 4 ftol and xtol, 0 = max_nfev reached: curve_fit
 *   raises RuntimeError there), -1 = a non-finite value among the inputs or the residuals at p0 (curve_fit raises
 *   ValueError: the reference falls back to the argmax).  Returns COREG_EINVAL for bad arguments / p0 outside the bounds. */
int coreg_fit_gaussian2d(int32_t m, const double* x, const double* y, const double* z, const double* p0,
                         const double* lb, const double* ub, int32_t jac, double ftol, double xtol, double gtol,
                         int32_t max_nfev, double* popt, int32_t* nfev, int32_t* status);


/* ---- All GPUs of the node from ONE process ------------------------------------------------------------------------
 * The reference's `Alignment(..., parallelism=True, counts_cpu_max=N)` uses the whole machine from a plain
 * `python script.py`: its lag loop is fanned out over a process pool (hdrshift/alignment.py:692-744, README.md:47-87).
 * A coreg_multi is the same from one process on this library: one host thread + one coreg_handle (own stream and
 * buffers) per GPU, full image replicas (one shared page-locked staging buffer, every device copies from it over its
 * own PCIe link), the lag set cut exactly as euispice_coreg_amd/parallel.py cuts it for the one-process-per-GPU form --
 * blocks of the (CRVAL1, CRVAL2) plane; contiguous slices of the raveled index when the plane is smaller than the
 * number of GPUs (alignment.py:677-687); shares of the GRID below 128 lag-points per GPU -- per-device sweeps with
 * device outputs and ONE collective over xGMI: an RCCL all-gather of the per-lag coefficients (all-reduce of the six
 * sums per lag in the grid-share mode), one communicator per device from ncclCommInitAll, all devices' calls between
 * ncclGroupStart / ncclGroupEnd on the handles' own streams; device 0 hands the map to the host.  RCCL is looked up at
 * run time (dlopen; a copy already loaded in the process, e.g. PyTorch's, first).  Without it -- or when
 * COREG_VIRTUAL_DEVICES=k maps k logical devices onto the GPUs present (tests of this path on a one-GPU box; RCCL
 * refuses two ranks on one device) -- every device copies its block to the host instead.
 * The image to align crosses PCIe once in all when RCCL is in use: device k uploads rows [k * ceil(H / N), ...) over its
 * own link and ONE all-gather over xGMI assembles the replica on every GPU (coreg_multi_set_option "image_shares" 0, a
 * failing collective, or no RCCL: the image is staged once in page-locked memory and every device copies all of it).
 * n_devices 0 = all visible (or COREG_VIRTUAL_DEVICES); device_ids NULL = 0 .. n-1.  The calls mirror the single-device
 * ones (dtype = COREG_F32 / COREG_F64 of the host pixels); sweeps return the WHOLE map, C order, in host memory. */
typedef struct coreg_multi coreg_multi;
int coreg_device_count(void); /* visible GPUs, or COREG_VIRTUAL_DEVICES when set */
int coreg_physical_device_count(void); /* visible GPUs (logical device L of a virtual set is physical device L mod this) */
int coreg_multi_create(coreg_multi** m, int n_devices, const int* device_ids);
void coreg_multi_destroy(coreg_multi* m);
int coreg_multi_size(const coreg_multi* m);
coreg_handle* coreg_multi_handle(coreg_multi* m, int k); /* device k's context (options, single-device utilities) */
const char* coreg_multi_last_error(const coreg_multi* m);
const char* coreg_multi_collective(const coreg_multi* m); /* "rccl", "host-copy" or "none" (what sweeps use / used) */
/* "ok", "not used" (one device, virtual devices, no library, COREG_MULTI_COLLECTIVE=host), or why RCCL was given up: the
 * communicators are tried once at creation with a known pattern (bounded wait), and a failing collective during a
 * sweep makes the handle fall back to host copies for that sweep and all later ones */
const char* coreg_multi_rccl_status(const coreg_multi* m);
int coreg_multi_last_mode(const coreg_multi* m);          /* partition of the last sweep: 0 none, 1 blocks, 2 slices, 3 points, 4 combos */
int coreg_multi_set_option(coreg_multi* m, const char* name, int64_t value);
int coreg_multi_set_small(coreg_multi* m, const void* img, int dtype, int32_t ny, int32_t nx);
/* the same with the pixels as the FITS data unit stores them (coreg_fits_pixels above) */
int coreg_multi_set_small_fits(coreg_multi* m, const coreg_fits_pixels* px, int32_t ny, int32_t nx);
int coreg_multi_prepare_reference_carrington_fits(coreg_multi* m, const coreg_fits_pixels* px, int32_t ny, int32_t nx,
                                                  const coreg_wcs2d* hdr_large, const coreg_carr_grid* grid,
                                                  double solar_r, int order);
int coreg_multi_prepare_reference_helioprojective_fits(coreg_multi* m, const coreg_fits_pixels* px, int32_t ny,
                                                       int32_t nx, const coreg_wcs2d* hdr_large,
                                                       const coreg_wcs2d* hdr_small, int order);
int coreg_multi_set_small_tiled(coreg_multi* m, const coreg_fits_tiled* t);
int coreg_multi_prepare_reference_carrington_tiled(coreg_multi* m, const coreg_fits_tiled* t, const coreg_wcs2d* hdr_large,
                                                   const coreg_carr_grid* grid, double solar_r, int order);
int coreg_multi_prepare_reference_helioprojective_tiled(coreg_multi* m, const coreg_fits_tiled* t,
                                                        const coreg_wcs2d* hdr_large, const coreg_wcs2d* hdr_small,
                                                        int order);
int coreg_multi_threshold_small(coreg_multi* m, int has_min, double vmin, int has_max, double vmax, long long* n_finite);
int coreg_multi_set_reference_on_grid(coreg_multi* m, const void* ref, int dtype, int32_t gy, int32_t gx);
int coreg_multi_prepare_reference_carrington(coreg_multi* m, const void* large, int dtype, int32_t ny, int32_t nx,
                                             const coreg_wcs2d* hdr_large, const coreg_carr_grid* grid, double solar_r,
                                             int order);
int coreg_multi_prepare_reference_helioprojective(coreg_multi* m, const void* large, int dtype, int32_t ny, int32_t nx,
                                                  const coreg_wcs2d* hdr_large, const coreg_wcs2d* hdr_small, int order);
int coreg_multi_sweep_carrington(coreg_multi* m, const coreg_wcs2d* hdr_small, const coreg_carr_grid* grid, double solar_r,
                                 const coreg_lags* lags, int order, int method, int cdelt_semantics, double* corr_out);
int coreg_multi_sweep_helioprojective(coreg_multi* m, const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr_small,
                                      const coreg_lags* lags, int order, int method, int cdelt_semantics, double* corr_out);
int coreg_multi_last_stats(coreg_multi* m, int k, coreg_stats* out);
/* The partition a sweep over (n_crval1, n_crval2, n_inner = n_cdelt1 n_cdelt2 n_crota) lag-points gets on `world` GPUs
 * (host-only; checked against euispice_coreg_amd/parallel.py): mode as coreg_multi_last_mode; the GPUs form a
 * g_combo x (g1 x g2) grid: the inner combinations are dealt in g_combo contiguous runs, the (CRVAL1, CRVAL2) plane is
 * cut in g1 x g2 blocks (mode 1 "blocks": g_combo = 1; mode 4 "combos": g_combo > 1).  per_combo_launch: 1 for the
 * Carrington / plate-carree sweeps (one precompute + launch per combination), 0 for helioprojective (TAN) sweeps (one
 * launch whatever the lag set): it enters the planner's cost model. */
int coreg_multi_plan(int32_t n_crval1, int32_t n_crval2, int64_t n_inner, int32_t world, int32_t per_combo_launch,
                     int32_t* mode, int32_t* g_combo, int32_t* g1, int32_t* g2);

#ifdef __cplusplus
}
#endif
#endif /* COREG_HIP_H */
