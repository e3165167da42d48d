"""
`Alignment` -- drop-in for euispice_coreg.hdrshift.Alignment (hdrshift/alignment.py:45-399) whose lag sweep
(`_find_best_header_parameters`, alignment.py:613-797) runs on MI355X through libcoreg_hip.so.

Same constructor keywords, same `align_using_helioprojective()` / `align_using_carrington()` signatures, same
6-D correlation array layout [crval1, crval2, cdelt1, cdelt2, crota, solar_r] and the same `AlignmentResults`
surface.  Differences, all deliberate and listed in DESIGN.md:
  * `parallelism` no longer chooses between a process pool and a Python loop: both values run on the GPU.  It keeps
    its *numerical* meaning for the helioprojective frame: True = reference image resampled once
    on the small image's own pixel grid, float32 (alignment.py:649-651); False = correlate on the full large-FOV
    grid, float64 reference (alignment.py:765).
  * `counts_cpu_max`, `display_progress_bar`, `path_save_figure` are accepted and ignored.
  * CDELT lags follow the intended semantics of utils/Util.py:199-215 (quirk Q2) unless
    `cdelt_semantics="reference"`.
  * inputs may be FITS paths or (data, header) pairs; astropy is optional.
  * with torch.distributed initialised (one process per GPU) the lag grid is sharded and all-gathered
    (euispice_coreg_amd/parallel.py); without it, `parallelism=True` drives every visible GPU from this one process
    (include/coreg_hip.h: coreg_multi) -- the unchanged user script uses the whole node, as it does with the reference.
There is no CPU fallback: without the HIP library / a GPU the sweep raises.
"""
from __future__ import annotations

import copy
import os
import warnings

import numpy as np

from .. import _lib, parallel
from ..utils import fits_io, header as hdrutil, wcs_tan
from .alignment_results import AlignmentResults


class Alignment:

    def __init__(self, large_fov_known_pointing, small_fov_to_correct, lag_crval1, lag_crval2, lag_cdelt1, lag_cdelt2,
                 lag_crota, lag_solar_r=None, small_fov_value_min=None, parallelism=False, display_progress_bar=False,
                 small_fov_value_max=None, counts_cpu_max=40, large_fov_window=-1, small_fov_window=-1,
                 path_save_figure=None, reprojection_order=2, force_crota_0=False, unit_lag="arcsec",
                 cdelt_semantics="intended", device=None):
        self.large_fov_known_pointing = large_fov_known_pointing
        self.small_fov_to_correct = small_fov_to_correct
        self.lag_crval1 = lag_crval1
        self.lag_crval2 = lag_crval2
        self.lag_cdelt1 = lag_cdelt1
        self.lag_cdelt2 = lag_cdelt2
        self.lag_crota = lag_crota
        self.lag_solar_r = lag_solar_r
        self.unit_lag = unit_lag
        self.unit_lag_input = copy.deepcopy(unit_lag)
        self.lonlims = None
        self.latlims = None
        self.shape = None
        self.reference_date = None
        self.parallelism = parallelism
        self.small_fov_window = small_fov_window
        self.large_fov_window = large_fov_window
        self.counts = counts_cpu_max
        self.small_fov_value_min = small_fov_value_min
        self.small_fov_value_max = small_fov_value_max
        self.path_save_figure = path_save_figure
        self.display_progress_bar = display_progress_bar
        self.force_crota_0 = force_crota_0
        self.order = reprojection_order
        self.method = None
        self.coordinate_frame = None
        self.data_large = self.data_small = self.hdr_large = self.hdr_small = None
        if cdelt_semantics not in ("intended", "reference"):
            raise ValueError("cdelt_semantics must be 'intended' or 'reference'")
        self.cdelt_semantics = cdelt_semantics
        self.device = device
        # plain image HDUs of local files go to the GPU as the file stores them (COREG_RAW_FITS=0: decode on the host)
        self.raw_fits_upload = os.environ.get("COREG_RAW_FITS", "1") != "0"
        self.last_stats = None
        self.last_sharding = None       # how the last sweep was spread over the GPUs (parallel.lag_sharding)
        # set by the jitter-correction session (jitter_correction/jitter_correction.py):
        self.shard_lags = True          # False: every rank runs whole sweeps (images, not lags, are spread over GPUs)
        self._preloaded_small = None    # (data, header) already decoded by the session's prefetch thread
        self._handle_slot = 0           # which of the device's library contexts this sweep runs in
        # alignment.py:137-140
        for name in ("lag_crval1", "lag_crval2", "lag_crota", "lag_cdelt1", "lag_cdelt2"):
            if getattr(self, name) is None:
                setattr(self, name, np.array([0.0]))

    # ------------------------------------------------------------------------------------------------------------
    def _load(self):
        """Headers of both files and the pixels of the image to align.  The reference image's pixels are decoded only
        when `_large_pixels()` is called: a sweep whose prepared reference is still resident on the GPU never does."""
        self.data_large = None
        self.hdr_large = fits_io.Header(fits_io.read_header(self.large_fov_known_pointing, self.large_fov_window))
        if self._preloaded_small is not None:
            ds, hs = self._preloaded_small
        else:
            # a plain image HDU of a local file is not decoded on the host at all: its bytes are memory-mapped and go
            # to the GPU as the file stores them (fits_io.RawImage; byte swap, BSCALE / BZERO and the float conversion
            # of alignment.py:198 / :314 run there).  Anything else (in-memory pairs, compressed images, URLs with
            # astropy) is read and decoded as before.
            # A tile-compressed image (EUI level-1 / level-2 files) likewise: its COMPRESSED bytes go up and the GPU
            # decodes them (fits_io.CompressedImage; cfitsio's RICE_1 codec restated, csrc/ricecomp.hpp).
            ds, hs = (fits_io.load_for_upload if self.raw_fits_upload else fits_io.read_image)(
                self.small_fov_to_correct, self.small_fov_window)
        # float32 (BITPIX=-32) pixels stay float32: the float64 cast of alignment.py:198 / :314 is exact
        self.data_small = fits_io.native_pixels(ds)
        self.hdr_small = fits_io.Header(hs)
        hdrutil.check_and_create_pcij_matrix(self.hdr_small, self.force_crota_0)  # alignment.py:232 / :310
        hdrutil.check_and_create_pcij_matrix(self.hdr_large, self.force_crota_0)

    def _large_pixels(self):
        if self.data_large is None:
            dl = (fits_io.load_for_upload if self.raw_fits_upload else fits_io.read_image)(
                self.large_fov_known_pointing, self.large_fov_window)[0]
            # alignment.py:191 / :301 cast to float64; float32 pixels (BITPIX=-32) are kept as they are -- the cast is
            # exact and the library does it on the GPU, half the bytes cross PCIe; a memory-mapped data unit (RawImage)
            # goes up as stored, and only the rectangle the target grid can touch
            self.data_large = fits_io.native_pixels(dl)
        return self.data_large

    def _reference_tag(self, frame, *what):
        """Identity of a prepared reference: (file identity of the reference image, preparation parameters), or None
        when the reference image is an in-memory array."""
        ident = fits_io.file_identity(self.large_fov_known_pointing, self.large_fov_window)
        if ident is None:
            return None

        def flat(v):
            return tuple(np.asarray(v, dtype=np.float64).ravel().tolist())
        return (ident, frame, bool(self.force_crota_0), int(self.order)) + tuple(flat(w) for w in what)

    def _set_initial_header_values(self, ang2pipi=True):
        """alignment.py:799-842."""
        h = self.hdr_small
        self.crval1_ref, self.crval2_ref = h["CRVAL1"], h["CRVAL2"]
        if "CROTA" in h:
            self.crota_ref = h["CROTA"]
        elif "CROTA2" in h:
            self.crota_ref = h["CROTA2"]
        else:
            s = -np.sign(h["PC1_2"]) + (h["PC1_2"] == 0)
            self.crota_ref = np.rad2deg(np.arccos(h["PC1_1"])) * s
            h["CROTA"] = np.rad2deg(np.arccos(h["PC1_1"]))
        self.cdelt1_ref, self.cdelt2_ref = h["CDELT1"], h["CDELT2"]
        self.unit1, self.unit2 = str(h["CUNIT1"]).strip(), str(h["CUNIT2"]).strip()
        if self.unit_lag in self.unit1:
            pass
        else:
            warnings.warn("Units of headers in deg: Modyfying inputs units to deg.")
            for name, unit in (("lag_crval1", self.unit1), ("lag_crval2", self.unit2), ("lag_cdelt1", self.unit1),
                               ("lag_cdelt2", self.unit2)):
                v = np.asarray(getattr(self, name), dtype=np.float64)
                if ang2pipi:
                    v = hdrutil.ang2pipi(v, self.unit_lag)
                setattr(self, name, hdrutil.convert(v, self.unit_lag, unit))
            self.unit_lag = self.unit1
        if self.unit1 != self.unit2:
            raise ValueError("CUNIT1 and CUNIT2 must be equal")
        if self.lag_solar_r is None:
            self.lag_solar_r = np.array([1.004])

    # ------------------------------------------------------------------------------------------------------------
    def align_using_carrington(self, lonlims=None, latlims=None, size_deg_carrington=None, shape=None,
                               reference_date=None, method="correlation", method_carrington_reprojection="fa",
                               return_type="AlignmentResults"):
        """alignment.py:144-261."""
        self.method = method
        self.coordinate_frame = "final_carrington"
        if method_carrington_reprojection == "sunpy":
            raise NotImplementedError("method_carrington_reprojection='sunpy' delegates to sunpy.reproject_to in the "
                                      "reference (alignment.py:939-985) and is out of scope of the GPU path")
        if method_carrington_reprojection != "fa":
            raise ValueError("method_carrington_reprojection must be either 'fa' or 'sunpy")
        self._load()
        if reference_date is None:
            if "DATE-AVG" not in self.hdr_large:
                raise ValueError("Either provide a reference date manualy or the reference file header must have a "
                                 "DATE-AVG keyword.")
            self.reference_date = self.hdr_large["DATE-AVG"]
        else:
            self.reference_date = reference_date  # no numerical effect on the 'fa' path (quirk Q5)
        if (lonlims is None) and (latlims is None) and (size_deg_carrington is not None):
            crln, crlt = self.hdr_small["CRLN_OBS"], self.hdr_small["CRLT_OBS"]
            self.lonlims = [crln - 0.5 * size_deg_carrington[0], crln + 0.5 * size_deg_carrington[0]]
            self.latlims = [crlt - 0.5 * size_deg_carrington[1], crlt + 0.5 * size_deg_carrington[1]]
            self.shape = [self.hdr_small["NAXIS1"], self.hdr_small["NAXIS2"]]
        elif (lonlims is not None) and (latlims is not None) and (shape is not None):
            self.lonlims, self.latlims, self.shape = lonlims, latlims, shape
        else:
            raise ValueError("either set lonlims as None, or not. no in between.")
        if self.shape[0] * self.shape[1] > 25000000:
            warnings.warn(f"shape parameter is {shape=}, which is very large.Computational time might significantly "
                          "increase")
        results = self._find_best_header_parameters()
        return self._wrap(results, return_type, restore_units=True)

    def align_using_helioprojective(self, method="correlation", return_type="AlignmentResults", fov_limits=None,
                                    remove_fov_limits=None):
        """alignment.py:263-342."""
        self.lonlims = self.latlims = self.shape = self.reference_date = None
        self.method = method
        self.coordinate_frame = "final_helioprojective"
        self._load()
        results = self._find_best_header_parameters(fov_limits=fov_limits, remove_fov_limits=remove_fov_limits)
        return self._wrap(results, return_type, restore_units=True)

    def align_using_initial_carrington(self, method="correlation", return_type="AlignmentResults"):
        """alignment.py:344-399: both inputs are already Carrington maps (CRLN-CAR / CRLT-CAR, plate carree).  The image
        reference map is resampled once on the pixel grid of the map to align (the sub-map of alignment.py:987-1016, in
        both the parallel and the serial branch), the map to align per lag on that same grid; no Carrington transform;
        both images are read as float32, lags are not wrapped (ang2pipi=False)."""
        self.lonlims = self.latlims = self.shape = self.reference_date = None
        self.method = method
        self.coordinate_frame = "initial_carrington"
        self._load()
        for hdr, what in ((self.hdr_small, "image to align"), (self.hdr_large, "reference image")):
            if not str(hdr.get("CTYPE1", "")).strip().upper().endswith("-CAR"):
                raise ValueError(f"align_using_initial_carrington: the {what} is not a CRLN-CAR / CRLT-CAR map")
        self.data_small = np.array(self.data_small, dtype=np.float32)  # alignment.py:384
        results = self._find_best_header_parameters(ang2pipi=False)
        if return_type == "corr":
            return results
        return AlignmentResults(corr=results, lag_crval1=self.lag_crval1, lag_crval2=self.lag_crval2,
                                lag_cdelt1=self.lag_cdelt1, lag_cdelt2=self.lag_cdelt2, lag_crota=self.lag_crota,
                                unit_lag=self.unit_lag, image_to_align_path=self.small_fov_to_correct,
                                image_to_align_window=self.small_fov_window,
                                reference_image_path=self.large_fov_known_pointing,
                                reference_image_window=self.large_fov_window)

    def _wrap(self, results, return_type, restore_units):
        if return_type == "corr":
            return results
        if return_type != "AlignmentResults":
            return results
        if restore_units:  # alignment.py:242-250 / :325-333
            for name in ("lag_crval1", "lag_crval2", "lag_cdelt1", "lag_cdelt2"):
                v = hdrutil.ang2pipi(getattr(self, name), self.unit_lag)
                setattr(self, name, hdrutil.convert(v, self.unit_lag, self.unit_lag_input))
            self.unit_lag = self.unit_lag_input
        return AlignmentResults(corr=results, lag_crval1=self.lag_crval1, lag_crval2=self.lag_crval2,
                                lag_cdelt1=self.lag_cdelt1, lag_cdelt2=self.lag_cdelt2, lag_crota=self.lag_crota,
                                unit_lag=self.unit_lag_input, image_to_align_path=self.small_fov_to_correct,
                                image_to_align_window=self.small_fov_window,
                                reference_image_path=self.large_fov_known_pointing,
                                reference_image_window=self.large_fov_window)

    # ------------------------------------------------------------------------------------------------------------
    def _set_remove_fov_limits_to_nan(self, remove_fov_limits):
        """alignment.py:863-874: pixels of the small image inside the lon/lat box -> NaN.  Limits: astropy Quantities
        or plain numbers in the input lag unit."""
        lon, lat = wcs_tan.pixel_lonlat(self.hdr_small)
        lonl = wcs_tan._lims_deg(remove_fov_limits[0], self.unit_lag_input)
        latl = wcs_tan._lims_deg(remove_fov_limits[1], self.unit_lag_input)
        inside = (lon >= lonl[0]) & (lon <= lonl[1]) & (lat >= latl[0]) & (lat <= latl[1])
        self.data_small[inside] = np.nan

    def _select_fov_in_small_data(self, fov_limits, handle):
        """alignment.py:1082-1127: re-grid the small image on a regular lon/lat grid restricted to `fov_limits`
        (new header: PC = identity, CROTA = 0).  Literal restatement, including the reference's use of the ROW count
        for CRPIX1 / NAXIS1 (it mixes the two axes, harmless for square selections).  The resample runs on the GPU."""
        h = self.hdr_small
        lon, lat = wcs_tan.pixel_lonlat(h)
        lonl = wcs_tan._lims_deg(fov_limits[0], self.unit_lag_input)
        latl = wcs_tan._lims_deg(fov_limits[1], self.unit_lag_input)
        long, latg, dlon, dlat = wcs_tan.build_regular_grid(lon, lat, lonl, latl)
        if long.size == 0:
            raise ValueError("fov_limits select no pixel of the small image")
        mid = [long.shape[0] // 2, long.shape[1] // 2]
        hg = h.copy()
        u1, u2 = hdrutil.unit_to_deg(h["CUNIT1"]), hdrutil.unit_to_deg(h["CUNIT2"])
        hg["CRVAL1"] = float(long[mid[0], mid[1]]) / u1
        hg["CRVAL2"] = float(latg[mid[0], mid[1]]) / u2
        hg["CRPIX1"] = mid[0] + 1
        hg["CRPIX2"] = mid[1] + 1
        hg["CDELT1"] = float(dlon) / u1
        hg["CDELT2"] = float(dlat) / u2
        hg["PC1_1"], hg["PC2_2"], hg["PC1_2"], hg["PC2_1"] = 1.0, 1.0, 0.0, 0.0
        hg["CROTA"] = 0.0
        hg["CROTA2"] = 0.0
        hg["NAXIS1"] = long.shape[0]
        hg["NAXIS2"] = long.shape[1]
        hg.pop("ZNAXIS1", None)
        hg.pop("ZNAXIS2", None)
        handle.set_small(self.data_small)
        self.data_small = handle.resample_helioprojective(hg, h, order=self.order, dtype=np.float64)
        self.hdr_small = hg

    def _find_best_header_parameters(self, ang2pipi=True, fov_limits=None, remove_fov_limits=None):
        """alignment.py:613-797 on the GPU.  Returns float64 [n_crval1, n_crval2, n_cdelt1, n_cdelt2, n_crota,
        n_solar_r]; lag-points the library could not evaluate are NaN, never 0 (quirk Q9)."""
        if self.method == "correlation":
            method = _lib.METHOD_CORRELATION
        elif self.method == "residus":
            method = _lib.METHOD_RESIDUS  # alignment.py:544-547: no NaN mask, NaN unless every grid pixel overlaps
        else:
            raise NotImplementedError  # alignment.py:549
        device = self.device
        rank, world = parallel.world_info()
        if not self.shard_lags:
            rank, world = 0, 1
        if device is None:
            device = -1
            if parallel.world_info()[1] > 1:
                import torch
                device = torch.cuda.current_device()
        # A plain script (no torch.distributed) that asks for `parallelism=True` gets what the reference gives it: the whole
        # machine (alignment.py:692-744 fans out over counts_cpu_max processes).  Here: every visible GPU, driven from
        # this one process by the library itself (coreg_multi: a host thread + context per device, lag-plane blocks, one
        # RCCL all-gather).  An explicit `device=`, a jitter session spreading images over GPUs, one visible GPU or
        # COREG_SINGLE_DEVICE=1 keep the single-device context.
        use_all = (self.parallelism and world == 1 and self.device is None and self.shard_lags
                   and parallel.world_info()[1] == 1 and os.environ.get("COREG_SINGLE_DEVICE", "0") != "1"
                   and _lib.device_count() > 1)
        if use_all:
            h = _lib.shared_multi_handle()
        else:
            h = _lib.shared_handle(device, self._handle_slot)  # long-lived: buffers are re-used by the next Alignment
        # with several ranks over RCCL each image crosses PCIe once in all -- 1/N per rank -- and is assembled on every
        # GPU by an all-gather over xGMI (parallel.replicate_image); one rank / gloo: the whole image from this host
        spread = world > 1

        def upload_small(data):
            if spread and isinstance(data, (fits_io.RawImage, fits_io.CompressedImage)):
                data = fits_io.native_pixels(data.decode())  # row shares + all-gather work on decoded pixels
            t = parallel.replicate_image(data) if spread and np.asarray(data).dtype in (np.float32, np.float64) else None
            if t is None:
                h.set_small(data)
            else:
                import torch
                torch.cuda.current_stream().synchronize()  # the all-gather ran on torch's stream, not the handle's
                h.set_small_from_device(t.data_ptr(), t.shape, np.float32 if t.element_size() == 4 else np.float64)
                h.synchronize()  # `t` may go once the copy has run

        # alignment.py:844-861: thresholds, then the box to remove, then the sub-FOV re-grid
        on_device = (remove_fov_limits is None) and (fov_limits is None)
        n_finite = None
        if on_device:
            # thresholds applied to the resident copy (self.data_small is left as loaded).  The upload runs on the handle's
            # upload stream; the threshold pass -- the first reader of the pixels -- is issued after the reference
            # preparation (`finite_pixels`, below), which therefore does not wait for the image's DMA.
            # COREG_ASYNC_UPLOAD=1 also hands the staging copies to the library's upload thread (option "async_upload";
            # measured +-0.03 ms on the headline call -- the DMA already overlaps the preparation -- hence off by default).
            # self.data_small is not modified until the sweep has returned.
            # (only when a reference preparation follows -- nothing to overlap with when the prepared reference is still
            # resident, and the hand-over to the thread then costs more than it saves: profiles/r05_api_timing*.json)
            will_prepare = True
            if self.coordinate_frame == "final_carrington":
                sr0 = 1.004 if self.lag_solar_r is None else float(np.atleast_1d(self.lag_solar_r)[0])
                tag0 = self._reference_tag("carrington", self.lonlims, self.latlims, self.shape, sr0)
                will_prepare = tag0 is None or tag0 != getattr(h, "reference_tag", None)
            use_async = (will_prepare and not use_all and not spread
                         and os.environ.get("COREG_ASYNC_UPLOAD", "0") == "1")
            if use_async:
                h.set_option("async_upload", 1)
            try:
                upload_small(self.data_small)
            finally:
                if use_async:
                    h.set_option("async_upload", 0)
        else:
            self.data_small = np.array(self.data_small, dtype=np.float64)
            hdrutil.set_threshold_minmax_to_nan(self.data_small, self.small_fov_value_min, self.small_fov_value_max)
            if remove_fov_limits is not None:
                self._set_remove_fov_limits_to_nan(remove_fov_limits)
            if fov_limits is not None:
                self._select_fov_in_small_data(fov_limits, h)
            n_finite = int(np.isfinite(self.data_small).sum())
            upload_small(self.data_small)
        self._set_initial_header_values(ang2pipi)

        def finite_pixels():
            """alignment.py:654-656, after the first reference preparation (the check itself reads the uploaded pixels)."""
            nonlocal n_finite
            if n_finite is None:
                n_finite = h.threshold_small(self.small_fov_value_min, self.small_fov_value_max)
            if n_finite == 0:
                raise ValueError("minimum or maximum value have set all small FOV to nan")

        if n_finite is not None:
            finite_pixels()
        if self.unit_lag != self.hdr_small["CUNIT1"]:
            raise ValueError("lag.unit and cUNIT are not the same")  # alignment.py:406

        lags = _lib.LagSet(self.lag_crval1, self.lag_crval2, self.lag_cdelt1, self.lag_cdelt2, self.lag_crota)
        sem = _lib.CDELT_INTENDED if self.cdelt_semantics == "intended" else _lib.CDELT_REFERENCE
        solar_rs = np.atleast_1d(np.asarray(self.lag_solar_r, dtype=np.float64))
        # Spreading the lag set over the GPUs (parallel.lag_sharding): blocks of the (CRVAL1, CRVAL2) plane + one
        # all-gather -- the partition bench.py measures; contiguous slices of the raveled index (the reference's literal
        # np.array_split, alignment.py:677-687) when the plane is smaller than the number of GPUs; shares of the GRID +
        # one all-reduce of the six sums per lag when there are few lag-points per GPU.
        # (the helioprojective sweep runs ONE launch whatever the lag set; the others one per (cdelt, crota) combination)
        per_combo = self.coordinate_frame != "final_helioprojective"
        mode = parallel.lag_sharding(lags.shape, world, per_combo)
        self.last_sharding = mode
        my_lags, lo, hi = lags, 0, lags.size
        combos = None  # (c_lo, c_hi): this rank's run of (cdelt1, cdelt2, crota) combinations, None = all of them
        if mode in ("blocks", "combos"):
            lo1, hi1, lo2, hi2, c_lo, c_hi = parallel.grid_share(lags.shape, world, rank, per_combo)
            a = lags.arrays
            my_lags = _lib.LagSet(a[0][lo1:hi1], a[1][lo2:hi2], a[2], a[3], a[4])
            inner = lags.shape[2] * lags.shape[3] * lags.shape[4]
            if (c_lo, c_hi) != (0, inner):
                combos = (c_lo, c_hi)
            hi = my_lags.shape[0] * my_lags.shape[1] * (c_hi - c_lo)
        elif mode == "slices":
            lo, hi, _ = parallel.shard_bounds(lags.size, world, rank)

        def select_combos():
            if combos is not None:  # one-shot options, consumed by the sweep call that follows
                h.set_option("combo_begin", combos[0])
                h.set_option("combo_end", combos[1])

        def prepare(kind, *args):
            """Once-only reference preparation.  The library uploads only the rectangle of the reference image the target
            grid can touch (usually a few hundred KB), so every rank simply sends its own crop: nothing to replicate."""
            getattr(h, "prepare_reference_" + kind)(self._large_pixels(), *args)

        out = np.full(lags.shape + (len(solar_rs),), np.nan)
        for kk, solar_r in enumerate(solar_rs):
            if self.coordinate_frame == "final_carrington":
                grid = _lib.Grid(self.lonlims, self.latlims, self.shape, numpy_lat_trig=True)
                tag = self._reference_tag("carrington", self.lonlims, self.latlims, self.shape, solar_r)
                if tag is None or tag != h.reference_tag:
                    prepare("carrington", self.hdr_large, grid, solar_r, self.order)
                    h.reference_tag = tag

                def run(g=grid, sr=solar_r):
                    select_combos()
                    return h.sweep_carrington(self.hdr_small, g, sr, my_lags, order=self.order, method=method,
                                              cdelt_semantics=sem, lag_begin=lo, lag_end=hi)
            elif self.coordinate_frame == "initial_carrington":
                # BOTH branches of the reference name this frame in their sub-map condition (alignment.py:649-651 parallel,
                # :765-767 serial): the reference map is resampled once on the grid of the map to align, float32, and
                # every lag re-samples the map to align on its own grid (round 5: found by the reference-run fixtures;
                # rounds 1-4 correlated on the reference map's full grid)
                # the reference casts the reference map to float32 BEFORE the sub-map interpolation in this frame
                # (alignment.py:371-372); exact and free for BITPIX=-32 pixels, a rounding for a float64 map
                big = self._large_pixels()
                if getattr(big, "dtype", None) != np.float32:  # (also a memory-mapped data unit that is not BITPIX=-32)
                    big = np.asarray(big).astype(np.float32)
                h.prepare_reference_helioprojective(big, self.hdr_large, self.hdr_small, self.order)

                def run():
                    select_combos()
                    return h.sweep_helioprojective(self.hdr_small, self.hdr_small, my_lags, order=self.order,
                                                   method=method, cdelt_semantics=sem, lag_begin=lo, lag_end=hi)
            else:
                if self.parallelism:
                    prepare("helioprojective", self.hdr_large, self.hdr_small, self.order)
                    target = self.hdr_small
                else:
                    h.set_reference_on_grid(np.asarray(self._large_pixels(), dtype=np.float64))  # quirk Q1: float64
                    target = self.hdr_large

                def run(t=target):
                    select_combos()
                    return h.sweep_helioprojective(t, self.hdr_small, my_lags, order=self.order, method=method,
                                                   cdelt_semantics=sem, lag_begin=lo, lag_end=hi)
            finite_pixels()
            if mode == "points":
                part = parallel.point_sharded_sweep(h, run, lags.size)
            elif mode in ("blocks", "combos"):
                part = parallel.allgather_lag_blocks(run(), lags.shape, per_combo_launch=per_combo)
            elif mode == "slices":
                part = parallel.allgather_lag_slices(run(), lags.size).cpu().numpy()
            else:
                part = run()
            out[..., kk] = np.asarray(part).reshape(lags.shape)
        self.last_stats = h.last_stats()
        if hasattr(h, "drop_small_keepalive"):
            h.drop_small_keepalive()  # (every sweep has returned: the upload thread is done with the pixels)
        if use_all:
            self.last_sharding = h.last_mode
        elif self.coordinate_frame != "final_carrington" and hasattr(h, "last_tap_fix"):
            #  samples whose coordinate comes back within 1e-8 px of an integer (odd spline orders) or of a
            # bound of the image (even orders) are re-evaluated with wcslib's own arithmetic; a list beyond "tap_cap"
            # entries is not applied -- say so
            if h.last_tap_fix()["overflow"]:
                warnings.warn("more noise-decided samples than the library's 'tap_cap': isolated lag-points of this "
                              "sweep may differ from the reference by up to 1e-5 (raise it with set_option('tap_cap', n))")
        return out
