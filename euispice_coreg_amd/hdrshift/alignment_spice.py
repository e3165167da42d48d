"""
`AlignmentSpice` -- drop-in for euispice_coreg.hdrshift.AlignmentSpice (hdrshift/alignment_spice.py:13-355): aligns a
SPICE L2 raster (4-D cube [time, wavelength, y, x]) or an L3 coefficient map with an imager / synthetic raster.  The
data preparation either side of the sweep (cube collapse over a wavelength interval, slit-edge masking, 4-D -> 2-D
header) is restated here without astropy; the sweep itself is the GPU path of `Alignment`.

Differences from the reference, all deliberate:
  * `sub_fov_window` compares longitudes wrapped to (-180, 180] deg (the reference compares wcslib's raw output, whose
    branch follows the sign of CRVAL1, against the user's limits);
  * the sunpy branch (alignment_spice.py:300-303) is not implemented;
  * inputs may be FITS paths or (data, header) pairs (then pass `level=`).
"""
from __future__ import annotations

import numpy as np

from .. import _lib
from ..utils import fits_io, header as hdrutil, spice_header, wcs_tan
from .alignment import Alignment


def _angstrom(v):
    return float(v.to("angstrom").value) if hasattr(v, "to") else float(v)


class AlignmentSpice(Alignment):

    def __init__(self, large_fov_known_pointing, small_fov_to_correct, lag_crval1=None, lag_crval2=None,
                 lag_cdelt1=None, lag_cdelt2=None, lag_crota=None, lag_solar_r=None, large_fov_window=-1,
                 small_fov_window=-1, parallelism=False, counts_cpu_max=40, display_progress_bar=False,
                 path_save_figure=None, wavelength_interval_to_sum="all", sub_fov_window="all", level=None,
                 cdelt_semantics="intended", device=None):
        super().__init__(large_fov_known_pointing=large_fov_known_pointing, small_fov_to_correct=small_fov_to_correct,
                         lag_crval1=lag_crval1, lag_crval2=lag_crval2, lag_cdelt1=lag_cdelt1, lag_cdelt2=lag_cdelt2,
                         lag_crota=lag_crota, display_progress_bar=display_progress_bar, lag_solar_r=lag_solar_r,
                         parallelism=parallelism, counts_cpu_max=counts_cpu_max, large_fov_window=large_fov_window,
                         small_fov_window=small_fov_window, path_save_figure=path_save_figure,
                         cdelt_semantics=cdelt_semantics, device=device)
        self.sub_fov_window = sub_fov_window
        self.extend_pixel_size = None
        self.cut_from_center = None
        self.wavelength_interval_to_sum = wavelength_interval_to_sum
        self.level = level

    # ------------------------------------------------------------------------------------------------------------
    def _level(self):
        """alignment_spice.py:94-98: from the file name, unless given."""
        if self.level is not None:
            return self.level
        name = self.small_fov_to_correct if isinstance(self.small_fov_to_correct, str) else ""
        if "L2" in name:
            return 2
        if "L3" in name:
            return 3
        return None

    def align_using_helioprojective(self, method="correlation", extend_pixel_size=False, cut_from_center=None,
                                    return_type="AlignmentResults", coefficient_l3=None):
        """alignment_spice.py:66-120."""
        self.lonlims = self.latlims = self.shape = self.reference_date = None
        self.method = method
        self.coordinate_frame = "final_helioprojective"
        self.extend_pixel_size = extend_pixel_size
        self.cut_from_center = cut_from_center
        self._extract_imager_data_header()
        self._extract_spice_data_header(level=self._level(), coeff=coefficient_l3)
        results = self._find_best_header_parameters()
        return self._wrap(results, return_type, restore_units=True)

    def align_using_carrington(self, lonlims, latlims, size_deg_carrington=None, shape=None, reference_date=None,
                               method="correlation", return_type="AlignmentResults", coefficient_l3=None):
        """alignment_spice.py:122-180.  (The reference reads hdr_small before it is loaded when only
        `size_deg_carrington` is given; here the SPICE header is loaded first.)"""
        self.reference_date = reference_date
        self.extend_pixel_size = False
        self.method = method
        self.coordinate_frame = "final_carrington"
        self._extract_imager_data_header()
        self._extract_spice_data_header(level=self._level(), coeff=coefficient_l3)
        if (lonlims is None) and (latlims is None) and (size_deg_carrington is not None):
            crln, crlt = self.hdr_small["CRLN_OBS"], self.hdr_small["CRLT_OBS"]
            self.lonlims = [crln - 0.5 * size_deg_carrington[0], crln + 0.5 * size_deg_carrington[0]]
            self.latlims = [crlt - 0.5 * size_deg_carrington[1], crlt + 0.5 * size_deg_carrington[1]]
            self.shape = [self.hdr_small["NAXIS1"], self.hdr_small["NAXIS2"]]
        elif (lonlims is not None) and (latlims is not None) and (shape is not None):
            self.lonlims, self.latlims, self.shape = lonlims, latlims, shape
        else:
            raise ValueError("either set lonlims as None, or not. no in between.")
        h = self.hdr_small  # alignment_spice.py:159-168: the Carrington transform works in arcsec
        for k in ("CRVAL1", "CRVAL2"):
            h[k] = float(hdrutil.convert(hdrutil.ang2pipi(h[k], h["CUNIT" + k[-1]]), h["CUNIT" + k[-1]], "arcsec"))
        for k in ("CDELT1", "CDELT2"):
            h[k] = float(hdrutil.convert(h[k], h["CUNIT" + k[-1]], "arcsec"))
        h["CUNIT1"] = h["CUNIT2"] = "arcsec"
        results = self._find_best_header_parameters()
        return self._wrap(results, return_type, restore_units=False)

    # ------------------------------------------------------------------------------------------------------------
    def _extract_imager_data_header(self):
        """alignment_spice.py:182-187."""
        # (the pixels are read when the reference is prepared -- `Alignment._large_pixels`: as stored, cropped to what the
        # target grid can touch, decoded on the GPU; not at all when the prepared reference is still resident)
        self.data_large = None
        self.hdr_large = fits_io.Header(fits_io.read_header(self.large_fov_known_pointing, self.large_fov_window))
        hdrutil.check_and_create_pcij_matrix(self.hdr_large, self.force_crota_0, warn=False)

    def _extract_spice_data_header(self, level, coeff=None):
        """alignment_spice.py:189-221."""
        # a big-endian view of the memory-mapped window: only the wavelength planes the sum needs are ever touched
        cube, hdr = fits_io.open_cube(self.small_fov_to_correct, self.small_fov_window)
        hdr = fits_io.Header(hdr)
        dt = hdr["PC4_1"]
        if level == 2:
            self._prepare_spice_from_l2(cube, hdr)
        elif level == 3:
            self._prepare_spice_from_l3(cube, hdr, coeff)
        else:
            raise ValueError("level must be 2 or 3")
        for k in ("SOLAR_B0", "RSUN_REF", "DSUN_OBS", "CROTA"):
            self.hdr_small[k] = hdr[k]
        for k in ("CRLN_OBS", "CRLT_OBS"):  # needed by the Carrington transform (rectify.py:387-415)
            if k in hdr and k not in self.hdr_small:
                self.hdr_small[k] = hdr[k]
        if self.extend_pixel_size:
            self._correct_solar_rotation(dt)
        hdrutil.check_and_create_pcij_matrix(self.hdr_small, self.force_crota_0, warn=False)

    def _correct_solar_rotation(self, dt):
        """alignment_spice.py:223-248: shrink CDELT1 by the apparent solar rotation during one raster step."""
        h = self.hdr_small
        B0 = np.deg2rad(h["SOLAR_B0"])
        band = self.hdr_large["WAVELNTH"]
        omega_car = np.deg2rad(360 / 25.38 / 86400)
        if band == 174:
            band = 171
        omega = omega_car + spice_header.diff_rot(B0, f"EIT {band}")
        Rsun, Dsun = h["RSUN_REF"], h["DSUN_OBS"]
        phi_rot = np.rad2deg(1.004 * omega * Rsun / (Dsun - 1.004 * Rsun)) * 3600  # arcsec / s
        alpha = np.deg2rad(h["CRVAL1"] * hdrutil.unit_to_deg(h["CUNIT1"]))
        phi = np.arcsin(((Dsun - 1.004 * Rsun) / (1.004 * Rsun)) * np.sin(alpha))
        dtx_old = float(hdrutil.convert(h["CDELT1"], h["CUNIT1"], "arcsec"))
        dtx_new = dtx_old - dt * phi_rot * np.cos(phi)
        h["CDELT1"] = float(hdrutil.convert(dtx_new, "arcsec", h["CUNIT1"]))

    def _prepare_spice_from_l2(self, cube, hdr):
        """alignment_spice.py:250-323."""
        cube = np.asarray(cube)
        if cube.ndim != 4:
            raise ValueError("a SPICE L2 window is a 4-D cube [time, wavelength, y, x]")
        ymin, ymax = spice_header.vertical_edges_limits(hdr)
        self.hdr_small = spice_header.celestial_header(hdr)
        # np.nansum(float64(cube)[0, sel], axis=0) of the reference, plane by plane in the same order (a reduction over
        # the outer axis is a sequential accumulation, so the sums are the same to the bit) without a float64 copy of the
        # whole cube; the rows outside [ymin, ymax) are NaN in the result either way
        if isinstance(self.wavelength_interval_to_sum, str) and self.wavelength_interval_to_sum == "all":
            sel = np.arange(cube.shape[1])
        elif isinstance(self.wavelength_interval_to_sum, (list, tuple)):
            wave = spice_header.wavelengths_angstrom(hdr)
            lo, hi = (_angstrom(v) for v in self.wavelength_interval_to_sum)
            sel = np.flatnonzero(np.logical_and(wave >= lo, wave <= hi))
        else:
            raise ValueError("wavelength_interval_to_sum must be a [wave_min * u.angstrom, wave_max * u.angstrom] "
                             "or 'all' str ")
        # big-endian float planes straight from the memory-mapped data unit: the library's threaded host routine (same
        # additions in the same order); anything else (native arrays handed over in memory, integer cubes): NumPy
        self.data_small = _lib.nansum_planes_be(cube[0], sel)
        if self.data_small is None:
            self.data_small = np.zeros(cube.shape[2:], dtype=np.float64) if len(sel) == 0 else None
            for plane in cube[0][sel]:
                # NaN -> 0 on the plane in its own precision (native byte order), the float64 cast inside the addition:
                # same values, same order of additions as np.nansum(float64(cube)), in two light passes per plane
                if plane.dtype.kind == "f":
                    q = plane.astype(plane.dtype.newbyteorder("="))
                    np.copyto(q, 0, where=np.isnan(q))
                else:
                    q = plane
                if self.data_small is None:
                    self.data_small = q.astype(np.float64)
                else:
                    np.add(self.data_small, q, out=self.data_small)
        self.data_small[:ymin, :] = np.nan
        self.data_small[ymax:, :] = np.nan
        if self.cut_from_center is not None:
            xlen = self.cut_from_center
            xmid = self.data_small.shape[1] // 2
            self.data_small[:, :(xmid - xlen // 2 - 1)] = np.nan
            self.data_small[:, (xmid + xlen // 2):] = np.nan
        self.hdr_small["NAXIS1"] = self.data_small.shape[1]
        self.hdr_small["NAXIS2"] = self.data_small.shape[0]
        if isinstance(self.sub_fov_window, str) and self.sub_fov_window == "all":
            pass
        elif isinstance(self.sub_fov_window, (list, tuple)):
            lon, lat = wcs_tan.pixel_lonlat(self.hdr_small)
            lon = hdrutil.ang2pipi(lon)
            lonl = wcs_tan._lims_deg(self.sub_fov_window[0:2], "arcsec")
            latl = wcs_tan._lims_deg(self.sub_fov_window[2:4], "arcsec")
            sel = (lon >= lonl[0]) & (lon <= lonl[1]) & (lat >= latl[0]) & (lat <= latl[1])
            self.data_small[~sel] = np.nan
        else:
            raise ValueError("sub_fov_window must be a [lon_min * u.arcsec, lon_max * u.arcsec,"
                             " lat_min * u.arcsec, lat_max * u.arcsec] or 'all' str ")

    def _prepare_spice_from_l3(self, cube, hdr, coeff):
        """alignment_spice.py:340-355.  (NAXIS1/2 are set here; the reference leaves them out of the 2-D header.)"""
        self.data_small = np.array(np.asarray(cube)[coeff, ...], dtype=np.float64)  # (only the plane asked for)
        ymin, ymax = spice_header.vertical_edges_limits(hdr)
        self.data_small[:ymin, :] = np.nan
        self.data_small[ymax:, :] = np.nan
        self.hdr_small = spice_header.celestial_header(hdr)
        self.hdr_small["NAXIS1"] = self.data_small.shape[-1]
        self.hdr_small["NAXIS2"] = self.data_small.shape[-2]
