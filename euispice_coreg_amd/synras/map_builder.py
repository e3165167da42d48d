"""
Synthetic-raster builder -- drop-in for euispice_coreg.synras.map_builder (synras/map_builder.py:15-349): for every
raster step of a SPICE window, take the imager frame closest in time and sample it (un-prefiltered quadratic B-spline,
`interpol2d(order=2, fill=NaN)`, map_builder.py:95-131) along the slit; the columns make an image with the SPICE
geometry that `AlignmentSpice` can use as its reference.

MI355X form: raster steps that share an imager frame are resampled in ONE launch of the library's gather kernel over
the whole SPICE pixel grid (TAN -> sky -> TAN is an exact homography) and their columns are copied
out; each imager frame is decoded and uploaded once.  Without astropy: SPICE header arithmetic in
utils/spice_header.py (pinned against astropy / wcslib golden vectors).  Level-2 windows (x, y, wavelength, time) and
level-3 coefficient maps (coefficient, x, y, time; map_builder.py:290-346) are both handled by locating the HPLN / HPLT /
time axes by CTYPE.  The sunpy branch is not implemented.
"""
from __future__ import annotations

import datetime as _dt
import os
import random
import warnings

import numpy as np

from .. import _lib
from ..utils import fits_io, header as hdrutil, spice_header, wcs_tan


def _seconds(q):
    return float(q.to("s").value) if hasattr(q, "to") else float(q)


class MapBuilder:

    def __init__(self):
        pass

    def process(self, path_output: str):
        pass


class ComposedMapBuilder(MapBuilder):

    def __init__(self, path_to_spectro, list_imager_paths, threshold_time, window_imager=-1, window_spectro=0,
                 device=None):
        """map_builder.py:26-56.  `threshold_time`: astropy Quantity or seconds."""
        super().__init__()
        self.path_to_spectro = path_to_spectro
        self.list_imager_paths = list(list_imager_paths)
        self.window_imager = window_imager
        self.window_spectro = window_spectro
        self.threshold_time = threshold_time
        self.path_composed_map = None
        self.path_output = None
        self.device = device
        self.data_composed = None
        self.hdr_composed = None
        self.dates_selected = None
        self.hdr_spice_ = None
        self._extract_imager_metadata()

    # ------------------------------------------------------------------------------------------------------------
    def process(self, folder_path_output=None, basename_output=None, print_filename=True, level=2,
                keep_original_imager_pixel_size=False, return_synras_name=False):
        """map_builder.py:58-81."""
        self.path_output = folder_path_output
        hdr_spice = fits_io.read_header(self.path_to_spectro, self.window_spectro)
        name = self._create_map_from_hdu(hdr_spice, basename_output, folder_path_output, print_filename=print_filename,
                                         level=level, keep_original_imager_pixel_size=keep_original_imager_pixel_size)
        if return_synras_name:
            return name

    def process_from_header(self, hdr_spice, path_output=None, basename_output=None, print_filename=False, level=2,
                            keep_original_imager_pixel_size=False):
        """map_builder.py:83-87."""
        self.path_output = path_output
        self._create_map_from_hdu(fits_io.Header(hdr_spice), basename_output, path_output,
                                  print_filename=print_filename, level=level,
                                  keep_original_imager_pixel_size=keep_original_imager_pixel_size)

    def get_path_to_composed_map(self):
        return self.path_composed_map

    # ------------------------------------------------------------------------------------------------------------
    def _extract_imager_metadata(self):
        """map_builder.py:220-227."""
        self.headers = [fits_io.read_header(p, self.window_imager) for p in self.list_imager_paths]
        self.dates = [spice_header.parse_date(h["DATE-AVG"]) for h in self.headers]

    def _find_closest_imager_time(self, utc_ref):
        """map_builder.py:229-231."""
        dt = np.array([abs((utc_ref - n).total_seconds()) for n in self.dates], dtype=np.float64)
        return int(dt.argmin()), float(dt.min())

    @staticmethod
    def _return_mean_time(utc_list):
        """map_builder.py:233-239."""
        ref = utc_list[0]
        delta = np.array([(ref - n).total_seconds() for n in utc_list], dtype=np.float64)
        return ref - _dt.timedelta(seconds=float(delta.mean())), delta

    def _prepare_spectro_data(self, hdr_spice, keep_original_imager_pixel_size, level):
        raise NotImplementedError

    def _create_map_from_hdu(self, hdr_spice, basename_output=None, path_output=None, print_filename=True, level=2,
                             keep_original_imager_pixel_size=False):
        """map_builder.py:89-214."""
        hdr_target, col_seconds, t_ref, hdr_im0 = self._prepare_spectro_data(hdr_spice, keep_original_imager_pixel_size,
                                                                             level)
        ny, nx = int(hdr_target["NAXIS2"]), int(hdr_target["NAXIS1"])
        threshold = _seconds(self.threshold_time)
        chosen = np.empty(nx, dtype=np.int64)
        self.dates_selected = [None] * nx
        for ii in range(nx):
            utc_slit = t_ref + _dt.timedelta(seconds=float(col_seconds[ii]))
            index_closest, dt = self._find_closest_imager_time(utc_slit)
            if dt > threshold:
                raise ValueError(f"{dt=}: Could not find imager sufficiently close in time")
            chosen[ii] = index_closest
            self.dates_selected[ii] = self.dates[index_closest]
        h = _lib.shared_handle(-1 if self.device is None else self.device)
        h.reference_tag = None
        self.data_composed = np.empty((ny, nx), dtype=np.float64)
        headers_used = {}
        for idx in np.unique(chosen):
            path = self.list_imager_paths[int(idx)]
            if print_filename:
                print(f"\nUse imager {os.path.basename(path)}")
            data_imager, hdr_imager = fits_io.read_image(path, self.window_imager)
            hdr_imager = fits_io.Header(hdr_imager)
            headers_used[int(idx)] = hdr_imager
            # interpol2d writes into an array of the imager's dtype (utils/Util.py:94-96): float32 pixels give
            # float32-rounded samples
            out_dtype = np.float32 if np.asarray(data_imager).dtype == np.float32 else np.float64
            h.set_small(data_imager if out_dtype == np.float32 else np.asarray(data_imager, dtype=np.float64))
            hw = hdr_imager.copy()
            hdrutil.check_and_create_pcij_matrix(hw, False, warn=False)
            sampled = h.resample_helioprojective(hdr_target, hw, order=2, dtype=out_dtype)
            cols = np.nonzero(chosen == idx)[0]
            self.data_composed[:, cols] = sampled[:, cols]
        # header: the imager frame of the middle raster step, SPICE pointing keywords on top (map_builder.py:133-153)
        used_in_order = [headers_used[int(i)] for i in chosen]
        self.hdr_composed = used_in_order[len(used_in_order) // 2].copy()
        keys = [f"{k}{n}" for k in ("CRPIX", "CRVAL", "CDELT", "CUNIT") for n in (1, 2, 3, 4)] + ["CROTA2", "CROTA"]
        keys += [f"PC{i}_{j}" for i in (1, 2, 3, 4) for j in (1, 2, 3, 4)]
        missing = []
        for k in keys:
            if k in self.hdr_spice_:
                self.hdr_composed[k] = self.hdr_spice_[k]
            else:
                missing.append(k)
        if missing:
            warnings.warn(f"{missing} not in original header. Not added to the synthetic raster header")
        for k in ("DATE-AVG", "DATE-OBS", "DATE-BEG"):
            self.hdr_composed[k] = hdr_spice[k]
        self.hdr_composed["SPECPATH"] = os.path.basename(str(self.path_to_spectro))
        utc_composed, _ = self._return_mean_time(self.dates_selected)
        wave = self.hdr_composed["WAVELNTH"]
        if "DETECTOR" in self.hdr_composed:
            detector = self.hdr_composed["DETECTOR"]
        elif "INSTRUME" in self.hdr_composed:
            detector = self.hdr_composed["INSTRUME"]
        else:
            raise ValueError("No info on reference instrument")
        if keep_original_imager_pixel_size:
            # map_builder.py:163-189, literally: imager pixel size, same roll angle, reference pixel at the centre of
            # the composed map, pointing at the sky position of the SPICE window's central pixel
            hc = self.hdr_composed
            sx, sy = self._spice_shape  # naxis1, naxis2 of map_builder.py:254-255 (level 2) / :293-294 (level 3)
            w_xy = wcs_tan.TanWcs(dict(self.hdr_spice_, NAXIS1=sx, NAXIS2=sy))
            x_mid, y_mid = (sx - 1) / 2, (sy - 1) / 2
            lon_mid, lat_mid = w_xy.pixel_to_world(np.array([x_mid]), np.array([y_mid]))
            hc["CDELT1"] = float(hdrutil.convert(hdr_im0["CDELT1"], hdr_im0["CUNIT1"], hc["CUNIT1"]))
            hc["CDELT2"] = float(hdrutil.convert(hdr_im0["CDELT2"], hdr_im0["CUNIT2"], hc["CUNIT2"]))
            lam = hc["CDELT2"] / hc["CDELT1"]
            rho = np.arccos(hc["PC1_1"]) * (-np.sign(hc["PC1_2"]))
            hc["PC1_2"] = float(-lam * np.sin(rho))
            hc["PC2_1"] = float((1 / lam) * np.sin(rho))
            hc["CRPIX1"] = (self.data_composed.shape[1] + 1) / 2
            hc["CRPIX2"] = (self.data_composed.shape[0] + 1) / 2
            hc["CRVAL1"] = float(hdrutil.convert(lon_mid[0], "deg", hc["CUNIT1"]))
            hc["CRVAL2"] = float(hdrutil.convert(lat_mid[0], "deg", hc["CUNIT2"]))
        if basename_output is None:
            date = utc_composed.strftime("%Y-%m-%dT%H_%M_%S")
            basename_new = f"solo_L3_{detector}{wave}-image-composed-{date}_{random.randint(1, 99999):05d}.fits"
        else:
            basename_new = basename_output
        if path_output is not None:
            fits_io.write_images(os.path.join(self.path_output, basename_new), [(self.data_composed, self.hdr_composed)])
            self.path_composed_map = os.path.join(self.path_output, basename_new)
            return self.path_composed_map
        if level != 2:
            raise NotImplementedError
        self.hdr_composed["NAXIS1"] = self.data_composed.shape[1]
        self.hdr_composed["NAXIS2"] = self.data_composed.shape[0]
        return None


class SPICEComposedMapBuilder(ComposedMapBuilder):

    def _prepare_spectro_data(self, hdr_spice, keep_original_imager_pixel_size, level):
        """map_builder.py:251-349: the 2-D geometry of the composed map and the time of each of its columns.  Level 2:
        axes (x, y, wavelength, time), the reference drops the wavelength axis (:254); level 3: axes (coefficient, x, y,
        time), it drops the coefficient axis (:297-299).  Either way what is left is the (HPLN, HPLT, time) WCS, found
        here by CTYPE.  Returns (target header with NAXIS1/2, seconds after the reference epoch per column, that epoch,
        header of the first imager)."""
        if level not in (2, 3):
            raise ValueError("level must be 2 or 3")
        lon_ax, lat_ax = spice_header._axes(hdr_spice)
        if (level == 2 and (lon_ax, lat_ax) != (1, 2)) or (level == 3 and (lon_ax, lat_ax) != (2, 3)):
            raise ValueError(f"level {level} SPICE header expected, found HPLN/HPLT on axes {lon_ax}, {lat_ax}")
        flat = spice_header.celestial_header(hdr_spice)
        self.hdr_spice_ = flat
        col_seconds, t_ref = spice_header.column_times(hdr_spice)
        hdr_im = self.headers[0]
        target = flat.copy()
        nx, ny = int(hdr_spice["NAXIS%d" % lon_ax]), int(hdr_spice["NAXIS%d" % lat_ax])
        self._spice_shape = (nx, ny)
        if keep_original_imager_pixel_size:
            # sample positions x = k * r1, y = l * r2 in SPICE pixels (np.arange(0, NAXIS, r), map_builder.py:262-276)
            # == an ordinary header with PC'_ij = PC_ij * r_j and CRPIX'_j = 1 + (CRPIX_j - 1) / r_j
            r1 = hdr_im["CDELT1"] / hdr_spice["CDELT%d" % lon_ax]  # (level 3: CDELT2 / CDELT3, map_builder.py:311-313)
            r2 = hdr_im["CDELT2"] / hdr_spice["CDELT%d" % lat_ax]
            xs, ys = np.arange(0, nx, r1), np.arange(0, ny, r2)
            pc = [[float(flat.get(f"PC{i}_{j}", 1.0 if i == j else 0.0)) for j in (1, 2)] for i in (1, 2)]
            for i in (1, 2):
                target[f"PC{i}_1"] = pc[i - 1][0] * r1
                target[f"PC{i}_2"] = pc[i - 1][1] * r2
            target["CRPIX1"] = 1.0 + (flat["CRPIX1"] - 1.0) / r1
            target["CRPIX2"] = 1.0 + (flat["CRPIX2"] - 1.0) / r2
            # time of the fractional columns: linear in x
            full = np.arange(nx, dtype=np.float64)
            col_seconds = np.interp(xs, full, col_seconds) if nx > 1 else np.repeat(col_seconds, len(xs))
            nx, ny = len(xs), len(ys)
        target["NAXIS1"], target["NAXIS2"] = nx, ny
        return target, col_seconds, t_ref, hdr_im
