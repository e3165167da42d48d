"""
`AlignmentResults` -- drop-in for euispice_coreg.hdrshift.AlignmentResults (hdrshift/AlignmentResults.py:23-354):
argmax of the 6-D correlation array, 2-D Gaussian sub-lag refinement (the reference's `curve_fit` call restated in the
library, `coreg_fit_gaussian2d`: same algorithm, same stopping rule, ~0.1 ms instead of ~20 ms; `fit="scipy"` makes the
literal scipy call), corrected-header / corrected-FITS output.  `plot_correlation` draws the correlation map (matplotlib, optional);
`plot_co_alignment` (image overlays, plot/plot.py) is presentation code outside the path and raises NotImplementedError.
"""
from __future__ import annotations

import os
import warnings

import numpy as np

from .. import _lib
from ..utils import fits_io, header as hdrutil


def twoD_Gaussian(xy, amplitude, xo, yo, sigma_x, sigma_y, offset):
    """AlignmentResults.py:12-21."""
    x, y = xy
    x0 = float(xo)
    y0 = float(yo)
    g = offset + amplitude * np.exp(-((((x - x0) ** 2) / (2 * sigma_x ** 2)) + (((y - y0) ** 2) / (2 * sigma_y ** 2))))
    return g.ravel()


class AlignmentResults:

    def __init__(self, corr, lag_crval1, lag_crval2, lag_cdelt1, lag_cdelt2, lag_crota, unit_lag,
                 image_to_align_path=None, image_to_align_window=None, reference_image_path=None,
                 reference_image_window=None, fit=None):
        # fit: "native" (default; COREG_GAUSSIAN_FIT overrides) = csrc/fit.hpp, "scipy" = scipy.optimize.curve_fit
        fit = fit or os.environ.get("COREG_GAUSSIAN_FIT", "native")
        if fit not in ("native", "scipy"):
            raise ValueError("fit must be 'native' or 'scipy'")
        self.fit = fit
        self.fit_info = None

        def arr(v):
            return np.array([0.0]) if v is None else np.atleast_1d(np.asarray(v, dtype=np.float64))

        lag_crval1, lag_crval2, lag_cdelt1, lag_cdelt2, lag_crota = (arr(v) for v in (lag_crval1, lag_crval2,
                                                                                      lag_cdelt1, lag_cdelt2,
                                                                                      lag_crota))
        corr = np.asarray(corr)
        self.max_index = np.unravel_index(np.nanargmax(corr), corr.shape)
        self.corr = corr
        # the reference stores astropy Quantities here; plain arrays in `unit_lag` (crota in deg) without astropy
        self.parameters_alignment = {"lag_crval1": lag_crval1, "lag_crval2": lag_crval2, "lag_cdelt1": lag_cdelt1,
                                     "lag_cdelt2": lag_cdelt2, "lag_crota": lag_crota}
        to_as = hdrutil.unit_to_deg(unit_lag) / hdrutil.unit_to_deg("arcsec")
        self.parameters_alignment_arcsec = {"lag_crval1": lag_crval1 * to_as, "lag_crval2": lag_crval2 * to_as,
                                            "lag_cdelt1": lag_cdelt1 * to_as, "lag_cdelt2": lag_cdelt2 * to_as,
                                            "lag_crota": lag_crota}
        self.image_to_align_path = image_to_align_path
        self.image_to_align_window = image_to_align_window
        self.reference_image_path = reference_image_path
        self.reference_image_window = reference_image_window
        self.unit_lag = unit_lag
        self.shift_pixels = None
        self.shift_arcsec = None
        self._compute_shift()

    # -- AlignmentResults.py:218-341 ---------------------------------------------------------------------------------
    def _argmax_shift(self):
        p = self.parameters_alignment_arcsec
        mi = self.max_index
        self.shift_pixels = (mi[0], mi[1], mi[2], mi[3], mi[4])
        self.shift_arcsec = (p["lag_crval1"][mi[0]], p["lag_crval2"][mi[1]], p["lag_cdelt1"][mi[2]],
                             p["lag_cdelt2"][mi[3]], p["lag_crota"][mi[4]])

    def _compute_shift(self, method="fitting_gaussian"):
        if method != "fitting_gaussian":
            raise NotImplementedError
        mi = self.max_index
        corr2d = self.corr[:, :, mi[2], mi[3], mi[4]]
        px, py = [mi[0]], [mi[1]]
        lenx, leny = corr2d.shape[0], corr2d.shape[1]
        for ii in (-2, -1, 0, 1, 2):
            for jj in (-2, -1, 0, 1, 2):
                x, y = mi[0] + ii, mi[1] + jj
                # (sic) only index -1 is excluded, the peak sample is duplicated: AlignmentResults.py:230-239, quirk Q13
                if (x != -1) and (x < lenx) and (y != -1) and (y < leny):
                    px.append(x)
                    py.append(y)
        if len(px) < 4:
            warnings.warn(" Cannot compute shift with Gaussian fitting: not enough points")
            self._argmax_shift()
            return None
        p0 = (np.float64(np.ravel(corr2d[mi[0], mi[1]])[0]), np.float64(mi[0]), np.float64(mi[1]), 1.0, 1.0, 0.9)
        bounds = ([0.0, mi[0] - 5.0, mi[1] - 5.0, 0.0, 0.0, -10.0], [10.0, mi[0] + 5.0, mi[1] + 5.0, 1000.0, 1000.0, 10.0])
        try:
            A = (np.float64(px), np.float64(py))
            # the un-excluded index -2 wraps around; on an axis shorter than 2 it is out of range and the reference
            # dies with IndexError -- here that case falls back to the argmax like a failed fit does
            B = np.float64(corr2d[px, py].ravel())
            if self.fit == "scipy":
                from scipy.optimize import curve_fit
                popt, _ = curve_fit(f=twoD_Gaussian, xdata=A, ydata=B, p0=p0, bounds=bounds)
            else:
                # curve_fit's own checks, in its order: finite data (check_finite=True), p0 inside the bounds,
                # finite residuals at p0 -- all ValueError, i.e. the reference's argmax fallback
                if not (np.all(np.isfinite(B)) and np.all(np.isfinite(p0))):
                    raise ValueError("array must not contain infs or NaNs")
                if not all(lo <= v <= hi for v, lo, hi in zip(p0, *bounds)):
                    raise ValueError("`x0` is infeasible.")
                popt, status, nfev = _lib.fit_gaussian2d(A[0], A[1], B, p0, bounds[0], bounds[1])
                self.fit_info = {"status": status, "nfev": nfev}
                if status == -1:
                    raise ValueError("Residuals are not finite in the initial point.")
                if status == 0:  # curve_fit: `if not res.success: raise RuntimeError` -- not caught by the reference
                    raise RuntimeError("Optimal parameters not found: The maximum number of function evaluations is "
                                       "exceeded.")
        except (ValueError, IndexError):
            warnings.warn("Gaussian fitting failed, setting shift params as the pixel of the maximal correlation")
            self._argmax_shift()
            return None
        p = self.parameters_alignment_arcsec
        sx = np.interp(popt[1], np.arange(len(p["lag_crval1"])), p["lag_crval1"])
        sy = np.interp(popt[2], np.arange(len(p["lag_crval2"])), p["lag_crval2"])
        self.shift_pixels = (popt[1], popt[2], mi[2], mi[3], mi[4])
        self.shift_arcsec = (sx, sy, p["lag_cdelt1"][mi[2]], p["lag_cdelt2"][mi[3]], p["lag_crota"][mi[4]])
        return True

    # -- outputs -----------------------------------------------------------------------------------------------------
    def return_corrected_header(self, window, path_to_l2_input=None):
        """AlignmentResults.py:191-214."""
        if path_to_l2_input is None:
            if self.image_to_align_path is None:
                raise ValueError("Please provide a path_to_l2_input parameter")
            path_to_l2_input = self.image_to_align_path
        hdr = fits_io.Header(fits_io.read_header(path_to_l2_input, window))  # (no pixel is decoded for a header)
        s = self.shift_arcsec
        hdrutil.correct_pointing_header(hdr, lag_crval1=s[0], lag_crval2=s[1], lag_cdelt1=s[2], lag_cdelt2=s[3],
                                        lag_crota=s[4])
        return hdr

    def write_corrected_fits(self, window_list_to_apply_shift, path_to_l3_output, path_to_l2_input=None):
        """AlignmentResults.py:149-178 -> utils/Util.py:106-159: copy every HDU, correct the pointing keywords of the
        selected windows (data written as float32, as the reference does)."""
        if path_to_l2_input is None:
            if self.image_to_align_path is None:
                raise ValueError("Please provide a path_to_l2_input parameter")
            path_to_l2_input = self.image_to_align_path
        s = self.shift_arcsec

        def selected(ii, n, hdr):
            extname = hdr.get("EXTNAME", "nothing98695")
            return (extname in window_list_to_apply_shift) or (ii in window_list_to_apply_shift) or \
                ((ii - n) in window_list_to_apply_shift)

        def correct(hdr):
            hdrutil.correct_pointing_header(hdr, lag_crval1=s[0], lag_crval2=s[1], lag_cdelt1=s[2], lag_cdelt2=s[3],
                                            lag_crota=s[4])
        if isinstance(path_to_l2_input, (tuple, list)):
            hdr = fits_io.Header(path_to_l2_input[1])
            data = np.asarray(path_to_l2_input[0])
            n_corrected = 0
            if selected(0, 1, hdr):
                correct(hdr)
                data = np.array(data, dtype="<f4")
                n_corrected = 1
            fits_io.write_images(path_to_l3_output, [(data, hdr)], overwrite=True)
        else:
            # headers re-written, data units copied as they are (no decode / encode of the pixels)
            n_corrected = fits_io.rewrite_with_corrected_headers(path_to_l2_input, path_to_l3_output, selected, correct)
        if n_corrected == 0:
            raise ValueError("has not corrected any window.")

    def plot_correlation(self, path_save_figure=None, show=False, fig=None, ax=None):
        """AlignmentResults.py:93-116 -> plot/plot.py:56-175: the (CRVAL1, CRVAL2) slice of the correlation array through
        its maximum, with the fitted shift marked.  Needs matplotlib (optional dependency)."""
        try:
            import matplotlib
            if not show:
                matplotlib.use("Agg", force=False)
            from matplotlib import pyplot as plt
        except ImportError as e:  # pragma: no cover
            raise NotImplementedError("plot_correlation needs matplotlib") from e
        if self.unit_lag not in ("arcsec", "deg"):
            raise NotImplementedError
        unit = "''" if self.unit_lag == "arcsec" else "deg"
        f = hdrutil.unit_to_deg("arcsec") / hdrutil.unit_to_deg(self.unit_lag)
        mi = self.max_index
        corr = np.asarray(self.corr)[:, :, mi[2], mi[3], mi[4]]
        corr = corr.reshape(corr.shape[0], corr.shape[1], -1)[:, :, 0]
        lag_dx = self.parameters_alignment_arcsec["lag_crval1"] * f
        lag_dy = self.parameters_alignment_arcsec["lag_crval2"] * f
        dx = lag_dx[1] - lag_dx[0] if len(lag_dx) > 1 else 1.0
        dy = lag_dy[1] - lag_dy[0] if len(lag_dy) > 1 else 1.0
        if fig is None:
            fig = plt.figure()
        if ax is None:
            ax = fig.add_subplot()
        finite = corr[np.isfinite(corr)]
        vmin = np.percentile(finite, 30) if finite.size else None
        im = ax.imshow(corr.T, origin="lower", interpolation="none", aspect="auto", vmin=vmin,
                       extent=(lag_dx[0] - 0.5 * dx, lag_dx[-1] + 0.5 * dx, lag_dy[0] - 0.5 * dy, lag_dy[-1] + 0.5 * dy))
        s = self.shift_arcsec
        ax.plot(s[0] * f, s[1] * f, marker="+", color="r", markersize=12,
                label=f"dx={s[0] * f:.2f}{unit}, dy={s[1] * f:.2f}{unit}, dcdelt=({s[2] * f:.3g}, {s[3] * f:.3g}){unit}, "
                      f"drota={s[4]:.3g} deg")
        ax.set_xlabel(f"CRVAL1 [{self.unit_lag}]")
        ax.set_ylabel(f"CRVAL2 [{self.unit_lag}]")
        ax.legend(loc="best", fontsize=7)
        fig.colorbar(im, ax=ax, label="correlation")
        if path_save_figure is not None:
            fig.savefig(path_save_figure)
        if show:
            plt.show()
        return fig, ax

    def plot_co_alignment(self, path_save_figure=None, show=False, lonlims=None, latlims=None, levels_percentile=None,
                          imin=2, imax=97, **kwargs):
        """AlignmentResults.py:121-147 -> plot/plot.py:608-925, its "compare_plot": the reference image, and on top of it
        the contours of the image to align before and after the pointing correction.  The image to align is put on the
        reference image's pixel grid by the library's own resampler (the exact TAN -> TAN map of the sweep, order 2),
        once with the header it came with and once with `return_corrected_header`; nothing else of the reference's
        plotting module is reproduced (no sunpy frames; 2-D image HDUs and SPICE level-2 windows in helioprojective
        coordinates; `wavelength_interval_to_sum` / `sub_fov_window` as the SPICE alignment took them).
        lonlims / latlims: (min, max) of the part of the reference image shown, in the lag unit.  Returns (fig, axes);
        `self.co_alignment` keeps the three arrays that were drawn.  Needs matplotlib (optional dependency)."""
        try:
            import matplotlib
            if not show:
                matplotlib.use("Agg", force=False)
            from matplotlib import pyplot as plt
        except ImportError as e:  # pragma: no cover
            raise NotImplementedError("plot_co_alignment needs matplotlib") from e
        if self.image_to_align_path is None or self.reference_image_path is None:
            raise ValueError("plot_co_alignment needs the paths of both images (AlignmentResults built by Alignment has them)")
        if kwargs.get("type_plot", "compare_plot") != "compare_plot":
            raise NotImplementedError("only type_plot='compare_plot'")
        from .. import _lib
        from ..utils import wcs_tan
        w_small = -1 if self.image_to_align_window is None else self.image_to_align_window
        w_ref = -1 if self.reference_image_window is None else self.reference_image_window
        ref, href = fits_io.read_image(self.reference_image_path, w_ref)
        if np.ndim(ref) != 2:
            raise NotImplementedError("plot_co_alignment: the reference image must be a 2-D image HDU")
        href = fits_io.Header(href)
        hdrutil.check_and_create_pcij_matrix(href, False)
        naxis = int(fits_io.read_header(self.image_to_align_path, w_small).get("NAXIS", 0))
        s = self.shift_arcsec
        if naxis == 4:
            # a SPICE level-2 window: the image the sweep saw -- wavelength planes added, slit edges masked, the 4-D
            # header flattened (hdrshift/alignment_spice.py) -- and the correction applied to that 2-D header
            from .alignment_spice import AlignmentSpice
            S = AlignmentSpice(self.reference_image_path, self.image_to_align_path, small_fov_window=w_small,
                               large_fov_window=w_ref,
                               wavelength_interval_to_sum=kwargs.get("wavelength_interval_to_sum", "all"),
                               sub_fov_window=kwargs.get("sub_fov_window", "all"))
            S.extend_pixel_size = False
            S._extract_spice_data_header(level=2)
            small, hs = S.data_small, fits_io.Header(S.hdr_small)
            hc = hs.copy()
            hdrutil.correct_pointing_header(hc, lag_crval1=s[0], lag_crval2=s[1], lag_cdelt1=s[2], lag_cdelt2=s[3],
                                            lag_crota=s[4])
        elif naxis == 2:
            small, hs = fits_io.load_for_upload(self.image_to_align_path, w_small)
            hs = fits_io.Header(hs)
            hdrutil.check_and_create_pcij_matrix(hs, False)
            hc = self.return_corrected_header(w_small)
        else:
            raise NotImplementedError("plot_co_alignment: 2-D image HDUs and SPICE level-2 windows only")
        hdrutil.check_and_create_pcij_matrix(hc, False)
        h = _lib.shared_handle(-1, 0)
        h.set_small(fits_io.native_pixels(small))
        before = h.resample_helioprojective(href, hs, order=2, dtype=np.float64)
        after = h.resample_helioprojective(href, hc, order=2, dtype=np.float64)
        ref = np.asarray(ref, dtype=np.float64)
        # what is shown: the part of the reference image either version of the image to align covers (or the limits asked
        # for), with a margin
        seen = np.isfinite(before) | np.isfinite(after)
        if lonlims is not None or latlims is not None:
            lon, lat = wcs_tan.pixel_lonlat(href)  # degrees
            lon = (lon + 180.0) % 360.0 - 180.0
            f = hdrutil.unit_to_deg(self.unit_lag)
            if lonlims is not None:
                seen &= (lon >= lonlims[0] * f) & (lon <= lonlims[1] * f)
            if latlims is not None:
                seen &= (lat >= latlims[0] * f) & (lat <= latlims[1] * f)
        if not seen.any():
            raise ValueError("plot_co_alignment: the image to align does not overlap the (selected part of the) reference image")
        jj, ii = np.nonzero(seen)
        m = max(4, int(0.05 * max(jj.max() - jj.min(), ii.max() - ii.min())))
        j0, j1 = max(0, jj.min() - m), min(ref.shape[0], jj.max() + m + 1)
        i0, i1 = max(0, ii.min() - m), min(ref.shape[1], ii.max() + m + 1)
        ref_c, before_c, after_c = ref[j0:j1, i0:i1], before[j0:j1, i0:i1], after[j0:j1, i0:i1]
        self.co_alignment = {"reference": ref_c, "before": before_c, "after": after_c, "window": (j0, j1, i0, i1)}
        levels_percentile = [85] if levels_percentile is None else list(levels_percentile)
        fin = ref_c[np.isfinite(ref_c)]
        vmin, vmax = (np.percentile(fin, imin), np.percentile(fin, imax)) if fin.size else (None, None)
        fig, axes = plt.subplots(1, 3, figsize=(15, 5), sharex=True, sharey=True)
        extent = (i0 - 0.5, i1 - 0.5, j0 - 0.5, j1 - 0.5)
        titles = ("reference image", "image to align: header as it came",
                  f"corrected: dx={s[0]:.2f}'' dy={s[1]:.2f}'' drota={s[4]:.3g} deg")
        for ax, over, title in zip(axes, (None, before_c, after_c), titles):
            ax.imshow(ref_c, origin="lower", interpolation="none", cmap="gray", vmin=vmin, vmax=vmax, extent=extent)
            if over is not None and np.isfinite(over).any():
                lv = sorted(set(float(np.nanpercentile(over, q)) for q in levels_percentile))
                ax.contour(np.arange(i0, i1), np.arange(j0, j1), np.where(np.isfinite(over), over, np.nanmin(over)),
                           levels=lv, colors="r", linewidths=0.6)
            ax.set_title(title, fontsize=9)
            ax.set_xlabel("reference pixel x")
        axes[0].set_ylabel("reference pixel y")
        fig.tight_layout()
        if path_save_figure is not None:
            fig.savefig(path_save_figure)
        if show:
            plt.show()
        return fig, axes

    def savefig(self, filename):
        raise NotImplementedError

    def saveyaml(self, filename, window, path_to_l2_input=None):
        raise NotImplementedError

    def __str__(self):
        s = self.shift_arcsec
        return (f"\n Shift : \n x = {s[0]} '' \n y = {s[1]} '' \n dx = {s[2]} '' "
                f"\n dy = {s[3]} '' \n dcrot = {s[4]} deg")

    __repr__ = __str__
