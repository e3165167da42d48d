// Part of libcoreg_hip.so's ONE translation unit (coreg_hip.hip includes the parts in order; round 6 split by concern,
// no behaviour change): C ABI: host-only helpers (header shift, homographies, wcslib chain, plane sums) + last-sweep counters.
#pragma once
extern "C" {
// ---- host-only helpers (no GPU needed): exported so that the header logic can be tested on CPU --------------------
int coreg_shift_header(const coreg_wcs2d* ref, double d_crval1, double d_crval2, double d_cdelt1, double d_cdelt2,
                       double d_crota, int cdelt_semantics, coreg_wcs2d* out) {
    if (!ref || !out) return COREG_EINVAL;
    return shift_header(*ref, d_crval1, d_crval2, d_cdelt1, d_cdelt2, d_crota, cdelt_semantics, out);
}

int coreg_car_map(const coreg_wcs2d* from, const coreg_wcs2d* to, int64_t n, const double* px, const double* py,
                  double* ox, double* oy) {
    if (!from || !to || n < 0 || (n > 0 && (!px || !py || !ox || !oy))) return COREG_EINVAL;
    if (from->proj != COREG_PROJ_CAR || to->proj != COREG_PROJ_CAR) return COREG_EINVAL;
    CarMapHost m;
    if (m.init(*from, *to)) return 1;
    for (int64_t i = 0; i < n; ++i) m.apply(px[i], py[i], &ox[i], &oy[i]);
    return COREG_OK;
}

int coreg_last_visit_counts(coreg_handle* h, int64_t* counts6) {
    int64_t* counts5 = counts6;
    if (!h || !counts5) return COREG_EINVAL;
    RETCHK(bind_device(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    long long info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (h->tile_info.p) HIPCHK(hipMemcpy(info, h->tile_info.p, 7 * sizeof(long long), hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; ++k) counts5[k] = info[3 + k];
    long long refined[2] = {0, 0};
    if (h->counters.p) HIPCHK(hipMemcpy(refined, h->counters.p, sizeof(refined), hipMemcpyDeviceToHost));
    counts5[4] = refined[0];
    counts6[5] = refined[1];
    return COREG_OK;
}

int coreg_last_tap_fix(coreg_handle* h, int64_t* counts3) {
    if (!h || !counts3) return COREG_EINVAL;
    for (int k = 0; k < 3; ++k) counts3[k] = h->tap_last[k];
    return COREG_OK;
}

int coreg_car_tile_margin(const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr_shifted, int32_t tile_w,
                          double tile_abs_lat_rad, double* margin_px) {
    if (!hdr_target || !hdr_shifted || !margin_px || tile_w < 1 || tile_w > kTilePts) return COREG_EINVAL;
    if (hdr_target->proj != COREG_PROJ_CAR || hdr_shifted->proj != COREG_PROJ_CAR) return COREG_EINVAL;
    CarMapHost m;
    if (m.init(*hdr_target, *hdr_shifted)) return 1;
    *margin_px = car_tile_margin(car_box_c(*hdr_target, *hdr_shifted, tile_w), tile_abs_lat_rad + car_pole_sep(m.r));
    return COREG_OK;
}

int coreg_homography(const coreg_wcs2d* from, const coreg_wcs2d* to, double* h9) {
    if (!from || !to || !h9) return COREG_EINVAL;
    if (from->proj != COREG_PROJ_TAN || to->proj != COREG_PROJ_TAN) return COREG_ENOTIMPL;
    homography(*from, *to, h9);
    return COREG_OK;
}

int coreg_lag_homography(const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr_small, const coreg_lags* lags,
                         const int32_t idx[5], int cdelt_semantics, double* h9) {
    if (!hdr_target || !hdr_small || !lags || !idx || !h9) return COREG_EINVAL;
    if (hdr_target->proj != COREG_PROJ_TAN || hdr_small->proj != COREG_PROJ_TAN) return COREG_ENOTIMPL;
    if (idx[0] < 0 || idx[0] >= lags->n_crval1 || idx[1] < 0 || idx[1] >= lags->n_crval2 || idx[2] < 0 ||
        idx[2] >= lags->n_cdelt1 || idx[3] < 0 || idx[3] >= lags->n_cdelt2 || idx[4] < 0 || idx[4] >= lags->n_crota)
        return COREG_EINVAL;
    coreg_wcs2d hc;
    if (shift_header(*hdr_small, 0.0, 0.0, lags->cdelt1[idx[2]], lags->cdelt2[idx[3]], lags->crota[idx[4]],
                     cdelt_semantics, &hc))
        return 1;
    HomographyFamily fam;
    fam.init(*hdr_target, *hdr_small, lags->crval1, lags->n_crval1, lags->crval2, lags->n_crval2);
    fam.get(HomographyFamily::combo(hc), idx[0], idx[1], h9);
    return COREG_OK;
}

int coreg_wcslib_pixel_to_pixel(const coreg_wcs2d* from, const coreg_wcs2d* to, int64_t n, const double* px,
                                const double* py, double* ox, double* oy, double* lng, double* lat) {
    if (!from || !to || n < 0 || (n > 0 && (!px || !py || !ox || !oy))) return COREG_EINVAL;
    if (from->proj == COREG_PROJ_CAR && to->proj == COREG_PROJ_CAR) {
        WcslibCar a, b;
        a.init(*from);
        b.init(*to);
        if (!a.valid || !b.valid) return COREG_EINVAL;
        for (int64_t i = 0; i < n; ++i) {
            double l, t;
            a.p2s(px[i], py[i], &l, &t);
            if (lng) lng[i] = l;
            if (lat) lat[i] = t;
            b.s2p(l, t, &ox[i], &oy[i]);
        }
        return COREG_OK;
    }
    if (from->proj != COREG_PROJ_TAN || to->proj != COREG_PROJ_TAN) return COREG_ENOTIMPL;
    WcslibTan a, b;
    a.init(*from);
    b.init(*to);
    for (int64_t i = 0; i < n; ++i) {
        double l, t;
        a.p2s(px[i], py[i], &l, &t);
        if (lng) lng[i] = l;
        if (lat) lat[i] = t;
        b.s2p(ang2pipi_deg(l), ang2pipi_deg(t), &ox[i], &oy[i]);
    }
    return COREG_OK;
}

int coreg_carrington_origin(const coreg_wcs2d* hdr, double* x0, double* y0) {
    if (!hdr || !x0 || !y0) return COREG_EINVAL;
    carr_origin(*hdr, x0, y0);
    return COREG_OK;
}

int coreg_nansum_planes_be(const void* cube, int32_t bitpix, int64_t n_pixels, const int64_t* plane_index, int32_t n_sel,
                           double* out) {
    if (!cube || !out || n_pixels < 0 || n_sel < 0 || (n_sel > 0 && !plane_index) || (bitpix != -32 && bitpix != -64))
        return COREG_EINVAL;
    for (int k = 0; k < n_sel; ++k)
        if (plane_index[k] < 0) return COREG_EINVAL;
    auto work = [&](int64_t lo, int64_t hi) {
        for (int64_t p = lo; p < hi; ++p) out[p] = 0.0;
        for (int k = 0; k < n_sel; ++k) {
            if (bitpix == -32) {
                const uint32_t* src = (const uint32_t*)cube + (size_t)plane_index[k] * (size_t)n_pixels;
                for (int64_t p = lo; p < hi; ++p) {
                    const uint32_t u = __builtin_bswap32(src[p]);
                    float f;
                    std::memcpy(&f, &u, sizeof(f));
                    const double v = (double)f;
                    out[p] += (v != v) ? 0.0 : v;
                }
            } else {
                const uint64_t* src = (const uint64_t*)cube + (size_t)plane_index[k] * (size_t)n_pixels;
                for (int64_t p = lo; p < hi; ++p) {
                    const uint64_t u = __builtin_bswap64(src[p]);
                    double v;
                    std::memcpy(&v, &u, sizeof(v));
                    out[p] += (v != v) ? 0.0 : v;
                }
            }
        }
    };
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int nt = (int)std::min<int64_t>(std::min<unsigned>(hw, 12u), std::max<int64_t>(1, n_pixels * std::max(n_sel, 1) / (1 << 18)));
    if (nt <= 1) {
        work(0, n_pixels);
        return COREG_OK;
    }
    std::vector<std::thread> th;
    const int64_t per = ((n_pixels + nt - 1) / nt + 7) & ~(int64_t)7;
    for (int t = 1; t < nt; ++t) th.emplace_back(work, std::min<int64_t>(n_pixels, t * per), std::min<int64_t>(n_pixels, (t + 1) * per));
    work(0, std::min<int64_t>(n_pixels, per));
    for (auto& t : th) t.join();
    return COREG_OK;
}

int coreg_fit_gaussian2d(int32_t m, const double* x, const double* y, const double* z, const double* p0,
                         const double* lb, const double* ub, int32_t jac, double ftol, double xtol, double gtol,
                         int32_t max_nfev, double* popt, int32_t* nfev, int32_t* status) {
    if (!x || !y || !z || !p0 || !lb || !ub || !popt || !status || m < 1 || m > coregfit::MMAX) return COREG_EINVAL;
    coregfit::Problem P{m, x, y, z};
    int n = 0;
    const int st = coregfit::fit(P, p0, lb, ub, jac != 0, ftol > 0 ? ftol : 1e-8, xtol > 0 ? xtol : 1e-8,
                                 gtol > 0 ? gtol : 1e-8, max_nfev, popt, &n, nullptr);
    if (st == -2) return COREG_EINVAL;
    if (nfev) *nfev = n;
    *status = st;
    return COREG_OK;
}

}  // extern "C"
