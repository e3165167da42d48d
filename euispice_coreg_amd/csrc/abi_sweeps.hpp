// Part of libcoreg_hip.so's ONE translation unit (coreg_hip.hip includes the parts in order; round 6 split by concern,
// no behaviour change): C ABI: coreg_sweep_carrington / coreg_sweep_helioprojective, grid-shared sums, pivots, stats.
#pragma once
extern "C" {
int coreg_sweep_carrington(coreg_handle* h, const coreg_wcs2d* hdr_small, const coreg_carr_grid* grid, double solar_r,
                           const coreg_lags* lags, int order, int method, int cdelt_semantics, int64_t lag_begin,
                           int64_t lag_end, double* corr_out, int out_on_device) {
    if (!h) return COREG_EINVAL;
    const ComboRange combo = take_combo_range(h);
    trace("sweep_carrington: enter");
    if (!hdr_small || !grid) return fail(h, COREG_EINVAL, "sweep_carrington: null header/grid");
    if (method != COREG_METHOD_CORRELATION && method != COREG_METHOD_RESIDUS)
        return fail(h, COREG_ENOTIMPL, "method must be COREG_METHOD_CORRELATION or COREG_METHOD_RESIDUS");
    RETCHK(check_order(h, order));
    RETCHK(check_wcs(h, hdr_small, true));
    RETCHK(check_grid(h, grid));
    if (!std::isfinite(solar_r) || !(solar_r > 0.0)) return fail(h, COREG_EINVAL, "solar_r must be positive");
    LagDims d;
    RETCHK(check_lags(h, lags, &d, lag_begin, lag_end, combo));
    RETCHK(bind_device_nowait(h));  // (the image to align is joined right before k_sweep: launch_sweep)
    if (h->ref.p && (h->gW != grid->n_lon || h->gH != grid->n_lat))
        return fail(h, COREG_EINVAL, "reference-on-grid shape differs from the Carrington grid");
    const long long n_out = lag_end - lag_begin;
    double* out_dev = nullptr;
    RETCHK(begin_sweep(h, n_out, corr_out, out_on_device, &out_dev));
    if (n_out == 0) return end_sweep(h, n_out, corr_out, out_on_device, out_dev);  // (nothing to fill)

    CarrDev cd;
    std::memset(&cd, 0, sizeof(cd));
    RETCHK(upload_carr_tables(h, *grid, *hdr_small, &cd));

    // ---- plan: local geometry (heuristic inputs only) -> tile shape + lag patch
    Geometry geo;
    {
        const CarrCommon c0 = carr_common(*hdr_small, solar_r);
        const int ic = grid->n_lon / 2, jc = grid->n_lat / 2;
        double a0, a1, b0, b1, c0x, c0y;
        carr_term_host(h->tabs, c0, ic, jc, &a0, &a1);
        carr_term_host(h->tabs, c0, std::min(ic + 1, grid->n_lon - 1), jc, &b0, &b1);
        carr_term_host(h->tabs, c0, ic, std::min(jc + 1, grid->n_lat - 1), &c0x, &c0y);
        geo.dx_di = b0 - a0;
        geo.dy_di = b1 - a1;
        geo.dx_dj = c0x - a0;
        geo.dy_dj = c0y - a1;
        // utils/rectify.py:399-404: X0 = -(c d1 + s d2)/cdelt1, Y0 = -(-s d1 + c d2)/cdelt2
        const double s1 = lag_step(lags->crval1, d.n1), s2 = lag_step(lags->crval2, d.n2);
        geo.ax = c0.cr * s1 / hdr_small->cdelt1;
        geo.ay = c0.sr * s1 / hdr_small->cdelt2;
        geo.bx = c0.sr * s2 / hdr_small->cdelt1;
        geo.by = c0.cr * s2 / hdr_small->cdelt2;
    }
    const long long row = (long long)d.n2 * d.nc;
    const int m1 = (int)((lag_end - 1) / row) - (int)(lag_begin / row) + 1;
    const Plan plan = choose_plan(h, geo, m1, d.n2, h->opt_use_lds ? lds_window_elems(h) : (1LL << 40));

    PrecomputeArgs pa;
    std::memset(&pa, 0, sizeof(pa));
    {
        const int th = kTilePts / plan.tile_w;
        RETCHK(reserve_tiles(h, ((h->gW + plan.tile_w - 1) / plan.tile_w) * ((h->gH + th - 1) / th)));
    }
    fill_precompute_common(h, &pa, plan.tile_w);
    pa.residus = method == COREG_METHOD_RESIDUS ? 1 : 0;
    const int n_tiles = pa.tiles_x * pa.tiles_y;

    // ---- every (cdelt1, cdelt2, crota) combination = one precompute + one sweep launch; all lag parameters of all
    //      launches are staged together and uploaded once
    struct Launch {
        size_t slot_off;
        int n_batches;
        CarrCommon cc;
        double f0lo, f0hi, f1lo, f1hi;
    };
    std::vector<Launch> launches;
    std::vector<double> params;  // per launch: [X0 x ns][Y0 x ns]
    std::vector<long long> outidx;
    SlotList slots;
    for (long long c = 0; c < d.nc; ++c) {
        {
            const long long first = (lag_begin - c + d.nc - 1) / d.nc;  // smallest k with k*nc + c >= begin
            if (first * d.nc + c >= lag_end) continue;
        }
        int i3, i4, i5;
        d.inner(c, &i3, &i4, &i5);
        coreg_wcs2d hc;
        if (shift_header(*hdr_small, 0.0, 0.0, lags->cdelt1[i3], lags->cdelt2[i4], lags->crota[i5], cdelt_semantics,
                         &hc))
            continue;  // reference semantics: this lag kills the worker -> NaN (already filled)
        build_slots(d, c, lag_begin, lag_end, plan.sw, plan.sh, &slots);
        if (slots.n_batches == 0) continue;
        const size_t ns = slots.i1.size();
        Launch L;
        L.slot_off = outidx.size();
        L.n_batches = slots.n_batches;
        L.cc = carr_common(hc, solar_r);
        const size_t pbase = params.size();
        params.resize(pbase + 2 * ns);
        double x0min = 1e300, x0max = -1e300, y0min = 1e300, y0max = -1e300;
        // utils/rectify.py:396-404 with the roll trig hoisted out of the per-lag loop (same values, same order)
        const double roll = hc.crota * kDeg2Rad;
        const double rc = std::cos(roll), rs = std::sin(roll);
        const double nan = std::numeric_limits<double>::quiet_NaN();
        for (size_t s = 0; s < ns; ++s) {
            // padding lanes get NaN: they fail the bounds rule for every point, so waves made only of padding
            // skip every point with one branch (their slots are never written by k_finalize)
            if (slots.outidx[s] < 0) {
                params[pbase + s] = nan;
                params[pbase + ns + s] = nan;
                continue;
            }
            const double v1 = hdr_small->crval1 + lags->crval1[slots.i1[s]];  // alignment.py:404
            const double v2 = hdr_small->crval2 + lags->crval2[slots.i2[s]];  // alignment.py:412
            const double dx = rc * v1 + rs * v2;
            const double dy = -rs * v1 + rc * v2;
            const double x0 = (hc.crpix1 - 1) - dx / hc.cdelt1;
            const double y0 = (hc.crpix2 - 1) - dy / hc.cdelt2;
            params[pbase + s] = x0;
            params[pbase + ns + s] = y0;
            x0min = std::min(x0min, x0);
            x0max = std::max(x0max, x0);
            y0min = std::min(y0min, y0);
            y0max = std::max(y0max, y0);
        }
        outidx.insert(outidx.end(), slots.outidx.begin(), slots.outidx.end());
        // a point can be in bounds for some lag only if X0 + t0 in [0, W-1] for some X0 in [x0min, x0max]
        L.f0lo = -x0max;
        L.f0hi = (double)(h->sW - 1) - x0min;
        L.f1lo = -y0max;
        L.f1hi = (double)(h->sH - 1) - y0min;
        launches.push_back(L);
    }
    if (launches.empty()) {
        RETCHK(fill_nan(h, out_dev, n_out));
        return end_sweep(h, n_out, corr_out, out_on_device, out_dev);
    }
    RETCHK(upload_plan(h, params, outidx, out_dev, n_out));
    RETCHK(prepare_sharded(h, outidx.size(), n_out, lag_begin));
    for (const Launch& L : launches) {
        set_carr_common(&cd, L.cc);
        pa.carr = cd;
        pa.f0lo = L.f0lo;
        pa.f0hi = L.f0hi;
        pa.f1lo = L.f1lo;
        pa.f1hi = L.f1hi;
        {
            // whole-tile skip bound (k_precompute): pixels per radian of grid-point motion, grid steps in radians
            const double dm1 = L.cc.dist - 1.0;
            pa.tile_skip = (h->opt_tile_skip && dm1 > 0.0) ? 1 : 0;
            const double per_rad = pa.tile_skip ? (1.0 / dm1 + 1.0 / (dm1 * dm1)) * kRad2Deg * 3600.0 : 0.0;
            pa.lip_x = per_rad / std::fabs(L.cc.cdelt1) * (1.0 + 1e-9);
            pa.lip_y = per_rad / std::fabs(L.cc.cdelt2) * (1.0 + 1e-9);
            // float32 linspace grid: the spacing is uniform to ~1e-7 relative of the coordinate
            pa.dlon = grid->n_lon > 1 ? (std::fabs(grid->lon1 - grid->lon0) / (grid->n_lon - 1) * 1.001 + 1e-4) * kDeg2Rad : 0.0;
            pa.dlat = grid->n_lat > 1 ? (std::fabs(grid->lat1 - grid->lat0) / (grid->n_lat - 1) * 1.001 + 1e-4) * kDeg2Rad : 0.0;
        }
        {
            const int ng = pick_groups(h, L.n_batches, n_tiles), nb = L.n_batches;
            RETCHK(launch_precompute<MODE_TRANSLATE>(h, pa, n_tiles, ng, nb));
            h->last_precompute = [pa, n_tiles, ng, nb](coreg_handle* hh) {
                return launch_precompute<MODE_TRANSLATE>(hh, pa, n_tiles, ng, nb);
            };
        }
        // SoA block of this launch starts at 2 * slot_off doubles (every earlier launch contributed 2 per slot)
        RETCHK(launch_sweep(h, MODE_TRANSLATE, order, method, h->lane_params.as<double>() + 2 * L.slot_off,
                            h->out_index.as<long long>() + L.slot_off, L.n_batches, n_tiles, lag_begin, out_dev, nullptr,
                            nullptr, (long long)L.slot_off,
                            pick_pitch(h, plan, h->opt_use_lds ? lds_window_elems(h) : 0, order)));
    }
    return end_sweep(h, n_out, corr_out, out_on_device, out_dev);
}

// Plate-carree maps on both sides (Alignment.align_using_initial_carrington, alignment.py:344-399 ->
// _interpolate_on_large_data_grid :1018-1029 with WCS(CRLN-CAR)): the per-lag map is a rotation of the sphere between
// the native frames of the two maps (a CRVAL2 lag makes the shifted map oblique).  One precompute (native angles of
// the target pixels), one sweep launch per (cdelt1, cdelt2, crota) combination (its native -> pixel affine map is a
// launch constant).  Lags whose header has no valid native pole get NaN (astropy raises for them).
static int sweep_car(coreg_handle* h, const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr_small, const coreg_lags* lags,
                     const LagDims& d, int order, int method, int cdelt_semantics, int64_t lag_begin, int64_t lag_end,
                     double* corr_out, int out_on_device, double* out_dev) {
    const long long n_out = lag_end - lag_begin;
    Mat3 r_target;
    if (car_native_to_celestial(*hdr_target, &r_target))
        return fail(h, COREG_EINVAL, "hdr_target: no valid native pole for this CRVAL2 / LONPOLE (CAR)");
    auto shifted_by = [&](const coreg_wcs2d& base, double v1, double v2) {
        coreg_wcs2d hl = base;
        hl.crval1 = hdr_small->crval1 + v1;  // alignment.py:404
        hl.crval2 = hdr_small->crval2 + v2;  // alignment.py:412
        return hl;
    };
    auto shifted = [&](const coreg_wcs2d& base, int i1, int i2) {
        return shifted_by(base, lags->crval1[i1], lags->crval2[i2]);
    };
    // ---- plan: local geometry from the maps of a central lag and of that lag plus one mean step on either axis
    Geometry geo;
    {
        int e1[3], e2[3];
        extreme_lags(lags->crval1, d.n1, e1);
        extreme_lags(lags->crval2, d.n2, e2);
        const double v1 = lags->crval1[e1[1]], v2 = lags->crval2[e2[1]];
        CarMapHost m0, m1h, m2h;
        if (m0.init(*hdr_target, shifted_by(*hdr_small, v1, v2)) ||
            m1h.init(*hdr_target, shifted_by(*hdr_small, v1 + lag_step(lags->crval1, d.n1), v2)) ||
            m2h.init(*hdr_target, shifted_by(*hdr_small, v1, v2 + lag_step(lags->crval2, d.n2)))) {
            geo.dx_di = geo.dy_dj = 1.0;  // central lag invalid: any plan will do, its lanes are NaN
            geo.dy_di = geo.dx_dj = geo.ax = geo.ay = geo.bx = geo.by = 0.0;
        } else {
            const double u = hdr_target->naxis1 * 0.5, v = hdr_target->naxis2 * 0.5;
            double x0, y0, x1, y1;
            m0.apply(u, v, &x0, &y0);
            m0.apply(u + 1, v, &x1, &y1);
            geo.dx_di = x1 - x0;
            geo.dy_di = y1 - y0;
            m0.apply(u, v + 1, &x1, &y1);
            geo.dx_dj = x1 - x0;
            geo.dy_dj = y1 - y0;
            m1h.apply(u, v, &x1, &y1);
            geo.ax = x1 - x0;
            geo.ay = y1 - y0;
            m2h.apply(u, v, &x1, &y1);
            geo.bx = x1 - x0;
            geo.by = y1 - y0;
        }
    }
    const long long row = (long long)d.n2 * d.nc;
    const int m1 = (int)((lag_end - 1) / row) - (int)(lag_begin / row) + 1;
    const Plan plan = choose_plan(h, geo, m1, d.n2, h->opt_use_lds ? lds_window_elems(h) : (1LL << 40));

    // rotation of every (CRVAL1, CRVAL2) lag: R = R_small(lag)^T * R_target  (PC / CDELT do not enter it)
    const int i1_lo = (int)(lag_begin / row), i1_hi = (int)((lag_end - 1) / row);
    std::vector<double> rot((size_t)(i1_hi - i1_lo + 1) * d.n2 * 9);
    const double nanv = std::numeric_limits<double>::quiet_NaN();
    for (int i1 = i1_lo; i1 <= i1_hi; ++i1)
        for (int i2 = 0; i2 < d.n2; ++i2) {
            double* r = &rot[((size_t)(i1 - i1_lo) * d.n2 + i2) * 9];
            Mat3 rs;
            if (car_native_to_celestial(shifted(*hdr_small, i1, i2), &rs)) {
                for (int k = 0; k < 9; ++k) r[k] = nanv;
                continue;
            }
            const Mat3 m = mat_mul(mat_T(rs), r_target);
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) r[3 * a + b] = (double)m.m[a][b];
        }

    struct Launch {
        size_t slot_off;
        int n_batches;
        LaunchU inv;
        bool identity;  // the one-slot launch of the identity lag-point (below)
        // the single-sample pass of this launch: the combination's header, the CRVAL lag indices of its
        // slots, and the slots the scan skips -- all but the lags that leave CRVAL1 or CRVAL2 of the target header alone:
        // only those bring whole rows or columns of coordinates (or, with a CROTA / CDELT lag on top, the reference pixel)
        // back within wcslib's noise of integers
        coreg_wcs2d hc;
        std::vector<int> i1, i2;
        std::vector<unsigned char> tap_skip;
        bool tap_any = false;
    };
    std::vector<Launch> launches;
    std::vector<double> params;  // per launch: SoA [9][slots of the launch]
    std::vector<long long> outidx;
    SlotList slots;
    // The identity lag-point (shifted header == target header: the zero lag of the sub-map semantics, where the target IS
    // the header of the map to align).  The reference's pixel -> world -> pixel round trip (alignment.py:1038-1069) returns
    // i + eps there and the sign of wcslib's rounding noise decides the bounds rule on every border pixel (and, for odd
    // spline orders, the tap set of every pixel).  The sphere rotation of this path cannot even return exact integers, so
    // that lag-point is taken out of the CAR launch and swept on its own with the EXACT identity map by the
    // helioprojective kernels, whose zero-lag machinery (k_border_fix / k_parity_fix) then applies what wcslib's chain
    // (geometry.hpp WcslibCar, bit-exact) decides.  "border_fix" 0: rotation path for that lag-point too.
    long long identity_out = -1;
    BorderFix id_fix;
    std::vector<unsigned char> id_flags;
    auto same_header = [](const coreg_wcs2d& a, const coreg_wcs2d& b) {
        auto eq = [](double x, double y) { return x == y || (x != x && y != y); };
        return a.proj == b.proj && a.crpix1 == b.crpix1 && a.crpix2 == b.crpix2 && a.crval1 == b.crval1 &&
               a.crval2 == b.crval2 && a.cdelt1 == b.cdelt1 && a.cdelt2 == b.cdelt2 && a.pc1_1 == b.pc1_1 &&
               a.pc1_2 == b.pc1_2 && a.pc2_1 == b.pc2_1 && a.pc2_2 == b.pc2_2 && a.unit_to_deg == b.unit_to_deg &&
               eq(a.lonpole, b.lonpole) && eq(a.latpole, b.latpole) && a.naxis1 == b.naxis1 && a.naxis2 == b.naxis2;
    };
    for (long long c = 0; c < d.nc; ++c) {
        const long long first = (lag_begin - c + d.nc - 1) / d.nc;
        if (first * d.nc + c >= lag_end) continue;
        int i3, i4, i5;
        d.inner(c, &i3, &i4, &i5);
        coreg_wcs2d hc;
        if (shift_header(*hdr_small, 0.0, 0.0, lags->cdelt1[i3], lags->cdelt2[i4], lags->crota[i5], cdelt_semantics,
                         &hc))
            continue;  // reference semantics: this lag kills the worker -> NaN (already filled)
        build_slots(d, c, lag_begin, lag_end, plan.sw, plan.sh, &slots);
        if (slots.n_batches == 0) continue;
        const size_t ns = slots.i1.size();
        Launch L;
        L.slot_off = outidx.size();
        L.n_batches = slots.n_batches;
        const Affine2 inv = car_native_to_pix(hc);
        std::memset(&L.inv, 0, sizeof(L.inv));
        L.inv.m00 = inv.m00;
        L.inv.m01 = inv.m01;
        L.inv.m10 = inv.m10;
        L.inv.m11 = inv.m11;
        L.inv.b0 = inv.b0;
        L.inv.b1 = inv.b1;
        L.inv.box_c = car_box_c(*hdr_target, hc, plan.tile_w);
        L.inv.pole_sep = 0.0;  // largest over the lags of this launch (below)
        L.identity = false;
        const size_t pbase = params.size();
        params.resize(pbase + 9 * ns);
        for (size_t s = 0; s < ns; ++s) {
            bool pad = slots.outidx[s] < 0;
            if (!pad && h->opt_border_fix && identity_out < 0 && h->gW == h->sW && h->gH == h->sH) {
                const coreg_wcs2d hl = shifted(hc, slots.i1[s], slots.i2[s]);
                if (same_header(hl, *hdr_target)) {
                    identity_out = slots.outidx[s];
                    slots.outidx[s] = -1;  // not this launch's: padding lane (NaN map, nothing written)
                    pad = true;
                }
            }
            const double* r = &rot[((size_t)(slots.i1[s] - i1_lo) * d.n2 + slots.i2[s]) * 9];
            for (int k = 0; k < 9; ++k) params[pbase + (size_t)k * ns + s] = pad ? nanv : r[k];
            if (!pad && r[8] == r[8]) L.inv.pole_sep = std::max(L.inv.pole_sep, car_pole_sep(r));
            unsigned char skip = 1;
            if (!pad && h->opt_tap_fix && r[8] == r[8] && h->gW == h->sW && h->gH == h->sH) {
                // (a CROTA / CDELT lag on top of it included: with both CRVAL equal the map is affine about CRPIX and
                // returns the reference pixel itself to within the noise)
                const coreg_wcs2d hl = shifted(hc, slots.i1[s], slots.i2[s]);
                if (hl.crval1 == hdr_target->crval1 || hl.crval2 == hdr_target->crval2) skip = 0;
            }
            L.tap_skip.push_back(skip);
            L.tap_any = L.tap_any || !skip;
        }
        L.hc = hc;
        L.i1 = slots.i1;
        L.i2 = slots.i2;
        outidx.insert(outidx.end(), slots.outidx.begin(), slots.outidx.end());
        launches.push_back(L);
    }
    if (identity_out >= 0) {
        // one batch, one live slot: the identity homography (every sample ON its pixel); the others are padding
        Launch L;
        std::memset(&L.inv, 0, sizeof(L.inv));
        L.slot_off = outidx.size();
        L.n_batches = 1;
        L.identity = true;
        const size_t ns = kBlock, pbase = params.size();
        params.resize(pbase + 9 * ns, nanv);
        const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int k = 0; k < 9; ++k) params[pbase + (size_t)k * ns] = ident[k];
        outidx.push_back(identity_out);
        outidx.insert(outidx.end(), ns - 1, -1);
        launches.push_back(L);
        {   // (on every rank of a grid-sharded sweep: launch_sweep applies it on rank 0 only, but all must agree that this
            // launch carries a correction)
            AxisInvariance inv;
            inv.rows = inv.cols = true;
            BorderFix::Item it;
            it.slot = 0;
            it.first = 0;
            wcslib_dropped_border_pixels(h, *hdr_target, *hdr_target, inv, &id_fix.pixels);
            it.n = (int)id_fix.pixels.size();
            it.flags_off = -1;
            if (order & 1) {
                it.flags_off = 0;
                id_flags = wcslib_tap_shift_flags(h, *hdr_target, *hdr_target, inv);
            }
            if (it.n > 0 || it.flags_off >= 0) id_fix.items.push_back(it);
        }
    }
    if (launches.empty()) {
        RETCHK(fill_nan(h, out_dev, n_out));
        return end_sweep(h, n_out, corr_out, out_on_device, out_dev);
    }
    trace("sweep_helioprojective: lane parameters laid out");
    RETCHK(upload_plan(h, params, outidx, out_dev, n_out));
    RETCHK(prepare_sharded(h, outidx.size(), n_out, lag_begin));
    trace("sweep_helioprojective: plan in page-locked memory");

    PrecomputeArgs pa;
    std::memset(&pa, 0, sizeof(pa));
    {
        const int th = kTilePts / plan.tile_w;
        RETCHK(reserve_tiles(h, ((h->gW + plan.tile_w - 1) / plan.tile_w) * ((h->gH + th - 1) / th)));
    }
    fill_precompute_common(h, &pa, plan.tile_w);
    pa.residus = method == COREG_METHOD_RESIDUS ? 1 : 0;
    const int n_tiles = pa.tiles_x * pa.tiles_y;
    const Affine2 fwd = car_pix_to_native(*hdr_target);
    pa.car_fwd.m00 = fwd.m00;
    pa.car_fwd.m01 = fwd.m01;
    pa.car_fwd.m10 = fwd.m10;
    pa.car_fwd.m11 = fwd.m11;
    pa.car_fwd.b0 = fwd.b0;
    pa.car_fwd.b1 = fwd.b1;
    const double inf = std::numeric_limits<double>::infinity();
    pa.f0lo = pa.f1lo = -inf;  // no culling by position: only non-finite reference values drop out
    pa.f0hi = pa.f1hi = inf;
    if (!id_fix.items.empty()) RETCHK(upload_border_pixels(h, id_fix.pixels));
    if (!id_flags.empty()) {
        HIPCHK(h->border_flags.reserve(id_flags.size()));
        HIPCHK(hipStreamSynchronize(h->stream));  // (pageable source, rare path)
        HIPCHK(hipMemcpy(h->border_flags.p, id_flags.data(), id_flags.size(), hipMemcpyHostToDevice));
    }
    int last_groups = -1, last_batches = -1;
    for (const Launch& L : launches) {
        if (L.identity) {
            // target pixel -> the same pixel of the map to align: base coordinates = pixel indices, no culling by position
            PrecomputeArgs pi = pa;
            std::memset(&pi.car_fwd, 0, sizeof(pi.car_fwd));
            const int ng = pick_groups(h, 1, n_tiles);
            RETCHK(launch_precompute<MODE_HOMOGRAPHY>(h, pi, n_tiles, ng, 1));
            h->last_precompute = [pi, n_tiles, ng](coreg_handle* hh) {
                return launch_precompute<MODE_HOMOGRAPHY>(hh, pi, n_tiles, ng, 1);
            };
            last_groups = last_batches = -1;  // (the compacted points now hold pixel indices, not unit vectors)
            RETCHK(launch_sweep(h, MODE_HOMOGRAPHY, order, method, h->lane_params.as<double>() + 9 * L.slot_off,
                                h->out_index.as<long long>() + L.slot_off, 1, n_tiles, lag_begin, out_dev, nullptr, &id_fix,
                                (long long)L.slot_off));
            continue;
        }
        // the work partition (k_tile_list) depends on the group count of the launch: redo it only when that changes
        const int ng = pick_groups(h, L.n_batches, n_tiles);
        if (ng != last_groups || L.n_batches != last_batches) RETCHK(launch_precompute<MODE_CAR>(h, pa, n_tiles, ng, L.n_batches));
        {
            const int nb = L.n_batches;
            h->last_precompute = [pa, n_tiles, ng, nb](coreg_handle* hh) {
                return launch_precompute<MODE_CAR>(hh, pa, n_tiles, ng, nb);
            };
        }
        last_groups = ng;
        last_batches = L.n_batches;
        BorderFix tap;  // (no whole-grid items here: the identity lag has its own launch)
        if (L.tap_any) {
            const double box[4] = {0.0, (double)(h->gW - 1), 0.0, (double)(h->gH - 1)};
            h->tap_last[0] = h->tap_last[1] = h->tap_last[2] = 0;
            RETCHK(prepare_tap_fix(
                h, MODE_CAR, order, *hdr_target, (long long)L.tap_skip.size(), L.tap_skip, box,
                [&](int slot) { return shifted(L.hc, L.i1[(size_t)slot], L.i2[(size_t)slot]); }, &tap,
                h->lane_params.as<double>() + 9 * L.slot_off, &L.inv, &pa.car_fwd));
        }
        RETCHK(launch_sweep(h, MODE_CAR, order, method, h->lane_params.as<double>() + 9 * L.slot_off,
                            h->out_index.as<long long>() + L.slot_off, L.n_batches, n_tiles, lag_begin, out_dev,
                            &L.inv, L.tap_any ? &tap : nullptr, (long long)L.slot_off));
    }
    return end_sweep(h, n_out, corr_out, out_on_device, out_dev);
}

int coreg_sweep_helioprojective(coreg_handle* h, const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr_small,
                                const coreg_lags* lags, int order, int method, int cdelt_semantics, int64_t lag_begin,
                                int64_t lag_end, double* corr_out, int out_on_device) {
    if (!h) return COREG_EINVAL;
    const ComboRange combo = take_combo_range(h);
    if (!hdr_target || !hdr_small) return fail(h, COREG_EINVAL, "sweep_helioprojective: null header");
    if (hdr_target->proj != hdr_small->proj || (hdr_small->proj != COREG_PROJ_TAN && hdr_small->proj != COREG_PROJ_CAR))
        return fail(h, COREG_ENOTIMPL, "both headers must be TAN (helioprojective) or both CAR (Carrington maps)");
    if (method != COREG_METHOD_CORRELATION && method != COREG_METHOD_RESIDUS)
        return fail(h, COREG_ENOTIMPL, "method must be COREG_METHOD_CORRELATION or COREG_METHOD_RESIDUS");
    RETCHK(check_order(h, order));
    RETCHK(check_wcs(h, hdr_target, false));
    RETCHK(check_wcs(h, hdr_small, false));
    LagDims d;
    RETCHK(check_lags(h, lags, &d, lag_begin, lag_end, combo));
    RETCHK(bind_device_nowait(h));
    if (h->ref.p && (h->gW != hdr_target->naxis1 || h->gH != hdr_target->naxis2))
        return fail(h, COREG_EINVAL, "reference-on-grid shape differs from hdr_target NAXIS1/NAXIS2");
    const long long n_out = lag_end - lag_begin;
    double* out_dev = nullptr;
    RETCHK(begin_sweep(h, n_out, corr_out, out_on_device, &out_dev));
    if (n_out == 0) return end_sweep(h, n_out, corr_out, out_on_device, out_dev);  // (nothing to fill)
    if (hdr_small->proj == COREG_PROJ_CAR)
        return sweep_car(h, hdr_target, hdr_small, lags, d, order, method, cdelt_semantics, lag_begin, lag_end, corr_out,
                         out_on_device, out_dev);

    trace("sweep_helioprojective: enter (checks done)");
    // ---- plan: local geometry from the maps of the central lag and of its two neighbours
    Geometry geo;
    {
        auto map_of = [&](double v1, double v2, double hm[9]) {
            coreg_wcs2d hl = *hdr_small;
            hl.crval1 = hdr_small->crval1 + v1;
            hl.crval2 = hdr_small->crval2 + v2;
            homography(*hdr_target, hl, hm);
        };
        int e1[3], e2[3];
        extreme_lags(lags->crval1, d.n1, e1);
        extreme_lags(lags->crval2, d.n2, e2);
        const double v1 = lags->crval1[e1[1]], v2 = lags->crval2[e2[1]];
        double m0[9], m1h[9], m2h[9];
        map_of(v1, v2, m0);
        map_of(v1 + lag_step(lags->crval1, d.n1), v2, m1h);
        map_of(v1, v2 + lag_step(lags->crval2, d.n2), m2h);
        const double u = hdr_target->naxis1 * 0.5, v = hdr_target->naxis2 * 0.5;
        double x0, y0, x1, y1;
        apply_h(m0, u, v, &x0, &y0);
        apply_h(m0, u + 1, v, &x1, &y1);
        geo.dx_di = x1 - x0;
        geo.dy_di = y1 - y0;
        apply_h(m0, u, v + 1, &x1, &y1);
        geo.dx_dj = x1 - x0;
        geo.dy_dj = y1 - y0;
        apply_h(m1h, u, v, &x1, &y1);
        geo.ax = x1 - x0;
        geo.ay = y1 - y0;
        apply_h(m2h, u, v, &x1, &y1);
        geo.bx = x1 - x0;
        geo.by = y1 - y0;
    }
    const long long row = (long long)d.n2 * d.nc;
    const int m1 = (int)((lag_end - 1) / row) - (int)(lag_begin / row) + 1;
    const Plan plan = choose_plan(h, geo, m1, d.n2, h->opt_use_lds ? lds_window_elems(h) : (1LL << 40));

    // ---- all slots of all (cdelt1, cdelt2, crota) combinations -> ONE launch
    // Two passes.  (1) per combination, independent of every other and of the handle -- a few host threads share them
    // when the lag set is large (cfg4: 21 combinations x 3 721 lag-points, 1.4 ms of 3 x 3 products on one core):
    // shifted header, slots, one homography per slot, the combination's corner of the cull box.  (2) in combination
    // order, on this thread: the lag-points decided by wcslib's rounding noise (handle caches), the bookkeeping of the
    // odd-order pass, the concatenation.
    std::vector<long long> outidx;
    int n_batches = 0;
    double fx0 = 1e300, fx1 = -1e300, fy0 = 1e300, fy1 = -1e300;  // cull box in target pixels
    HomographyFamily fam;
    fam.init(*hdr_target, *hdr_small, lags->crval1, d.n1, lags->crval2, d.n2, d.nc > 1);
    BorderFix fix;
    std::vector<std::vector<unsigned char>> flags_host;  // per noise-decided lag-point (odd spline orders only)
    // odd spline orders: what prepare_tap_fix needs to rebuild a slot's shifted header
    // (grid shares across GPUs: every rank lists the samples -- the re-evaluation of a flagged lag-point needs them on
    // every rank; the correction itself is launched on rank 0 only, launch_sweep)
    // Odd orders: every sample within 1e-8 px of an integer coordinate (the sign of wcslib's noise picks the taps); even
    // orders: only those within 1e-8 px of a BOUND of the image (the sign decides the bounds rule) -- a pure CRVAL1 or
    // CRVAL2 lag under an unrotated header keeps whole border rows / columns of the grid there.
    const bool tap_fixing = h->opt_tap_fix != 0;
    std::vector<coreg_wcs2d> tap_combo;      // the (cdelt, crota)-shifted header of each combination
    std::vector<int> tap_slot_combo, tap_slot_i1, tap_slot_i2;
    std::vector<unsigned char> tap_skip;     // padding lanes and lag-points the structured fix handles
    const int i1_lo = (int)(lag_begin / row), i1_hi = (int)((lag_end - 1) / row);
    fam.fill_products(i1_lo, i1_hi);  // (read-only from here on: `get` is safe to call from several threads)
    const double nanv = std::numeric_limits<double>::quiet_NaN();
    struct ComboPlan {
        bool used = false;
        coreg_wcs2d hc;
        SlotList slots;
        std::vector<double> hs;  // AoS [slot][9]
        double box[4] = {1e300, -1e300, 1e300, -1e300};
    };
    std::vector<ComboPlan> cps((size_t)d.nc);
    // which target pixels can ever be in bounds: inverse maps of the small image's corners for the lags on the
    // boundary of the (CRVAL1, CRVAL2) rectangle, chosen BY VALUE (the reference accepts lag lists in any order):
    // smallest, largest and the value nearest the middle of each axis (the maps vary smoothly and monotonically
    // with the lag value, the +-3 px margin below covers the curvature in between)
    int e1[3], e2[3];
    extreme_lags(lags->crval1 + i1_lo, i1_hi - i1_lo + 1, e1);
    extreme_lags(lags->crval2, d.n2, e2);
    auto plan_combo = [&](long long c) {
        ComboPlan& cp = cps[(size_t)c];
        const long long first = (lag_begin - c + d.nc - 1) / d.nc;
        if (first * d.nc + c >= lag_end) return;
        int i3, i4, i5;
        d.inner(c, &i3, &i4, &i5);
        if (shift_header(*hdr_small, 0.0, 0.0, lags->cdelt1[i3], lags->cdelt2[i4], lags->crota[i5], cdelt_semantics,
                         &cp.hc))
            return;
        build_slots(d, c, lag_begin, lag_end, plan.sw, plan.sh, &cp.slots);
        if (cp.slots.n_batches == 0) return;
        cp.used = true;
        const Mat3d B = HomographyFamily::combo(cp.hc);
        const size_t n = cp.slots.i1.size();
        cp.hs.resize(9 * n);
        for (size_t s = 0; s < n; ++s) {
            double* hm = &cp.hs[9 * s];
            if (cp.slots.outidx[s] < 0) {  // padding lane: NaN map -> never in bounds
                for (int k = 0; k < 9; ++k) hm[k] = nanv;
            } else {
                fam.get(B, cp.slots.i1[s], cp.slots.i2[s], hm);
            }
        }
        for (int a1 = 0; a1 < 3; ++a1)
            for (int a2 = 0; a2 < 3; ++a2) {
                coreg_wcs2d hl = cp.hc;
                hl.crval1 = hdr_small->crval1 + lags->crval1[i1_lo + e1[a1]];
                hl.crval2 = hdr_small->crval2 + lags->crval2[e2[a2]];
                double hi[9];
                homography(hl, *hdr_target, hi);
                for (int k = 0; k < 4; ++k) {
                    double px, py;
                    apply_h(hi, (k & 1) ? (double)(h->sW - 1) : 0.0, (k & 2) ? (double)(h->sH - 1) : 0.0, &px, &py);
                    cp.box[0] = std::min(cp.box[0], px);
                    cp.box[1] = std::max(cp.box[1], px);
                    cp.box[2] = std::min(cp.box[2], py);
                    cp.box[3] = std::max(cp.box[3], py);
                }
            }
    };
    trace("sweep_helioprojective: geometry + lag family ready");
    const unsigned plan_threads = (d.nc >= 2 && (long long)d.nc * d.n1 * d.n2 >= 16384)
                                      ? std::min<unsigned>({8u, (unsigned)d.nc, std::max(1u, std::thread::hardware_concurrency())})
                                      : 1u;
    if (plan_threads <= 1) {
        for (long long c = 0; c < d.nc; ++c) plan_combo(c);
    } else {
        std::atomic<long long> next(0);
        auto worker = [&] {
            for (long long c = next.fetch_add(1); c < d.nc; c = next.fetch_add(1)) plan_combo(c);
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < plan_threads; ++t) th.emplace_back(worker);
        worker();
        for (auto& x : th) x.join();
    }
    for (long long c = 0; c < d.nc; ++c) {
        ComboPlan& cp = cps[(size_t)c];
        if (!cp.used) continue;
        const SlotList& slots = cp.slots;
        const coreg_wcs2d& hc = cp.hc;
        if (tap_fixing) {
            tap_combo.push_back(hc);
            for (size_t s = 0; s < slots.i1.size(); ++s) {
                tap_slot_combo.push_back((int)tap_combo.size() - 1);
                tap_slot_i1.push_back(slots.outidx[s] < 0 ? 0 : slots.i1[s]);
                tap_slot_i2.push_back(slots.outidx[s] < 0 ? 0 : slots.i2[s]);
                tap_skip.push_back(slots.outidx[s] < 0 ? 1 : 0);
            }
        }
        if (h->opt_border_fix) {
            for (size_t s = 0; s < slots.i1.size(); ++s) {
                if (slots.outidx[s] < 0) continue;
                // same tangent point as the target (sub-map path, zero CRVAL lag) and an invariant image axis: exact
                // invariant map on the device, border pixels decided as the reference's wcslib round trip decides them
                // (geometry.hpp WcslibTan, k_border_fix).  (The tangent points are compared first: every other
                // lag-point is dismissed without building its header.)
                const double v1 = hdr_small->crval1 + lags->crval1[slots.i1[s]];
                const double v2 = hdr_small->crval2 + lags->crval2[slots.i2[s]];
                if (v1 != hdr_target->crval1 || v2 != hdr_target->crval2) continue;
                coreg_wcs2d hl = hc;
                hl.crval1 = v1;
                hl.crval2 = v2;
                double* hm = &cp.hs[9 * s];
                const AxisInvariance inv = snap_invariant_axes(*hdr_target, hl, h->gW, h->gH, hm);
                if (inv.rows || inv.cols) {
                    BorderFix::Item it;
                    it.slot = (long long)(outidx.size() + s);
                    it.first = (int)fix.pixels.size();
                    wcslib_dropped_border_pixels(h, *hdr_target, hl, inv, &fix.pixels);
                    it.n = (int)fix.pixels.size() - it.first;
                    it.flags_off = -1;
                    if (order & 1) {
                        it.flags_off = (long long)flags_host.size() * h->gW * h->gH;
                        flags_host.push_back(wcslib_tap_shift_flags(h, *hdr_target, hl, inv));  // (copy: the cache may evict)
                    }
                    if (it.n > 0 || it.flags_off >= 0) fix.items.push_back(it);
                    if (tap_fixing) tap_skip[(size_t)it.slot] = 1;  // (its whole grid sits on integers: k_parity_fix)
                }
            }
        }
        fx0 = std::min(fx0, cp.box[0]);
        fx1 = std::max(fx1, cp.box[1]);
        fy0 = std::min(fy0, cp.box[2]);
        fy1 = std::max(fy1, cp.box[3]);
        outidx.insert(outidx.end(), slots.outidx.begin(), slots.outidx.end());
        n_batches += slots.n_batches;
    }
    if (n_batches == 0) {
        RETCHK(fill_nan(h, out_dev, n_out));
        return end_sweep(h, n_out, corr_out, out_on_device, out_dev);
    }
    trace("sweep_helioprojective: combinations planned");
    const size_t ns = outidx.size();
    std::vector<double> params(9 * ns);
    double eps_max = 0.0;  // largest |h6 x + h7 y| over the target grid and all lags
    {
        // AoS per combination -> SoA [9][ns] of the launch, the same threads over the combinations
        std::vector<size_t> off((size_t)d.nc + 1, 0);
        for (long long c = 0; c < d.nc; ++c) off[(size_t)c + 1] = off[(size_t)c] + (cps[(size_t)c].used ? cps[(size_t)c].slots.i1.size() : 0);
        std::vector<double> eps_of((size_t)d.nc, 0.0);
        auto transpose = [&](long long c) {
            const ComboPlan& cp = cps[(size_t)c];
            if (!cp.used) return;
            const size_t n = cp.slots.i1.size(), at = off[(size_t)c];
            double em = 0.0;
            for (size_t s = 0; s < n; ++s) {
                const double* hm = &cp.hs[9 * s];
                for (int k = 0; k < 9; ++k) params[(size_t)k * ns + at + s] = hm[k];
                const double e = std::fabs(hm[6]) * (double)h->gW + std::fabs(hm[7]) * (double)h->gH;
                if (e == e) em = std::max(em, e);
            }
            eps_of[(size_t)c] = em;
        };
        if (plan_threads <= 1) {
            for (long long c = 0; c < d.nc; ++c) transpose(c);
        } else {
            std::atomic<long long> next(0);
            auto worker = [&] {
                for (long long c = next.fetch_add(1); c < d.nc; c = next.fetch_add(1)) transpose(c);
            };
            std::vector<std::thread> th;
            for (unsigned t = 1; t < plan_threads; ++t) th.emplace_back(worker);
            worker();
            for (auto& x : th) x.join();
        }
        for (double e : eps_of) eps_max = std::max(eps_max, e);
    }
    // 1/(1 + eps) = 1 - eps + eps^2 is exact to float64 below ~4e-6 (eps^3 < 1e-16); wider fields divide exactly
    const int sweep_mode = (h->opt_h_series && eps_max < 4.0e-6) ? MODE_HOMOGRAPHY_SERIES : MODE_HOMOGRAPHY;
    RETCHK(upload_plan(h, params, outidx, out_dev, n_out));
    RETCHK(prepare_sharded(h, outidx.size(), n_out, lag_begin));

    PrecomputeArgs pa;
    std::memset(&pa, 0, sizeof(pa));
    {
        const int th = kTilePts / plan.tile_w;
        RETCHK(reserve_tiles(h, ((h->gW + plan.tile_w - 1) / plan.tile_w) * ((h->gH + th - 1) / th)));
    }
    fill_precompute_common(h, &pa, plan.tile_w);
    pa.residus = method == COREG_METHOD_RESIDUS ? 1 : 0;
    const int n_tiles = pa.tiles_x * pa.tiles_y;
    // the maps are projective and the image corners bound its interior
    pa.f0lo = std::floor(fx0) - 3.0;
    pa.f0hi = std::ceil(fx1) + 3.0;
    pa.f1lo = std::floor(fy0) - 3.0;
    pa.f1hi = std::ceil(fy1) + 3.0;
    if (!fix.items.empty()) RETCHK(upload_border_pixels(h, fix.pixels));
    if (!flags_host.empty()) {
        const size_t each = (size_t)h->gW * h->gH;
        HIPCHK(h->border_flags.reserve(each * flags_host.size()));
        HIPCHK(hipStreamSynchronize(h->stream));  // (pageable source, rare path: blocking copies are fine)
        for (size_t k = 0; k < flags_host.size(); ++k)
            HIPCHK(hipMemcpy(h->border_flags.as<unsigned char>() + k * each, flags_host[k].data(), each,
                             hipMemcpyHostToDevice));
    }
    {
        const int ng = pick_groups(h, n_batches, n_tiles);
        RETCHK(launch_precompute<MODE_HOMOGRAPHY>(h, pa, n_tiles, ng, n_batches));
        h->last_precompute = [pa, n_tiles, ng, n_batches](coreg_handle* hh) {
            return launch_precompute<MODE_HOMOGRAPHY>(hh, pa, n_tiles, ng, n_batches);
        };
    }
    h->tap_last[0] = h->tap_last[1] = h->tap_last[2] = 0;
    const double tap_box[4] = {pa.f0lo, pa.f0hi, pa.f1lo, pa.f1hi};
    trace("sweep_helioprojective: precompute launched");
    if (tap_fixing)
        RETCHK(prepare_tap_fix(
            h, sweep_mode, order, *hdr_target, (long long)ns, tap_skip, tap_box,
            [&](int slot) {
                coreg_wcs2d hl = tap_combo[(size_t)tap_slot_combo[(size_t)slot]];
                hl.crval1 = hdr_small->crval1 + lags->crval1[tap_slot_i1[(size_t)slot]];
                hl.crval2 = hdr_small->crval2 + lags->crval2[tap_slot_i2[(size_t)slot]];
                return hl;
            },
            &fix));
    trace("sweep_helioprojective: single-sample scan done");
    RETCHK(launch_sweep(h, sweep_mode, order, method, h->lane_params.as<double>(), h->out_index.as<long long>(), n_batches,
                        n_tiles, lag_begin, out_dev, nullptr, &fix, 0,
                        pick_pitch(h, plan, h->opt_use_lds ? lds_window_elems(h) : 0, order)));
    return end_sweep(h, n_out, corr_out, out_on_device, out_dev);
}

int coreg_sums_size(coreg_handle* h, int64_t* n_doubles) {
    if (!h || !n_doubles) return COREG_EINVAL;
    *n_doubles = (int64_t)h->sums_slots * kNumSums;
    return COREG_OK;
}

int coreg_copy_sums(coreg_handle* h, double* dst, int dst_on_device) {
    if (!h || !dst) return COREG_EINVAL;
    if (h->sums_slots <= 0 || h->pending_fin.empty()) return fail(h, COREG_ESTATE, "no point-sharded sweep is pending");
    RETCHK(bind_device(h));
    const size_t bytes = (size_t)h->sums_slots * kNumSums * sizeof(double);
    HIPCHK(hipMemcpyAsync(dst, h->sums.p, bytes, dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                          h->stream));
    if (!dst_on_device) HIPCHK(hipStreamSynchronize(h->stream));
    return COREG_OK;
}

int coreg_finalize_sums(coreg_handle* h, const double* sums, int sums_on_device, double* corr_out, int out_on_device) {
    if (!h || !sums) return COREG_EINVAL;
    if (h->sums_slots <= 0 || h->pending_fin.empty()) return fail(h, COREG_ESTATE, "no point-sharded sweep is pending");
    if (!corr_out && h->pending_n_out > 0) return fail(h, COREG_EINVAL, "corr_out is null");
    RETCHK(bind_device(h));
    const long long n_out = h->pending_n_out;
    const size_t bytes = (size_t)h->sums_slots * kNumSums * sizeof(double);
    if ((const void*)sums != h->sums.p) {
        HIPCHK(hipMemcpyAsync(h->sums.p, sums, bytes, sums_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                              h->stream));
        // a host buffer belongs to the caller again on return (with a device destination nothing below waits)
        if (!sums_on_device) HIPCHK(hipStreamSynchronize(h->stream));
    }
    double* out_dev = corr_out;
    if (!out_on_device) {
        HIPCHK(h->out_dev.reserve((size_t)std::max<long long>(n_out, 1) * sizeof(double)));
        out_dev = h->out_dev.as<double>();
    }
    if (n_out > 0) {
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, h->stream, out_dev,
                           (long long)n_out, std::numeric_limits<double>::quiet_NaN());
        HIPCHK(hipGetLastError());
    }
    const size_t n_pending = h->pending_fin.size();
    size_t points_of = n_pending - 1;  // the launch whose compacted points the handle holds: the sweep's last one
    // a plate-carree sweep has one pending launch per combination, and asking each one "anything flagged?"
    // used to cost a host round trip apiece.  With several launches pending, all of them are finalized and their flags
    // COUNTED first (k_refine_list adds into a counter of its own, counters[2]): ONE read-back -- and when the total is
    // zero, the normal case, that was all.  Otherwise the per-launch pass below runs as before (k_finalize is idempotent).
    bool probe_says_nothing_flagged = false;
    if (n_pending > 1) {
        long long* probe = h->counters.as<long long>() + 2;
        HIPCHK(hipMemsetAsync(probe, 0, sizeof(long long), h->stream));
        for (size_t ip = 0; ip < n_pending; ++ip) {
            const coreg_handle::PendingFinalize& pf = h->pending_fin[ip];
            FinalizeArgs f = {};
            RETCHK(fill_refine(h, &f.refine, pf.refine.mode, pf.refine.order, pf.refine.lane_params, pf.refine.car_inv,
                               pf.n_slots));
            f.refine.enabled = pf.refine.enabled;
            f.refine_count = nullptr;  // (the counters of the sweep are written by the pass that re-evaluates)
            f.partials = h->sums.as<double>() + pf.slot_off;
            f.n_groups = 1;
            f.n_slots = pf.n_slots;
            f.part_stride = h->sums_slots;
            f.out_index = h->fin_outidx.as<long long>() + pf.slot_off;
            f.lag_begin = pf.lag_begin;
            f.out = out_dev;
            f.residus = pf.residus;
            f.n_required = (long long)h->gW * h->gH;
            hipLaunchKernelGGL(k_finalize, dim3((unsigned)((pf.n_slots + kFinSlots - 1) / kFinSlots)),
                               dim3(kFinSlots * kFinLanes), 0, h->stream, f);
            if (f.refine.enabled)
                hipLaunchKernelGGL(k_refine_list, dim3(1), dim3(kListThreads), 0, h->stream, f.refine, pf.n_slots, probe);
        }
        HIPCHK(hipGetLastError());
        long long flagged = 0;
        HIPCHK(hipMemcpyAsync(&flagged, probe, sizeof(flagged), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        probe_says_nothing_flagged = flagged == 0;
    }
    for (size_t ip = 0; ip < n_pending && !probe_says_nothing_flagged; ++ip) {
        const coreg_handle::PendingFinalize& pf = h->pending_fin[ip];
        FinalizeArgs f = {};
        // flags from the REDUCED sums: the same on every rank.  (work-space pointers taken afresh: a later launch of the
        // sweep may have grown the buffers)
        RETCHK(fill_refine(h, &f.refine, pf.refine.mode, pf.refine.order, pf.refine.lane_params, pf.refine.car_inv,
                           pf.n_slots));
        f.refine.enabled = pf.refine.enabled;
        const RefineArgs rf = f.refine;
        f.refine_count = h->counters.as<long long>();
        f.partials = h->sums.as<double>() + pf.slot_off;
        f.n_groups = 1;
        f.n_slots = pf.n_slots;
        f.part_stride = h->sums_slots;
        f.out_index = h->fin_outidx.as<long long>() + pf.slot_off;
        f.lag_begin = pf.lag_begin;
        f.out = out_dev;
        f.residus = pf.residus;
        f.n_required = (long long)h->gW * h->gH;
        f.sums_out = nullptr;
        f.sums_stride = f.sums_off = 0;
        hipLaunchKernelGGL(k_finalize, dim3((unsigned)((pf.n_slots + kFinSlots - 1) / kFinSlots)), dim3(kFinSlots * kFinLanes),
                           0, h->stream, f);
        if (!rf.enabled) continue;
        // Ill-conditioned lag-points: every rank holds both images and re-evaluates them over the WHOLE grid (not its
        // share) with the same kernels in the same order -- identical coefficients on every rank, and equal to the
        // single-GPU sweep's, without a second collective.  The compacted points of a launch that was not the sweep's
        // last have been overwritten by the later launches: computed again, only when something is flagged.
        hipLaunchKernelGGL(k_refine_list, dim3(1), dim3(kListThreads), 0, h->stream, rf, pf.n_slots, h->counters.as<long long>());
        HIPCHK(hipGetLastError());
        int head[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(head, rf.head, sizeof(head), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (head[0] == 0) continue;
        if (points_of != ip && pf.replay_precompute) {
            RETCHK(pf.replay_precompute(h));
            points_of = ip;
        }
        RefineArgs rf2 = rf;
        if (!pf.fixes.empty()) {
            // the launch's noise-decided samples about the flagged slots' own pivots (as launch_sweep does on one GPU)
            HIPCHK(h->rf_fix_slab.reserve((size_t)kNumSums * pf.n_slots * sizeof(double)));
            HIPCHK(hipMemsetAsync(h->rf_fix_slab.p, 0, (size_t)kNumSums * pf.n_slots * sizeof(double), h->stream));
            rf2.fix_slab = h->rf_fix_slab.as<double>();
            RETCHK(launch_fix_kernels(h, pf.fixes, h->rf_fix_slab.as<double>(), rf.slot_pivots, rf.flags));
        }
        RETCHK(launch_refine(h, rf2, pf.n_slots, f.out_index, pf.lag_begin, out_dev, false));
    }
    HIPCHK(hipGetLastError());
    if (!out_on_device && n_out > 0) {
        HIPCHK(hipMemcpyAsync(corr_out, out_dev, (size_t)n_out * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return COREG_OK;
}

int coreg_get_pivots(coreg_handle* h, double* pivots2) {
    if (!h || !pivots2) return COREG_EINVAL;
    RETCHK(bind_device(h));
    HIPCHK(hipMemcpyAsync(pivots2, h->pivots.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return COREG_OK;
}

int coreg_set_pivots(coreg_handle* h, const double* pivots2) {
    if (!h || !pivots2) return COREG_EINVAL;
    if (!(pivots2[0] == pivots2[0]) || !(pivots2[1] == pivots2[1]) || std::isinf(pivots2[0]) || std::isinf(pivots2[1]))
        return fail(h, COREG_EINVAL, "set_pivots: pivots must be finite");
    RETCHK(bind_device(h));
    HIPCHK(hipMemcpyAsync(h->pivots.p, pivots2, 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));  // the caller's two doubles are free again on return
    return COREG_OK;
}

int coreg_last_stats(coreg_handle* h, coreg_stats* out) {
    if (!h || !out) return COREG_EINVAL;
    RETCHK(bind_device(h));
    RETCHK(collect_stats(h));  // waits for an in-flight device-output sweep
    *out = h->stats;
    return COREG_OK;
}

}  // extern "C"
