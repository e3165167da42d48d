// Part of libcoreg_hip.so's ONE translation unit (coreg_hip.hip includes the parts in order; round 6 split by concern,
// no behaviour change): noise-decided samples, host side: invariant axes, wcslib's dropped border pixels and tap-shift flags, the single-sample scan + lists.
#pragma once
namespace {
// Zero-CRVAL lags of a helioprojective sweep share the target header's tangent point, so the map target pixel ->
// shifted pixel is exactly affine, A = (CDELT' PC')^-1 (CDELT PC) about CRPIX.  When A leaves an image axis invariant
// (the zero lag: A = I; a CDELT1-only lag: rows map to rows; CDELT2-only: columns to columns; `reference` CDELT
// semantics: A = I up to the rebuilt PC's last bit) the border rows / columns of the grid sit ON the bounds rule
// c < 0 or c > n-1 and what the reference does with them is decided by the rounding noise of its wcslib round trip
// (alignment.py:1038-1069).  For such a lag the device map is SNAPPED to the exact invariant form (every border
// pixel in bounds along that axis) and the pixels wcslib drops are listed for k_border_fix.
struct AxisInvariance {
    bool rows = false, cols = false;
};
AxisInvariance snap_invariant_axes(const coreg_wcs2d& target, const coreg_wcs2d& shifted, int gw, int gh, double hm[9]) {
    AxisInvariance inv;
    if (target.crval1 != shifted.crval1 || target.crval2 != shifted.crval2 || target.lonpole != shifted.lonpole ||
        target.unit_to_deg != shifted.unit_to_deg)
        return inv;
    const Mat3 a = mat_mul(iwc_to_pix(shifted), pix_to_iwc(target));
    const double tol = 1e-6;  // pixels, over the whole grid; rounding noise is < 1e-9, a real lag moves >> 1e-6
    const double a00 = (double)a.m[0][0], a01 = (double)a.m[0][1], a02 = (double)a.m[0][2];
    const double a10 = (double)a.m[1][0], a11 = (double)a.m[1][1], a12 = (double)a.m[1][2];
    inv.rows = std::fabs(a10) * gw + std::fabs(a11 - 1.0) * gh + std::fabs(a12) < tol;
    inv.cols = std::fabs(a00 - 1.0) * gw + std::fabs(a01) * gh + std::fabs(a02) < tol;
    if (inv.rows || inv.cols) {
        hm[0] = inv.cols ? 1.0 : a00;
        hm[1] = inv.cols ? 0.0 : a01;
        hm[2] = inv.cols ? 0.0 : a02;
        hm[3] = inv.rows ? 0.0 : a10;
        hm[4] = inv.rows ? 1.0 : a11;
        hm[5] = inv.rows ? 0.0 : a12;
        hm[6] = hm[7] = 0.0;
        hm[8] = 1.0;
    }
    return inv;
}

// Pixels of the invariant border rows / columns that the reference's round trip pixel -> sky (target header) ->
// ang2pipi -> pixel (shifted header) sends outside [0, W-1] x [0, H-1] of the image to align.  Appended to `out`.
template <typename Chain>
void wcslib_dropped_border_pixels_t(coreg_handle* h, const coreg_wcs2d& target, const coreg_wcs2d& shifted,
                                    AxisInvariance inv, std::vector<int>* out) {
    const int gw = h->gW, gh = h->gH;
    std::vector<double> key = {(double)target.proj, target.latpole == target.latpole ? target.latpole : -999.0, target.crpix1, target.crpix2, target.crval1, target.crval2, target.cdelt1, target.cdelt2,
                               target.pc1_1, target.pc1_2, target.pc2_1, target.pc2_2, target.unit_to_deg,
                               target.lonpole == target.lonpole ? target.lonpole : -999.0,
                               shifted.crpix1, shifted.crpix2, shifted.cdelt1, shifted.cdelt2, shifted.pc1_1,
                               shifted.pc1_2, shifted.pc2_1, shifted.pc2_2, (double)gw, (double)gh, (double)h->sW,
                               (double)h->sH, inv.rows ? 1.0 : 0.0, inv.cols ? 1.0 : 0.0};
    auto hit = h->border_cache.find(key);
    if (hit == h->border_cache.end()) {
        Chain wf, wt;
        wf.init(target);
        wt.init(shifted);
        std::vector<int> cand;  // row-major, each pixel once
        const int jb = h->sH - 1, ib = h->sW - 1;
        for (int j = 0; j < gh; ++j) {
            const bool row = inv.rows && (j == 0 || j == jb);
            if (row) {
                for (int i = 0; i < gw; ++i) cand.push_back(j * gw + i);
            } else if (inv.cols) {
                cand.push_back(j * gw);
                if (ib > 0 && ib < gw) cand.push_back(j * gw + ib);
            }
        }
        std::vector<char> drop(cand.size(), 0);
        const double wmax = (double)(h->sW - 1), hmax = (double)(h->sH - 1);
        auto work = [&](size_t lo, size_t hi) {
            for (size_t k = lo; k < hi; ++k) {
                double x, y;
                wcslib_pixel_to_pixel(wf, wt, (double)(cand[k] % gw), (double)(cand[k] / gw), &x, &y);
                drop[k] = !((x >= 0.0) && (x <= wmax) && (y >= 0.0) && (y <= hmax));  // NaN -> dropped
            }
        };
        unsigned nt = std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency()));
        if (cand.size() < 2048) nt = 1;
        if (nt <= 1) {
            work(0, cand.size());
        } else {
            std::vector<std::thread> th;
            const size_t step = (cand.size() + nt - 1) / nt;
            for (unsigned t = 0; t < nt; ++t) {
                const size_t lo = std::min(cand.size(), (size_t)t * step), hi = std::min(cand.size(), lo + step);
                if (hi > lo) th.emplace_back(work, lo, hi);
            }
            for (auto& x : th) x.join();
        }
        std::vector<int> dropped;
        for (size_t k = 0; k < cand.size(); ++k)
            if (drop[k]) dropped.push_back(cand[k]);
        if (h->border_cache.size() >= 64) h->border_cache.clear();
        hit = h->border_cache.emplace(std::move(key), std::move(dropped)).first;
    }
    out->insert(out->end(), hit->second.begin(), hit->second.end());
}

void wcslib_dropped_border_pixels(coreg_handle* h, const coreg_wcs2d& target, const coreg_wcs2d& shifted,
                                  AxisInvariance inv, std::vector<int>* out) {
    if (target.proj == COREG_PROJ_CAR) wcslib_dropped_border_pixels_t<WcslibCar>(h, target, shifted, inv, out);
    else wcslib_dropped_border_pixels_t<WcslibTan>(h, target, shifted, inv, out);
}

// Odd spline orders: for every grid pixel, does the reference's round trip come back BELOW the integer along an
// invariant axis (bit 0: rows / y, bit 1: columns / x)?  Then floor(c) -- the first tap of an odd-order spline -- is one
// less than at the exact integer the sweep used (k_parity_fix).  W x H evaluations of the wcslib chain, in threads;
// cached per header pair.
template <typename Chain>
const std::vector<unsigned char>& wcslib_tap_shift_flags_t(coreg_handle* h, const coreg_wcs2d& target,
                                                           const coreg_wcs2d& shifted, AxisInvariance inv) {
    const int gw = h->gW, gh = h->gH;
    std::vector<double> key = {(double)target.proj, target.latpole == target.latpole ? target.latpole : -999.0, target.crpix1, target.crpix2, target.crval1, target.crval2, target.cdelt1, target.cdelt2,
                               target.pc1_1, target.pc1_2, target.pc2_1, target.pc2_2, target.unit_to_deg,
                               target.lonpole == target.lonpole ? target.lonpole : -999.0,
                               shifted.crpix1, shifted.crpix2, shifted.cdelt1, shifted.cdelt2, shifted.pc1_1,
                               shifted.pc1_2, shifted.pc2_1, shifted.pc2_2, (double)gw, (double)gh, (double)h->sW,
                               (double)h->sH, inv.rows ? 1.0 : 0.0, inv.cols ? 1.0 : 0.0};
    auto hit = h->flags_cache.find(key);
    if (hit != h->flags_cache.end()) return hit->second;
    Chain wf, wt;
    wf.init(target);
    wt.init(shifted);
    std::vector<unsigned char> flags((size_t)gw * gh, 0);
    const double wmax = (double)(h->sW - 1), hmax = (double)(h->sH - 1);
    auto work = [&](int j0, int j1) {
        for (int j = j0; j < j1; ++j)
            for (int i = 0; i < gw; ++i) {
                double x, y;
                wcslib_pixel_to_pixel(wf, wt, (double)i, (double)j, &x, &y);
                unsigned char f = 0;
                if (inv.rows && y < (double)j) f |= 1;
                if (inv.cols && x < (double)i) f |= 2;
                // the bounds rule drops the pixel altogether (border pixels only; k_border_fix has taken it out)
                if (!((x >= 0.0) && (x <= wmax) && (y >= 0.0) && (y <= hmax))) f |= 4;
                flags[(size_t)j * gw + i] = f;
            }
    };
    unsigned nt = std::min<unsigned>(12, std::max(1u, std::thread::hardware_concurrency()));
    if ((long long)gw * gh < 4096) nt = 1;
    if (nt <= 1) {
        work(0, gh);
    } else {
        std::vector<std::thread> th;
        const int step = (gh + (int)nt - 1) / (int)nt;
        for (unsigned t = 0; t < nt; ++t) {
            const int lo = std::min(gh, (int)t * step), hi = std::min(gh, lo + step);
            if (hi > lo) th.emplace_back(work, lo, hi);
        }
        for (auto& x : th) x.join();
    }
    if (h->flags_cache.size() >= 4) h->flags_cache.clear();
    return h->flags_cache.emplace(std::move(key), std::move(flags)).first->second;
}

const std::vector<unsigned char>& wcslib_tap_shift_flags(coreg_handle* h, const coreg_wcs2d& target,
                                                         const coreg_wcs2d& shifted, AxisInvariance inv) {
    if (target.proj == COREG_PROJ_CAR) return wcslib_tap_shift_flags_t<WcslibCar>(h, target, shifted, inv);
    return wcslib_tap_shift_flags_t<WcslibTan>(h, target, shifted, inv);
}

int upload_border_pixels(coreg_handle* h, const std::vector<int>& pixels) {
    const size_t bytes = std::max<size_t>(1, pixels.size()) * sizeof(int);
    HIPCHK(h->border_dev.reserve(bytes));
    HIPCHK(hipStreamSynchronize(h->stream));  // an earlier sweep may still read the old list / the staging buffer
    HIPCHK(h->pin_border.reserve(bytes));
    std::memcpy(h->pin_border.p, pixels.data(), pixels.size() * sizeof(int));
    if (!pixels.empty())
        HIPCHK(hipMemcpyAsync(h->border_dev.p, h->pin_border.p, pixels.size() * sizeof(int), hipMemcpyHostToDevice,
                              h->stream));
    return COREG_OK;
}

// Odd spline orders, general case (kernels.hpp k_tap_scan / k_tap_fix).  Called between the precompute launch (whose
// prologue has put the lag parameters on the device) and the sweep launch: list the (slot, pixel) samples whose mapped
// coordinate lies within 1e-8 px of an integer, evaluate wcslib's chain for them on the host (`shifted_of(slot)` gives the
// slot's shifted header), and leave everything k_tap_fix needs on the device.  The list is sorted (slot, pixel): one
// workgroup per slot adds its entries in a fixed order.  A list beyond "tap_cap" entries (a pure CRVAL1 / CRVAL2 lag set
// under an unrotated header at full size) is not applied at all -- recorded in tap_last, coreg_last_tap_fix.
template <typename ShiftedOf>
// `hom_dev`: the launch's lane parameters (null: the handle's whole buffer); `car_inv` / `car_fwd`: MODE_CAR launches.
int prepare_tap_fix(coreg_handle* h, int sweep_mode, int order, const coreg_wcs2d& target, long long n_slots,
                    const std::vector<unsigned char>& skip, const double box[4], ShiftedOf shifted_of, BorderFix* fix,
                    const double* hom_dev = nullptr, const LaunchU* car_inv = nullptr, const LaunchU* car_fwd = nullptr) {
    // the list starts small (64 K entries, or what an earlier sweep needed) and is grown -- and the scan repeated --
    // only when a sweep lists more, up to "tap_cap"
    const unsigned cap_max = (unsigned)h->opt_tap_cap;
    unsigned cap = (unsigned)std::min<size_t>(cap_max, std::max<size_t>((size_t)1 << 16, h->tap_list.cap / sizeof(uint2)));
    HIPCHK(h->tap_count.reserve(2 * sizeof(unsigned)));  // [0] listed samples, [1] queued segments
    const unsigned seg_cap = 1u << 20;                   // 16 MiB of (slot, row, first, end); beyond: tested in-thread
    HIPCHK(h->tap_segq.reserve((size_t)seg_cap * sizeof(uint4)));
    HIPCHK(h->tap_list.reserve((size_t)cap * sizeof(uint2)));
    HIPCHK(h->tap_skip.reserve((size_t)n_slots));
    HIPCHK(hipMemcpyAsync(h->tap_skip.p, skip.data(), (size_t)n_slots, hipMemcpyHostToDevice, h->stream));
    TapScanArgs a;
    std::memset(&a.cu, 0, sizeof(a.cu));
    std::memset(&a.fwd, 0, sizeof(a.fwd));
    if (car_inv) a.cu = *car_inv;
    if (car_fwd) a.fwd = *car_fwd;
    a.hom = hom_dev ? hom_dev : h->lane_params.as<double>();
    a.n_slots = n_slots;
    a.skip = h->tap_skip.as<unsigned char>();
    a.ref = h->ref.p;
    a.ref_f32 = h->ref_dtype == COREG_F32 ? 1 : 0;
    a.gw = h->gW;
    a.gh = h->gH;
    a.wmax = (double)(h->sW - 1);
    a.hmax = (double)(h->sH - 1);
    a.tol = 1e-8;  // wcslib's round-trip noise stays below 1e-9 px, the homography's below 1e-11
    a.img = h->small.p;
    a.img_f32 = h->small_f32 ? 1 : 0;
    a.W = h->sW;
    a.H = h->sH;
    a.order = order;
    a.nan_filter = (int)h->opt_tap_nan_filter;
    a.bounds_only = (order & 1) ? 0 : 1;
    if (a.bounds_only) a.nan_filter = 0;  // (nothing read from the image to align: no join with its upload either)
    a.seg_list = h->tap_segq.as<uint4>();
    a.seg_count = h->tap_count.as<unsigned>() + 1;
    a.seg_cap = seg_cap;
    if (a.nan_filter) RETCHK(join_small(h));  // (the scan reads the image to align)
    a.count = h->tap_count.as<unsigned>();
    a.list = h->tap_list.as<uint2>();
    a.cap = cap;
    // the sweep's cull box (target pixels that can map into the image for some lag, 3 px of margin): [x0, x1, y0, y1]
    a.i_lo = (int)std::max(0.0, std::min((double)h->gW, box[0]));
    a.i_hi = (int)std::min((double)(h->gW - 1), std::max(-1.0, box[1]));
    a.j_lo = (int)std::max(0.0, std::min((double)h->gH, box[2]));
    a.j_hi = (int)std::min((double)(h->gH - 1), std::max(-1.0, box[3]));
    h->tap_last[0] = h->tap_last[1] = h->tap_last[2] = 0;
    if (a.i_hi < a.i_lo || a.j_hi < a.j_lo) return COREG_OK;
    const int n_rows = a.j_hi - a.j_lo + 1;
    const unsigned gx = (unsigned)((n_slots + 255) / 256);
    const unsigned gy = (unsigned)std::max(1, std::min(n_rows, (int)(4096 / std::max(1u, gx))));
    a.rows_per_block = (n_rows + (int)gy - 1) / (int)gy;
    const dim3 grid(gx, (unsigned)((n_rows + a.rows_per_block - 1) / a.rows_per_block));
    unsigned count = 0;
    for (int pass = 0; pass < 2; ++pass) {
        a.list = h->tap_list.as<uint2>();
        a.cap = cap;
        HIPCHK(hipMemsetAsync(h->tap_count.p, 0, 2 * sizeof(unsigned), h->stream));
        if (sweep_mode == MODE_CAR) {
            hipLaunchKernelGGL((k_tap_scan<MODE_CAR>), grid, dim3(256), 0, h->stream, a);
            hipLaunchKernelGGL((k_tap_scan_segments<MODE_CAR>), dim3(2048), dim3(256), 0, h->stream, a);
        } else if (sweep_mode == MODE_HOMOGRAPHY_SERIES) {
            hipLaunchKernelGGL((k_tap_scan<MODE_HOMOGRAPHY_SERIES>), grid, dim3(256), 0, h->stream, a);
            hipLaunchKernelGGL((k_tap_scan_segments<MODE_HOMOGRAPHY_SERIES>), dim3(2048), dim3(256), 0, h->stream, a);
        } else {
            hipLaunchKernelGGL((k_tap_scan<MODE_HOMOGRAPHY>), grid, dim3(256), 0, h->stream, a);
            hipLaunchKernelGGL((k_tap_scan_segments<MODE_HOMOGRAPHY>), dim3(2048), dim3(256), 0, h->stream, a);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&count, h->tap_count.p, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (count <= cap || count > cap_max) break;
        cap = count;  // (the scan is deterministic in what it lists: the second pass finds exactly `count` entries)
        HIPCHK(h->tap_list.reserve((size_t)cap * sizeof(uint2)));
    }
    h->tap_last[0] = count;
    h->tap_last[1] = 0;
    h->tap_last[2] = count > cap ? 1 : 0;
    if (count == 0 || count > cap) return COREG_OK;
    std::vector<uint2> list(count);
    HIPCHK(hipMemcpy(list.data(), h->tap_list.p, (size_t)count * sizeof(uint2), hipMemcpyDeviceToHost));
    // group by slot (counting sort), then every segment is put in pixel order by the thread that evaluates it: the
    // summation order of k_tap_fix does not depend on the order the scan's atomics happened to list the samples in
    std::vector<int> first((size_t)n_slots + 1, 0);
    for (unsigned k = 0; k < count; ++k) ++first[(size_t)list[k].x + 1];
    for (long long sl = 0; sl < n_slots; ++sl) first[(size_t)sl + 1] += first[(size_t)sl];
    std::vector<unsigned> pixel(count);
    {
        std::vector<int> at(first.begin(), first.end() - 1);
        for (unsigned k = 0; k < count; ++k) pixel[(size_t)at[list[k].x]++] = list[k].y;
    }
    std::vector<int> seg_slot, seg_begin;
    for (long long sl = 0; sl < n_slots; ++sl)
        if (first[(size_t)sl + 1] > first[(size_t)sl]) {
            seg_slot.push_back((int)sl);
            seg_begin.push_back(first[(size_t)sl]);
        }
    seg_begin.push_back((int)count);
    const int n_seg = (int)seg_slot.size();
    std::vector<double> xw(count), yw(count);
    WcslibTan wf;
    WcslibCar wfc;
    const bool car = target.proj == COREG_PROJ_CAR;
    if (car) wfc.init(target);
    else wf.init(target);
    const int gw = h->gW;
    auto work = [&](int s0, int s1) {
        for (int sg = s0; sg < s1; ++sg) {
            std::sort(pixel.begin() + seg_begin[sg], pixel.begin() + seg_begin[sg + 1]);
            if (car) {
                WcslibCar wt;
                wt.init(shifted_of(seg_slot[sg]));
                for (int e = seg_begin[sg]; e < seg_begin[sg + 1]; ++e)
                    wcslib_pixel_to_pixel(wfc, wt, (double)(pixel[e] % (unsigned)gw), (double)(pixel[e] / (unsigned)gw), &xw[e], &yw[e]);
            } else {
                WcslibTan wt;
                wt.init(shifted_of(seg_slot[sg]));
                for (int e = seg_begin[sg]; e < seg_begin[sg + 1]; ++e)
                    wcslib_pixel_to_pixel(wf, wt, (double)(pixel[e] % (unsigned)gw), (double)(pixel[e] / (unsigned)gw), &xw[e], &yw[e]);
            }
        }
    };
    unsigned nt = std::min<unsigned>(12, std::max(1u, std::thread::hardware_concurrency()));
    if (count < 4096 || n_seg < 2) nt = 1;
    if (nt <= 1) {
        work(0, n_seg);
    } else {
        // segments dealt in runs of about equal entry counts
        std::vector<std::thread> th;
        int s0 = 0;
        for (unsigned t = 0; t < nt && s0 < n_seg; ++t) {
            const long long want = (long long)count * (t + 1) / nt;
            int s1 = s0 + 1;
            while (s1 < n_seg && seg_begin[s1] < want) ++s1;
            if (t + 1 == nt) s1 = n_seg;
            th.emplace_back(work, s0, s1);
            s0 = s1;
        }
        for (auto& x : th) x.join();
    }
    HIPCHK(h->tap_seg_slot.reserve((size_t)n_seg * sizeof(int)));
    HIPCHK(h->tap_seg_begin.reserve((size_t)(n_seg + 1) * sizeof(int)));
    HIPCHK(h->tap_pixel.reserve((size_t)count * sizeof(unsigned)));
    HIPCHK(h->tap_xw.reserve((size_t)count * sizeof(double)));
    HIPCHK(h->tap_yw.reserve((size_t)count * sizeof(double)));
    // (pageable sources, rare path: blocking copies)
    HIPCHK(hipMemcpy(h->tap_seg_slot.p, seg_slot.data(), (size_t)n_seg * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->tap_seg_begin.p, seg_begin.data(), (size_t)(n_seg + 1) * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->tap_pixel.p, pixel.data(), (size_t)count * sizeof(unsigned), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->tap_xw.p, xw.data(), (size_t)count * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->tap_yw.p, yw.data(), (size_t)count * sizeof(double), hipMemcpyHostToDevice));
    fix->tap_segs = n_seg;
    fix->tap_count = (long long)count;
    fix->tap_mode = sweep_mode;
    fix->tap.seg_slot = h->tap_seg_slot.as<int>();
    fix->tap.seg_begin = h->tap_seg_begin.as<int>();
    fix->tap.pixel = h->tap_pixel.as<unsigned>();
    fix->tap.xw = h->tap_xw.as<double>();
    fix->tap.yw = h->tap_yw.as<double>();
    fix->tap.cu = a.cu;
    fix->tap.fwd = a.fwd;
    h->tap_last[1] = n_seg;
    return COREG_OK;
}

}  // namespace
