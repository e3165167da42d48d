// Part of libcoreg_hip.so's ONE translation unit (coreg_hip.hip includes the parts in order; round 6 split by concern,
// no behaviour change): planning of a sweep: lag dimensions and checks, tile shape + lag patch, lag slots, tile groups, precompute launch, LDS pitch.
#pragma once
namespace {
// ---- lag batching ---------------------------------------------------------------------------------------------
struct LagDims {
    int n1, n2, n3, n4, n5;
    long long nc;  // (cdelt1, cdelt2, crota) combinations this sweep covers: n3*n4*n5, or the "combo_begin/_end" range
    long long c0;  // first of them in the lag set's inner C-order index ((i3 * n4 + i4) * n5 + i5)
    long long total() const { return (long long)n1 * n2 * nc; }
    void inner(long long c, int* i3, int* i4, int* i5) const {  // c in [0, nc)
        const long long g = c + c0;
        *i5 = (int)(g % n5);
        *i4 = (int)((g / n5) % n4);
        *i3 = (int)(g / ((long long)n5 * n4));
    }
};

// The one-shot combination range ("combo_begin" / "combo_end") belongs to THE NEXT sweep call, whatever becomes of it: the
// entry points take it off the handle before any validation (a call that failed early used to leave it armed
// for an unrelated later sweep) and hand it to check_lags.
struct ComboRange {
    long long begin = 0, end = 0;
};
ComboRange take_combo_range(coreg_handle* h) {
    ComboRange r;
    r.begin = h->opt_combo_begin;
    r.end = h->opt_combo_end;
    h->opt_combo_begin = h->opt_combo_end = 0;
    return r;
}
int check_lags(coreg_handle* h, const coreg_lags* l, LagDims* d, int64_t begin, int64_t end, ComboRange combo = ComboRange()) {
    if (!l || !l->crval1 || !l->crval2 || !l->cdelt1 || !l->cdelt2 || !l->crota)
        return fail(h, COREG_EINVAL, "lags: null array");
    if (l->n_crval1 < 1 || l->n_crval2 < 1 || l->n_cdelt1 < 1 || l->n_cdelt2 < 1 || l->n_crota < 1)
        return fail(h, COREG_EINVAL, "lags: every axis needs at least one value");
    d->n1 = l->n_crval1;
    d->n2 = l->n_crval2;
    d->n3 = l->n_cdelt1;
    d->n4 = l->n_cdelt2;
    d->n5 = l->n_crota;
    d->nc = (long long)d->n3 * d->n4 * d->n5;
    d->c0 = 0;
    {
        // one-shot combination range of a multi-GPU sweep (taken off the handle by the entry point)
        const long long cb = combo.begin, ce = combo.end;
        if (cb != 0 || ce != 0) {
            if (cb < 0 || ce <= cb || ce > d->nc)
                return fail(h, COREG_EINVAL, "combo_begin/combo_end outside [0, n_cdelt1 * n_cdelt2 * n_crota]");
            d->c0 = cb;
            d->nc = ce - cb;
        }
    }
    if (begin < 0 || end > d->total() || begin > end)
        return fail(h, COREG_EINVAL, "lag_begin/lag_end outside [0, n_lags]");
    const double* ax[5] = {l->crval1, l->crval2, l->cdelt1, l->cdelt2, l->crota};
    const int32_t na[5] = {l->n_crval1, l->n_crval2, l->n_cdelt1, l->n_cdelt2, l->n_crota};
    for (int k = 0; k < 5; ++k)
        for (int32_t i = 0; i < na[k]; ++i)
            if (!std::isfinite(ax[k][i])) return fail(h, COREG_EINVAL, "lags: a non-finite value");
    return COREG_OK;
}

// Headers / grids that cannot give finite pixel coordinates are refused before anything is planned or launched
// (geometry.hpp wcs_problem: the reference would hand such a header to astropy, which raises, or return NaN everywhere).
static int check_wcs(coreg_handle* h, const coreg_wcs2d* w, bool carrington_transform) {
    const char* why = wcs_problem(*w, carrington_transform);
    return why ? fail(h, COREG_EINVAL, why) : COREG_OK;
}
// Pixel and grid-point counts are 32-bit in the kernels' lists (active points, border pixels, tiles): refuse what does not fit.
static bool too_many(long long a, long long b) { return a * b > 2147483647ll; }
static int check_grid(coreg_handle* h, const coreg_carr_grid* g) {
    if (g->n_lon >= 1 && g->n_lat >= 1 && too_many(g->n_lon, g->n_lat))
        return fail(h, COREG_EINVAL, "Carrington grid: more than 2^31 - 1 points");
    const char* why = grid_problem(*g);
    return why ? fail(h, COREG_EINVAL, why) : COREG_OK;
}

// Local pixel-space geometry of a sweep (host estimates; they steer the plan, never the results):
// small-image pixels per grid step (d?_di, d?_dj) and per CRVAL1 / CRVAL2 lag step (a?, b?).
struct Geometry {
    double dx_di = 1, dx_dj = 0, dy_di = 0, dy_dj = 1;
    double ax = 0, ay = 0, bx = 0, by = 0;
};
struct Plan {
    int tile_w = 32;      // grid tile = tile_w x (kTilePts / tile_w) points
    int sw = 16, sh = 16; // lag patch of a workgroup: sw CRVAL1 lags x sh CRVAL2 lags (sw * sh <= 256)
    double window = 0;    // estimated LDS window (elements)
    double win_w = 0, win_h = 0;  // its estimated width / height in pixels
};

// Tile shape and lag patch chosen together: fewest lag batches (= least padded lane slots) among the combinations
// whose LDS window (tile extent (+) patch extent, in pixels) fits; ties -> smaller window.  m1 x n2 = lag plane.
Plan choose_plan(coreg_handle* h, const Geometry& g, int m1, int n2, long long lds_elems) {
    Plan best, fallback;
    double best_cost = std::numeric_limits<double>::max(), fb_win = std::numeric_limits<double>::max();
    const int sw_hi = h->opt_patch_w > 0 ? std::min<int>((int)h->opt_patch_w, kBlock) : kBlock;
    for (int tw = 4; tw <= 256; tw *= 2) {
        if (h->opt_tile_w > 0 && tw != h->opt_tile_w) continue;
        const int th = kTilePts / tw;
        if (th < 1) continue;
        const double tex = tw * std::fabs(g.dx_di) + th * std::fabs(g.dx_dj);
        const double tey = tw * std::fabs(g.dy_di) + th * std::fabs(g.dy_dj);
        for (int sw = 1; sw <= std::min(m1, sw_hi); ++sw) {
            int sh = std::min(n2, kBlock / sw);
            if (sh < 1) break;
            const int cols = (m1 + sw - 1) / sw, rows = (n2 + sh - 1) / sh;
            sh = (n2 + rows - 1) / rows;                    // smallest sh with the same batch count
            const int sw2 = (m1 + cols - 1) / cols;         // likewise for sw
            const double ex = tex + sw2 * std::fabs(g.ax) + sh * std::fabs(g.bx) + 6.0;
            const double ey = tey + sw2 * std::fabs(g.ay) + sh * std::fabs(g.by) + 6.0;
            const double win = (ex + 1.0) * ey;
            if (win < fb_win) {
                fb_win = win;
                fallback.tile_w = tw;
                fallback.sw = sw2;
                fallback.sh = sh;
                fallback.window = win;
                fallback.win_w = ex + 1.0;
                fallback.win_h = ey;
            }
            if (win > 0.94 * (double)lds_elems) continue;
            const double cost = (double)cols * rows * (1.0 + 0.08 * win / (double)lds_elems);
            if (cost < best_cost) {
                best_cost = cost;
                best.tile_w = tw;
                best.sw = sw2;
                best.sh = sh;
                best.window = win;
                best.win_w = ex + 1.0;
                best.win_h = ey;
            }
        }
    }
    const Plan& r = best_cost < std::numeric_limits<double>::max() ? best : fallback;
    if (std::getenv("COREG_DEBUG_PLAN"))
        std::fprintf(stderr, "[coreg plan] tile %d x %d, lag patch %d x %d, window estimate %.0f (%.1f x %.1f) of %lld elements; "
                     "geometry d/di (%.3f, %.3f) d/dj (%.3f, %.3f) lag1 (%.3f, %.3f) lag2 (%.3f, %.3f)\n", r.tile_w,
                     kTilePts / r.tile_w, r.sw, r.sh, r.window, r.win_w, r.win_h, lds_elems, g.dx_di, g.dy_di, g.dx_dj, g.dy_dj, g.ax, g.ay,
                     g.bx, g.by);
    return r;
}

// indices of the smallest, the most central and the largest value of a lag axis (any order, NaNs ignored)
void extreme_lags(const double* v, int n, int out[3]) {
    int lo = 0, hi = 0;
    for (int i = 1; i < n; ++i) {
        if (v[i] < v[lo] || v[lo] != v[lo]) lo = i;
        if (v[i] > v[hi] || v[hi] != v[hi]) hi = i;
    }
    const double mid = 0.5 * (v[lo] + v[hi]);
    int m = lo;
    for (int i = 0; i < n; ++i)
        if (std::fabs(v[i] - mid) < std::fabs(v[m] - mid)) m = i;
    out[0] = lo;
    out[1] = m;
    out[2] = hi;
}

// mean spacing of a lag axis over its value range (plan heuristics only; lists may come in any order)
double lag_step(const double* v, int n) {
    if (n < 2) return 0.0;
    double lo = v[0], hi = v[0];
    for (int i = 1; i < n; ++i) {
        lo = std::min(lo, v[i]);
        hi = std::max(hi, v[i]);
    }
    return (hi - lo) / (double)(n - 1);
}

struct SlotList {
    std::vector<int> i1, i2;        // lag indices supplying the lane parameters (clamped for padding)
    std::vector<long long> outidx;  // raveled C-order lag index or -1
    int n_batches = 0;
};

// slots for combo c (= (i3*n4 + i4)*n5 + i5) restricted to the raveled slice [begin, end)
void build_slots(const LagDims& d, long long c, long long begin, long long end, int sw, int sh, SlotList* s) {
    s->i1.clear();
    s->i2.clear();
    s->outidx.clear();
    s->n_batches = 0;
    {
        const size_t cap = (size_t)((d.n1 + sw - 1) / sw + 1) * ((d.n2 + sh - 1) / sh) * kBlock;
        s->i1.reserve(cap);
        s->i2.reserve(cap);
        s->outidx.reserve(cap);
    }
    const long long row = (long long)d.n2 * d.nc;
    const int i1_lo = (int)(begin / row);
    const int i1_hi = (int)((end - 1) / row);
    const int m1 = i1_hi - i1_lo + 1;
    for (int p1 = 0; p1 * sw < m1; ++p1)
        for (int p2 = 0; p2 * sh < d.n2; ++p2) {
            bool any = false;
            const size_t at = s->i1.size();
            for (int t = 0; t < kBlock; ++t) {
                int lx = t % sw, ly = t / sw;
                bool valid = ly < sh;
                if (!valid) lx = ly = 0;
                int i1 = i1_lo + p1 * sw + lx, i2 = p2 * sh + ly;
                if (i1 > i1_hi) {
                    i1 = i1_hi;
                    valid = false;
                }
                if (i2 > d.n2 - 1) {
                    i2 = d.n2 - 1;
                    valid = false;
                }
                const long long idx = ((long long)i1 * d.n2 + i2) * d.nc + c;
                if (idx < begin || idx >= end) valid = false;
                s->i1.push_back(i1);
                s->i2.push_back(i2);
                s->outidx.push_back(valid ? idx : -1);
                any |= valid;
            }
            if (!any) {
                s->i1.resize(at);
                s->i2.resize(at);
                s->outidx.resize(at);
            } else {
                s->n_batches++;
            }
        }
}

int pick_groups(coreg_handle* h, int n_batches, int n_tiles) {
    // the compacted points are cut in n_groups EQUAL shares (k_tile_list), so n_groups * n_batches workgroups of equal
    // work: 256 groups make every round of 256 CUs full; fewer when there are many lag batches
    if (h->opt_shard_world > 1) {
        // point sharding: every rank takes a multiple of 8 groups (the XCD-aware block mapping of k_sweep), as close
        // to one full round of 256 workgroups as the batch count allows
        const long long per_rank = 8 * std::max<long long>(1, std::llround(256.0 / (8.0 * n_batches)));
        return (int)std::min<long long>(per_rank * h->opt_shard_world, 1000 / (8 * h->opt_shard_world) * 8 * h->opt_shard_world);
    }
    long long g = h->opt_n_groups > 0 ? h->opt_n_groups : (4096 + n_batches - 1) / n_batches;
    (void)n_tiles;
    g = std::max<long long>(8, std::min<long long>(h->opt_n_groups > 0 ? 1000 : 256, ((g + 7) / 8) * 8));
    return (int)g;
}

// Tapered shares (kernels.hpp group_start) pay when the launch has many rounds of workgroups; with few rounds the large
// early shares would simply finish last (measured on the translation sweep, profiles/taper_sweep.sh: -3.5 % at 15
// rounds, -4 % at 8, about even at 4, +14 % at 2) ...
void pick_taper(const coreg_handle* h, int n_groups, int n_batches, int* tmin, int* tfrac) {
    *tmin = (int)h->opt_taper_min;
    if (h->opt_taper_frac >= 0) {
        *tfrac = (int)h->opt_taper_frac;
        return;
    }
    // ... and when the groups are many (a fine taper) -- cfg4's 16 groups of a 156-tile grid lost 27 % to it, the
    // 4-round plate-carree launch 20 %
    const double rounds = (double)n_groups * n_batches / 256.0;
    *tfrac = (rounds >= (double)h->opt_taper_rounds && n_groups >= 128 && h->opt_shard_world <= 1) ? 512 : 0;
}

template <int MODE>
int launch_precompute(coreg_handle* h, const PrecomputeArgs& a, int n_tiles, int n_groups, int n_batches) {
    EventPair* ev = next_event(h, h->ev_pre, h->ev_pre_used);
    if (!ev) return fail(h, COREG_EHIP, "hipEventCreate failed");
    int tmin, tfrac;
    pick_taper(h, n_groups, n_batches, &tmin, &tfrac);
    PrecomputeArgs b = a;
    b.prologue = h->pending_prologue;  // (the first launch after upload_plan carries the sweep's prologue)
    const bool with_prologue = b.prologue.src != nullptr;
    std::memset(&h->pending_prologue, 0, sizeof(h->pending_prologue));
    HIPCHK(hipEventRecord(ev->a, h->stream));
    if (h->ref_dtype == COREG_F32)
        hipLaunchKernelGGL((k_precompute<MODE, float>), dim3(n_tiles), dim3(256), 0, h->stream, b);
    else
        hipLaunchKernelGGL((k_precompute<MODE, double>), dim3(n_tiles), dim3(256), 0, h->stream, b);
    (void)with_prologue;
    hipLaunchKernelGGL(k_tile_list, dim3(1), dim3(1024), 0, h->stream, (const int*)a.tile_count, n_tiles, n_groups,
                       h->tile_list.as<int>(), h->tile_cum.as<int>(), h->group_first.as<int>(),
                       h->tile_info.as<long long>(), tmin, tfrac);
    // (no closing event: the sweep launch that follows opens with one, and that is where this interval ends --
    // collect_stats; one marker packet less between the kernels of a sweep)
    ev->b_is_next_sweep = true;
    ev->next_sweep_index = h->ev_sweep_used;
    HIPCHK(hipGetLastError());
    return COREG_OK;
}

int reserve_tiles(coreg_handle* h, int n_tiles) {
    // + an explicit tail: the rolling scalar prefetch of tile_points reads up to kPointGroups chunks past a tile's last
    // chunk (never used), which for the last tile is past the requested size whatever capacity an earlier, larger
    // reservation left
    const size_t pts = (size_t)n_tiles * kTilePts + (size_t)(kPointGroups + 1) * kChunk;
    HIPCHK(h->pts.reserve(pts * sizeof(Pt)));
    HIPCHK(h->tile_count.reserve(n_tiles * sizeof(int)));
    HIPCHK(h->tile_list.reserve(n_tiles * sizeof(int)));
    HIPCHK(h->tile_cum.reserve((n_tiles + 1) * sizeof(int)));
    HIPCHK(h->group_first.reserve(2 * 1024 * sizeof(int) + 64));  // [0, 1024): first list entry; [1024, ...): first unit
    HIPCHK(h->tile_info.reserve(8 * sizeof(long long)));
    HIPCHK(h->tile_bbox.reserve((size_t)n_tiles * 4 * sizeof(double)));
    return COREG_OK;
}

void fill_precompute_common(coreg_handle* h, PrecomputeArgs* a, int tile_w) {
    a->ref = h->ref.p;
    a->gw = h->gW;
    a->gh = h->gH;
    a->tile_w = tile_w;
    a->tile_h = kTilePts / tile_w;
    a->tiles_x = (h->gW + a->tile_w - 1) / a->tile_w;
    a->tiles_y = (h->gH + a->tile_h - 1) / a->tile_h;
    a->pivot_a = h->pivots.as<double>();
    a->pts = h->pts.as<Pt>();
    a->tile_count = h->tile_count.as<int>();
    a->tile_bbox = h->tile_bbox.as<double>();
}

// one sweep-kernel launch + finalize over n_batches * 256 slots whose parameters (SoA [np][n_slots]) and output
// indices are already on the device
long long lds_window_elems(const coreg_handle* h);

// Compile-time LDS window pitch for the Carrington order-2 sweep: the smallest instantiated pitch that holds the planned
// window (width + slack) within the LDS.  All of them are 25 mod 32, the residue that spreads the ~2 px lag lattice best
// over the 32 bank pairs in the conflict simulation.  0 = pitch chosen per visit.
int pick_pitch(const coreg_handle* h, const Plan& plan, long long lds_elems, int order = 2) {
    if (h->opt_pitch == 0 || !h->opt_use_lds) return 0;
    static const int kPitches[] = {89, 121, 153, 185, 217};
    if (h->opt_pitch > 0) {
        for (int p : kPitches)
            if (p == h->opt_pitch) return p;
        return 0;
    }
    for (int p : kPitches)
        if ((double)p >= plan.win_w + (order > 2 ? 4.0 : 2.0) &&
            (double)p * (plan.win_h + (order > 2 ? 2.0 : 0.0)) <= 0.985 * (double)lds_elems)
            return p;
    return 0;
}

}  // namespace
