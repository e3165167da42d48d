// Host-side sub-lag refinement of the correlation peak: the bounded non-linear least-squares fit of a 2-D Gaussian
// that AlignmentResults._compute_shift delegates to scipy (hdrshift/AlignmentResults.py:12-21 the model, :218-341 the
// call `curve_fit(f=twoD_Gaussian, xdata, ydata, p0, bounds)`).  curve_fit with bounds is
// scipy.optimize.least_squares(method='trf', jac='2-point', x_scale=1, ftol = xtol = gtol = 1e-8, max_nfev = 100 n):
// third-party code (scipy, pinned 1.17.1; 1.15.3 in the build container), absent from the reference tree.  This file
// restates its published algorithm -- Branch, Coleman & Li's trust-region reflective method with the exact (SVD)
// trust-region solver of More, the Coleman-Li scaling, the reflected / Cauchy step selection, the forward-difference
// Jacobian with scipy's step rule, and the same termination tests -- so that the fit follows the SAME iteration path
// and stops at the SAME iterate as the reference's call (scipy stops well short of the mathematical minimum on the flat
// peaks of a correlation map: up to 1e-3 px, so "a" least-squares solver would not do).  No GPU work, no dependency.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace coregfit {

constexpr int N = 6;      // amplitude, xo, yo, sigma_x, sigma_y, offset
constexpr int MMAX = 64;  // at most 26 points are ever handed over (5 x 5 neighbours + the duplicated peak)

struct Problem {
    int m;
    const double *x, *y, *z;
};

// twoD_Gaussian(xy, *p) - z, operation for operation (AlignmentResults.py:12-21; curve_fit subtracts ydata)
static inline void residuals(const Problem& P, const double* p, double* f) {
    const double a = p[0], x0 = p[1], y0 = p[2], sx = p[3], sy = p[4], off = p[5];
    const double dx2 = 2 * (sx * sx), dy2 = 2 * (sy * sy);
    for (int i = 0; i < P.m; ++i) {
        const double ex = P.x[i] - x0, ey = P.y[i] - y0;
        f[i] = (off + a * std::exp(-(((ex * ex) / dx2) + ((ey * ey) / dy2)))) - P.z[i];
    }
}

static inline bool all_finite(const double* v, int n) {
    for (int i = 0; i < n; ++i)
        if (!std::isfinite(v[i])) return false;
    return true;
}
static inline double dot(const double* a, const double* b, int n) {
    double s = 0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}
static inline double norm2(const double* a, int n) { return std::sqrt(dot(a, a, n)); }

// scipy _numdiff.approx_derivative(method='2-point', rel_step=None, bounds): h = sqrt(eps) * sign(x) * max(1, |x|),
// flipped or shrunk where x + h leaves [lb, ub] (_adjust_scheme_to_bounds, '1-sided'); J[:, i] = (f(x + h e_i) - f0) / dx
static void jac_2point(const Problem& P, const double* x, const double* f0, const double* lb, const double* ub,
                       double* J /* m x N, row-major */) {
    const double rstep = std::sqrt(std::numeric_limits<double>::epsilon());
    double f1[MMAX], xp[N];
    for (int i = 0; i < N; ++i) {
        double h = rstep * (x[i] >= 0 ? 1.0 : -1.0) * std::max(1.0, std::fabs(x[i]));
        const double lower = x[i] - lb[i], upper = ub[i] - x[i];
        const double xt = x[i] + h;
        const bool violated = (xt < lb[i]) || (xt > ub[i]);
        const bool fitting = std::fabs(h) <= std::max(lower, upper);
        if (violated && fitting) h = -h;
        else if (!fitting) h = (upper >= lower) ? upper : -lower;
        std::memcpy(xp, x, sizeof(xp));
        xp[i] = x[i] + h;
        const double dx = xp[i] - x[i];
        residuals(P, xp, f1);
        for (int r = 0; r < P.m; ++r) J[r * N + i] = (f1[r] - f0[r]) / dx;
    }
}

static void jac_analytic(const Problem& P, const double* p, double* J) {
    const double a = p[0], x0 = p[1], y0 = p[2], sx = p[3], sy = p[4];
    for (int i = 0; i < P.m; ++i) {
        const double ex = P.x[i] - x0, ey = P.y[i] - y0;
        const double e = std::exp(-(((ex * ex) / (2 * (sx * sx))) + ((ey * ey) / (2 * (sy * sy)))));
        double* r = J + i * N;
        r[0] = e;
        r[1] = a * e * ex / (sx * sx);
        r[2] = a * e * ey / (sy * sy);
        r[3] = a * e * ex * ex / (sx * sx * sx);
        r[4] = a * e * ey * ey / (sy * sy * sy);
        r[5] = 1.0;
    }
}

// thin SVD of an (rows x N) matrix by one-sided Jacobi rotations (Hestenes): A = U diag(s) V^T, s descending.
// U overwrites A (rows x N); 32 x 6 at most here, where Jacobi is both the simplest and the most accurate choice.
static void svd_jacobi(double* A, int rows, double* s, double* V /* N x N row-major */) {
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) V[i * N + j] = (i == j);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < N - 1; ++p)
            for (int q = p + 1; q < N; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int r = 0; r < rows; ++r) {
                    const double ap = A[r * N + p], aq = A[r * N + q];
                    al += ap * ap, be += aq * aq, ga += ap * aq;
                }
                if (ga == 0.0) continue;
                const double lim = std::sqrt(al * be);
                if (std::fabs(ga) <= 1e-17 * lim) continue;
                off = std::max(off, std::fabs(ga) / (lim > 0 ? lim : 1.0));
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
                for (int r = 0; r < rows; ++r) {
                    const double ap = A[r * N + p], aq = A[r * N + q];
                    A[r * N + p] = c * ap - sn * aq;
                    A[r * N + q] = sn * ap + c * aq;
                }
                for (int r = 0; r < N; ++r) {
                    const double vp = V[r * N + p], vq = V[r * N + q];
                    V[r * N + p] = c * vp - sn * vq;
                    V[r * N + q] = sn * vp + c * vq;
                }
            }
        if (off < 1e-16) break;
    }
    int order[N];
    for (int j = 0; j < N; ++j) {
        double n2 = 0;
        for (int r = 0; r < rows; ++r) n2 += A[r * N + j] * A[r * N + j];
        s[j] = std::sqrt(n2);
        order[j] = j;
    }
    std::sort(order, order + N, [&](int a, int b) { return s[a] > s[b]; });
    double As[(MMAX + N) * N], Vs[N * N], ss[N];
    for (int k = 0; k < N; ++k) {
        const int j = order[k];
        ss[k] = s[j];
        const double inv = s[j] > 0 ? 1.0 / s[j] : 0.0;
        for (int r = 0; r < rows; ++r) As[r * N + k] = A[r * N + j] * inv;
        for (int r = 0; r < N; ++r) Vs[r * N + k] = V[r * N + j];
    }
    std::memcpy(A, As, sizeof(double) * rows * N);
    std::memcpy(V, Vs, sizeof(Vs));
    std::memcpy(s, ss, sizeof(ss));
}

// --- scipy/optimize/_lsq/common.py, restated -------------------------------------------------------------------------
static double step_size_to_bound(const double* x, const double* s, const double* lb, const double* ub, int* hits) {
    double steps[N], mn = INFINITY;
    for (int i = 0; i < N; ++i) {
        steps[i] = INFINITY;
        if (s[i] != 0) steps[i] = std::max((lb[i] - x[i]) / s[i], (ub[i] - x[i]) / s[i]);
        mn = std::min(mn, steps[i]);
    }
    if (hits)
        for (int i = 0; i < N; ++i) hits[i] = (steps[i] == mn) ? (s[i] > 0) - (s[i] < 0) : 0;
    return mn;
}
static bool in_bounds(const double* x, const double* lb, const double* ub) {
    for (int i = 0; i < N; ++i)
        if (!(x[i] >= lb[i] && x[i] <= ub[i])) return false;
    return true;
}
static void make_strictly_feasible(double* x, const double* lb, const double* ub, double rstep) {
    for (int i = 0; i < N; ++i) {
        const double xi = x[i];
        int active = 0;
        if (rstep == 0) {
            if (xi <= lb[i]) active = -1;
            if (xi >= ub[i]) active = 1;
        } else {
            const double ld = xi - lb[i], ud = ub[i] - xi;
            const double lt = rstep * std::max(1.0, std::fabs(lb[i])), ut = rstep * std::max(1.0, std::fabs(ub[i]));
            if (std::isfinite(lb[i]) && ld <= std::min(ud, lt)) active = -1;
            if (std::isfinite(ub[i]) && ud <= std::min(ld, ut)) active = 1;
        }
        if (active == -1) x[i] = rstep == 0 ? std::nextafter(lb[i], ub[i]) : lb[i] + rstep * std::max(1.0, std::fabs(lb[i]));
        if (active == 1) x[i] = rstep == 0 ? std::nextafter(ub[i], lb[i]) : ub[i] - rstep * std::max(1.0, std::fabs(ub[i]));
        if (x[i] < lb[i] || x[i] > ub[i]) x[i] = 0.5 * (lb[i] + ub[i]);
    }
}
static void cl_scaling(const double* x, const double* g, const double* lb, const double* ub, double* v, double* dv) {
    for (int i = 0; i < N; ++i) {
        v[i] = 1, dv[i] = 0;
        if (g[i] < 0 && std::isfinite(ub[i])) v[i] = ub[i] - x[i], dv[i] = -1;
        if (g[i] > 0 && std::isfinite(lb[i])) v[i] = x[i] - lb[i], dv[i] = 1;
    }
}
static void matvec(const double* J, int m, const double* s, double* out) {
    for (int r = 0; r < m; ++r) out[r] = dot(J + r * N, s, N);
}
static double evaluate_quadratic(const double* Jh, int m, const double* gh, const double* s, const double* diag) {
    double Js[MMAX];
    matvec(Jh, m, s, Js);
    double q = dot(Js, Js, m);
    for (int i = 0; i < N; ++i) q += s[i] * diag[i] * s[i];
    return 0.5 * q + dot(s, gh, N);
}
static void build_quadratic_1d(const double* Jh, int m, const double* gh, const double* s, const double* diag,
                               const double* s0, double* a, double* b, double* c) {
    double v[MMAX];
    matvec(Jh, m, s, v);
    double aa = dot(v, v, m);
    for (int i = 0; i < N; ++i) aa += s[i] * diag[i] * s[i];
    aa *= 0.5;
    double bb = dot(gh, s, N), cc = 0;
    if (s0) {
        double u[MMAX];
        matvec(Jh, m, s0, u);
        bb += dot(u, v, m);
        cc = 0.5 * dot(u, u, m) + dot(gh, s0, N);
        for (int i = 0; i < N; ++i) bb += s0[i] * diag[i] * s[i], cc += 0.5 * s0[i] * diag[i] * s0[i];
    }
    *a = aa, *b = bb;
    if (c) *c = cc;
}
static void minimize_quadratic_1d(double a, double b, double lo, double hi, double c, double* t_out, double* y_out) {
    double t[3] = {lo, hi, 0};
    int nt = 2;
    if (a != 0) {
        const double ext = -0.5 * b / a;
        if (lo < ext && ext < hi) t[nt++] = ext;
    }
    int best = 0;
    double yb = 0;
    for (int k = 0; k < nt; ++k) {
        const double y = t[k] * (a * t[k] + b) + c;
        if (k == 0 || y < yb) yb = y, best = k;  // np.argmin: first minimum
    }
    *t_out = t[best], *y_out = yb;
}
static double intersect_trust_region_pos(const double* x, const double* s, double Delta) {
    const double a = dot(s, s, N), b = dot(x, s, N), c = dot(x, x, N) - Delta * Delta;
    const double d = std::sqrt(b * b - a * c);
    const double q = -(b + std::copysign(d, b));
    const double t1 = q / a, t2 = c / q;
    return t1 < t2 ? t2 : t1;
}
static void solve_lsq_trust_region(int m, const double* uf, const double* s, const double* V, double Delta,
                                   double* alpha_io, double* p) {
    const double EPS = std::numeric_limits<double>::epsilon();
    double suf[N];
    for (int i = 0; i < N; ++i) suf[i] = s[i] * uf[i];
    const bool full_rank = (m >= N) && (s[N - 1] > EPS * m * s[0]);
    auto apply = [&](const double* w) {  // p = -V w
        for (int i = 0; i < N; ++i) {
            double acc = 0;
            for (int j = 0; j < N; ++j) acc += V[i * N + j] * w[j];
            p[i] = -acc;
        }
    };
    if (full_rank) {
        double w[N];
        for (int i = 0; i < N; ++i) w[i] = uf[i] / s[i];
        apply(w);
        if (norm2(p, N) <= Delta) {
            *alpha_io = 0.0;
            return;
        }
    }
    auto phi_dphi = [&](double alpha, double* phi, double* dphi) {
        double t[N], sum3 = 0;
        for (int i = 0; i < N; ++i) {
            const double den = s[i] * s[i] + alpha;
            t[i] = suf[i] / den;
            sum3 += suf[i] * suf[i] / (den * den * den);
        }
        const double pn = norm2(t, N);
        *phi = pn - Delta;
        *dphi = -sum3 / pn;
    };
    double alpha_upper = norm2(suf, N) / Delta, alpha_lower = 0.0;
    if (full_rank) {
        double phi, dphi;
        phi_dphi(0.0, &phi, &dphi);
        alpha_lower = -phi / dphi;
    }
    double alpha = *alpha_io;
    if (!full_rank && alpha == 0) alpha = std::max(0.001 * alpha_upper, std::sqrt(alpha_lower * alpha_upper));
    for (int it = 0; it < 10; ++it) {
        if (alpha < alpha_lower || alpha > alpha_upper)
            alpha = std::max(0.001 * alpha_upper, std::sqrt(alpha_lower * alpha_upper));
        double phi, dphi;
        phi_dphi(alpha, &phi, &dphi);
        if (phi < 0) alpha_upper = alpha;
        const double ratio = phi / dphi;
        alpha_lower = std::max(alpha_lower, alpha - ratio);
        alpha -= (phi + Delta) * ratio / Delta;
        if (std::fabs(phi) < 0.01 * Delta) break;
    }
    double w[N];
    for (int i = 0; i < N; ++i) w[i] = suf[i] / (s[i] * s[i] + alpha);
    apply(w);
    const double sc = Delta / norm2(p, N);
    for (int i = 0; i < N; ++i) p[i] *= sc;
    *alpha_io = alpha;
}

// scipy/optimize/_lsq/trf.py: select_step
static double select_step(const double* x, const double* Jh, int m, const double* diag_h, const double* g_h, double* p,
                          double* p_h, const double* d, double Delta, const double* lb, const double* ub, double theta,
                          double* step, double* step_h) {
    double xp[N];
    for (int i = 0; i < N; ++i) xp[i] = x[i] + p[i];
    if (in_bounds(xp, lb, ub)) {
        std::memcpy(step, p, sizeof(double) * N);
        std::memcpy(step_h, p_h, sizeof(double) * N);
        return -evaluate_quadratic(Jh, m, g_h, p_h, diag_h);
    }
    int hits[N];
    const double p_stride = step_size_to_bound(x, p, lb, ub, hits);
    double r_h[N], r[N], x_on_bound[N];
    for (int i = 0; i < N; ++i) {
        r_h[i] = hits[i] ? -p_h[i] : p_h[i];
        r[i] = d[i] * r_h[i];
        p[i] *= p_stride, p_h[i] *= p_stride;
        x_on_bound[i] = x[i] + p[i];
    }
    const double to_tr0 = intersect_trust_region_pos(p_h, r_h, Delta);
    const double to_bound0 = step_size_to_bound(x_on_bound, r, lb, ub, nullptr);
    double r_stride = std::min(to_bound0, to_tr0), r_stride_l, r_stride_u;
    if (r_stride > 0) {
        r_stride_l = (1 - theta) * p_stride / r_stride;
        r_stride_u = (r_stride == to_bound0) ? theta * to_bound0 : to_tr0;
    } else {
        r_stride_l = 0, r_stride_u = -1;
    }
    double r_value = INFINITY;
    if (r_stride_l <= r_stride_u) {
        double a, b, c;
        build_quadratic_1d(Jh, m, g_h, r_h, diag_h, p_h, &a, &b, &c);
        minimize_quadratic_1d(a, b, r_stride_l, r_stride_u, c, &r_stride, &r_value);
        for (int i = 0; i < N; ++i) {
            r_h[i] = r_h[i] * r_stride + p_h[i];
            r[i] = r_h[i] * d[i];
        }
    }
    for (int i = 0; i < N; ++i) p[i] *= theta, p_h[i] *= theta;
    const double p_value = evaluate_quadratic(Jh, m, g_h, p_h, diag_h);
    double ag_h[N], ag[N];
    for (int i = 0; i < N; ++i) ag_h[i] = -g_h[i], ag[i] = d[i] * ag_h[i];
    const double to_tr = Delta / norm2(ag_h, N);
    const double to_bound = step_size_to_bound(x, ag, lb, ub, nullptr);
    double ag_stride = (to_bound < to_tr) ? theta * to_bound : to_tr, ag_value, a, b;
    build_quadratic_1d(Jh, m, g_h, ag_h, diag_h, nullptr, &a, &b, nullptr);
    minimize_quadratic_1d(a, b, 0, ag_stride, 0, &ag_stride, &ag_value);
    for (int i = 0; i < N; ++i) ag_h[i] *= ag_stride, ag[i] *= ag_stride;
    const double *sel = ag, *sel_h = ag_h;
    double val = ag_value;
    if (p_value < r_value && p_value < ag_value) sel = p, sel_h = p_h, val = p_value;
    else if (r_value < p_value && r_value < ag_value) sel = r, sel_h = r_h, val = r_value;
    std::memcpy(step, sel, sizeof(double) * N);
    std::memcpy(step_h, sel_h, sizeof(double) * N);
    return -val;
}

// status: scipy's (1 gtol, 2 ftol, 3 xtol, 4 both, 0 max_nfev reached), -1: residuals not finite at p0, -2 bad input
static int fit(const Problem& P, const double* p0, const double* lb, const double* ub, int analytic_jac, double ftol,
               double xtol, double gtol, int max_nfev, double* popt, int* nfev_out, double* cost_out) {
    const int m = P.m;
    if (m < 1 || m > MMAX) return -2;
    for (int i = 0; i < N; ++i)
        if (!(lb[i] < ub[i]) || !std::isfinite(p0[i])) return -2;
    if (!all_finite(P.x, m) || !all_finite(P.y, m) || !all_finite(P.z, m)) return -1;
    double x[N];
    std::memcpy(x, p0, sizeof(x));
    if (!in_bounds(x, lb, ub)) return -2;  // least_squares: "`x0` is infeasible."
    make_strictly_feasible(x, lb, ub, 1e-10);
    double f[MMAX], J[MMAX * N], g[N];
    residuals(P, x, f);
    if (!all_finite(f, m)) return -1;
    auto jac = [&](const double* xx, const double* ff) {
        if (analytic_jac) jac_analytic(P, xx, J);
        else jac_2point(P, xx, ff, lb, ub, J);
    };
    auto grad = [&]() {
        for (int i = 0; i < N; ++i) {
            double acc = 0;
            for (int r = 0; r < m; ++r) acc += J[r * N + i] * f[r];
            g[i] = acc;
        }
    };
    jac(x, f);
    int nfev = 1;
    double cost = 0.5 * dot(f, f, m);
    grad();
    double v[N], dv[N];
    cl_scaling(x, g, lb, ub, v, dv);  // x_scale = 1: scale = scale_inv = 1
    double Delta;
    {
        double t[N];
        for (int i = 0; i < N; ++i) t[i] = x[i] / std::sqrt(v[i]);
        Delta = norm2(t, N);
        if (Delta == 0) Delta = 1.0;
    }
    if (max_nfev <= 0) max_nfev = 100 * N;
    double alpha = 0.0;
    int status = -100;  // None
    double Jaug[(MMAX + N) * N], faug[MMAX + N], sv[N], V[N * N], uf[N];
    while (true) {
        cl_scaling(x, g, lb, ub, v, dv);
        double g_norm = 0;
        for (int i = 0; i < N; ++i) g_norm = std::max(g_norm, std::fabs(g[i] * v[i]));
        if (g_norm < gtol) status = 1;
        if (status != -100 || nfev == max_nfev) break;
        double d[N], diag_h[N], g_h[N];
        for (int i = 0; i < N; ++i) {
            d[i] = std::sqrt(v[i]);
            diag_h[i] = g[i] * dv[i];
            g_h[i] = d[i] * g[i];
        }
        for (int r = 0; r < m; ++r) {
            faug[r] = f[r];
            for (int i = 0; i < N; ++i) Jaug[r * N + i] = J[r * N + i] * d[i];
        }
        double Jh[MMAX * N];
        std::memcpy(Jh, Jaug, sizeof(double) * m * N);
        for (int i = 0; i < N; ++i) {
            faug[m + i] = 0;
            for (int j = 0; j < N; ++j) Jaug[(m + i) * N + j] = (i == j) ? std::sqrt(diag_h[i]) : 0.0;
        }
        svd_jacobi(Jaug, m + N, sv, V);  // Jaug now holds U
        for (int k = 0; k < N; ++k) {
            double acc = 0;
            for (int r = 0; r < m + N; ++r) acc += Jaug[r * N + k] * faug[r];
            uf[k] = acc;
        }
        const double theta = std::max(0.995, 1 - g_norm);
        double actual_reduction = -1, x_new[N], f_new[MMAX], cost_new = cost;
        while (actual_reduction <= 0 && nfev < max_nfev) {
            double p_h[N], p[N], step[N], step_h[N];
            solve_lsq_trust_region(m, uf, sv, V, Delta, &alpha, p_h);  // (sic) scipy passes the m of J, not m + n
            for (int i = 0; i < N; ++i) p[i] = d[i] * p_h[i];
            const double predicted = select_step(x, Jh, m, diag_h, g_h, p, p_h, d, Delta, lb, ub, theta, step, step_h);
            for (int i = 0; i < N; ++i) x_new[i] = x[i] + step[i];
            make_strictly_feasible(x_new, lb, ub, 0.0);
            residuals(P, x_new, f_new);
            ++nfev;
            const double step_h_norm = norm2(step_h, N);
            if (!all_finite(f_new, m)) {
                Delta = 0.25 * step_h_norm;
                continue;
            }
            cost_new = 0.5 * dot(f_new, f_new, m);
            actual_reduction = cost - cost_new;
            double ratio;  // update_tr_radius
            if (predicted > 0) ratio = actual_reduction / predicted;
            else if (predicted == 0 && actual_reduction == 0) ratio = 1;
            else ratio = 0;
            double Delta_new = Delta;
            if (ratio < 0.25) Delta_new = 0.25 * step_h_norm;
            else if (ratio > 0.75 && step_h_norm > 0.95 * Delta) Delta_new = Delta * 2.0;
            const double step_norm = norm2(step, N), x_norm = norm2(x, N);
            const bool f_ok = actual_reduction < ftol * cost && ratio > 0.25;  // check_termination
            const bool x_ok = step_norm < xtol * (xtol + x_norm);
            if (f_ok && x_ok) status = 4;
            else if (f_ok) status = 2;
            else if (x_ok) status = 3;
            if (status != -100) break;
            alpha *= Delta / Delta_new;
            Delta = Delta_new;
        }
        if (actual_reduction > 0) {
            std::memcpy(x, x_new, sizeof(x));
            std::memcpy(f, f_new, sizeof(double) * m);
            cost = cost_new;
            jac(x, f);
            grad();
        }
    }
    if (status == -100) status = 0;
    std::memcpy(popt, x, sizeof(x));
    if (nfev_out) *nfev_out = nfev;
    if (cost_out) *cost_out = cost;
    return status;
}

}  // namespace coregfit
