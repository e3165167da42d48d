// Sanitizer harness for the Gaussian fit of the library (csrc/fit.hpp: scipy's bounded trust-region-reflective
// least-squares restated), built by tests/test_fit_sanitizers_cpu.py with g++ -fsanitize=address,undefined.
// Random peaks as AlignmentResults hands them over (<= 26 points on a lag lattice), plus what real maps produce at their
// worst: flat maps, a single point, points on one line, NaN / Inf samples, start values on a bound, bounds a hair apart,
// infinite bounds.  Properties: no sanitizer report, the call returns one of scipy's status codes, a successful fit stays
// inside the bounds and is finite, and the same input gives the same bits twice.
// usage: fuzz_fit <iterations> <seed>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../euispice_coreg_amd/csrc/fit.hpp"

int main(int argc, char** argv) {
    const long iters = argc > 1 ? std::atol(argv[1]) : 2000;
    std::mt19937_64 rng(argc > 2 ? std::atoll(argv[2]) : 1);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    int failures = 0;
    long ok_fits = 0;
    for (long it = 0; it < iters; ++it) {
        const int kind = (int)(rng() % 10);
        int m = kind == 1 ? 1 : (int)(rng() % 26) + 1;
        double* x = (double*)std::malloc(sizeof(double) * m);  // (allocated to the element: ASan guards both ends)
        double* y = (double*)std::malloc(sizeof(double) * m);
        double* z = (double*)std::malloc(sizeof(double) * m);
        const double step = 0.25 + 4 * U(rng), cx = 20 * U(rng) - 10, cy = 20 * U(rng) - 10;
        const double a = 0.05 + U(rng), sx = (0.3 + 3 * U(rng)) * step, sy = (0.3 + 3 * U(rng)) * step, off = U(rng) - 0.5;
        for (int i = 0; i < m; ++i) {
            const int gx = i % 5 - 2, gy = i / 5 - 2;
            x[i] = cx + step * gx;
            y[i] = kind == 2 ? cy : cy + step * gy;  // kind 2: every point on one line
            const double dx = x[i] - (cx + 0.4 * step * (U(rng) - 0.5)), dy = y[i] - (cy + 0.3 * step);
            z[i] = off + a * std::exp(-(dx * dx / (2 * sx * sx) + dy * dy / (2 * sy * sy))) + 1e-3 * (U(rng) - 0.5);
            if (kind == 3) z[i] = 0.7;  // a flat map
        }
        if (kind == 4) z[rng() % m] = std::numeric_limits<double>::quiet_NaN();
        if (kind == 5) z[rng() % m] = std::numeric_limits<double>::infinity();
        double p0[6] = {a, cx, cy, step, step, off};
        double lb[6] = {0, cx - 3 * step, cy - 3 * step, 0, 0, -1};
        double ub[6] = {2, cx + 3 * step, cy + 3 * step, 50 * step, 50 * step, 1};
        if (kind == 6) p0[1] = lb[1];  // on a bound
        if (kind == 7) {               // bounds a hair apart
            lb[3] = step;
            ub[3] = std::nextafter(step, 1e9);
            p0[3] = step;
        }
        if (kind == 8)
            for (int i = 0; i < 6; ++i) {
                lb[i] = -std::numeric_limits<double>::infinity();
                ub[i] = std::numeric_limits<double>::infinity();
            }
        if (kind == 9) p0[4] = 1e-12 * step;  // a needle as the start value
        coregfit::Problem P{m, x, y, z};
        double popt[6], popt2[6], cost = 0;
        int nfev = 0, nfev2 = 0;
        const int jac = (int)(rng() & 1);
        const int st = coregfit::fit(P, p0, lb, ub, jac, 1e-8, 1e-8, 1e-8, 600, popt, &nfev, &cost);
        const int st2 = coregfit::fit(P, p0, lb, ub, jac, 1e-8, 1e-8, 1e-8, 600, popt2, &nfev2, nullptr);
        if (st != st2 || nfev != nfev2 || (st >= 0 && std::memcmp(popt, popt2, sizeof(popt)) != 0)) {
            ++failures;
            std::fprintf(stderr, "it %ld: not reproducible (%d / %d)\n", it, st, st2);
        }
        if (st < -2 || st > 4) {
            ++failures;
            std::fprintf(stderr, "it %ld: status %d\n", it, st);
        }
        if ((kind == 4 || kind == 5) && st != -1) {
            ++failures;
            std::fprintf(stderr, "it %ld: non-finite sample not refused (%d)\n", it, st);
        }
        if (st > 0) {
            ++ok_fits;
            for (int i = 0; i < 6; ++i)
                if (!std::isfinite(popt[i]) || popt[i] < lb[i] || popt[i] > ub[i]) {
                    ++failures;
                    std::fprintf(stderr, "it %ld kind %d: popt[%d] = %g outside [%g, %g]\n", it, kind, i, popt[i], lb[i], ub[i]);
                }
        }
        std::free(x);
        std::free(y);
        std::free(z);
    }
    if (failures) std::fprintf(stderr, "%d failure(s)\n", failures);
    else std::printf("ok: %ld iterations, %ld converged fits\n", iters, ok_fits);
    return failures ? 1 : 0;
}
