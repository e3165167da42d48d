// Sanitizer harness for the host-side header arithmetic the C ABI runs on whatever header it is handed
// (csrc/geometry.hpp: shift_header, the TAN -> TAN homography and its lag family, wcslib's TAN and CAR chains restated
// -- WcslibTan / WcslibCar --, the plate-carree maps, the Carrington transform parameters and grid tables), built by
// tests/test_geometry_sanitizers_cpu.py with g++ -fsanitize=address,undefined,float-cast-overflow.
// Headers: sane random ones (roll, unequal CDELT, arcsec / degrees, off-centre CRPIX, both hemispheres) and hostile ones
// (zero / negative / denormal CDELT, singular PC, CRVAL at the poles or far outside, NaN and Inf cards, LONPOLE / LATPOLE
// defaults and nonsense, huge multiples of 90 degrees).  Properties: no sanitizer report on any of them; on sane headers
// the identity map returns the pixel it was given (wcslib chain and homography both), the homography agrees with the
// wcslib chain, and the same input gives the same bits twice.
// usage: fuzz_geometry <iterations> <seed>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>

#include "../../euispice_coreg_amd/csrc/geometry.hpp"

using namespace coreg;

static std::mt19937_64 rng;
static double U(double a, double b) { return a + (b - a) * std::uniform_real_distribution<double>(0.0, 1.0)(rng); }

static coreg_wcs2d sane(int proj) {
    coreg_wcs2d w;
    std::memset(&w, 0, sizeof w);
    w.naxis1 = 16 + (int)(rng() % 2048);
    w.naxis2 = 16 + (int)(rng() % 2048);
    const bool deg = rng() % 3 == 0 || proj == COREG_PROJ_CAR;
    w.unit_to_deg = deg ? 1.0 : 1.0 / 3600.0;
    const double as = deg ? 1.0 / 3600.0 : 1.0;  // one arcsec in header units
    w.crpix1 = (w.naxis1 + 1) / 2.0 + U(-20, 20);
    w.crpix2 = (w.naxis2 + 1) / 2.0 + U(-20, 20);
    if (proj == COREG_PROJ_CAR) {  // (a map that stays inside (-180, 180] x (-90, 90) of its native frame)
        w.naxis1 = 16 + w.naxis1 % 1000;
        w.naxis2 = 16 + w.naxis2 % 1000;
        w.crpix1 = (w.naxis1 + 1) / 2.0 + U(-20, 20);
        w.crpix2 = (w.naxis2 + 1) / 2.0 + U(-20, 20);
        w.crval1 = U(0, 360);
        w.crval2 = U(-40, 40);
        w.cdelt1 = U(0.01, 0.05);
        w.cdelt2 = U(0.01, 0.05);
        w.lonpole = rng() % 2 ? std::numeric_limits<double>::quiet_NaN() : (w.crval2 >= 0 ? 0.0 : 180.0);
        w.latpole = rng() % 2 ? std::numeric_limits<double>::quiet_NaN() : 90.0;
    } else {
        w.crval1 = U(-1500, 1500) * as;
        w.crval2 = U(-1500, 1500) * as;
        w.cdelt1 = U(0.4, 5.0) * as;
        w.cdelt2 = U(0.4, 5.0) * as;
        w.lonpole = 180.0;
        w.latpole = std::numeric_limits<double>::quiet_NaN();
    }
    w.crota = rng() % 4 == 0 ? 0.0 : U(-30, 30);
    const double rho = w.crota * kDeg2Rad, lam = w.cdelt2 / w.cdelt1;
    w.pc1_1 = std::cos(rho);
    w.pc2_2 = std::cos(rho);
    w.pc1_2 = -lam * std::sin(rho);
    w.pc2_1 = std::sin(rho) / lam;
    w.dsun_obs = U(0.28, 1.02) * 1.495978707e11;
    w.crln_obs = U(0, 360);
    w.crlt_obs = U(-8, 8);
    w.proj = proj;
    return w;
}

static double nasty() {
    static const double v[] = {0.0, -0.0, 1e-320, -1e-320, 1e300, -1e300, 90.0, -90.0, 180.0, 270.0, 360.0, 999.0,
                               9.0e20, -9.0e20, 1e-30, std::numeric_limits<double>::quiet_NaN(),
                               std::numeric_limits<double>::infinity(), -std::numeric_limits<double>::infinity()};
    return v[rng() % (sizeof v / sizeof v[0])];
}

static coreg_wcs2d hostile(int proj) {
    coreg_wcs2d w = sane(proj);
    double* f[] = {&w.crpix1, &w.crpix2, &w.crval1, &w.crval2, &w.cdelt1, &w.cdelt2, &w.pc1_1, &w.pc1_2, &w.pc2_1,
                   &w.pc2_2, &w.crota, &w.unit_to_deg, &w.lonpole, &w.dsun_obs, &w.crln_obs, &w.crlt_obs, &w.latpole};
    const int n = 1 + (int)(rng() % 4);
    for (int i = 0; i < n; ++i) *f[rng() % (sizeof f / sizeof f[0])] = nasty();
    if (rng() % 8 == 0) w.pc1_1 = w.pc1_2 = w.pc2_1 = w.pc2_2 = 0.0;  // singular
    return w;
}

// everything the library computes from a header pair, once; returns a checksum of the finite outputs
static double exercise(const coreg_wcs2d& a, const coreg_wcs2d& b, bool sane_pair, int* failures) {
    double acc = 0.0;
    auto add = [&](double v) {
        if (v == v && std::fabs(v) < 1e300) acc += v;
    };
    const double px[4] = {0.0, (double)(a.naxis1 - 1), 0.37 * a.naxis1, -3.5};
    const double py[4] = {0.0, (double)(a.naxis2 - 1), 0.81 * a.naxis2, a.naxis2 + 2.25};
    coreg_wcs2d s;
    shift_header(b, 1.5 * b.cdelt1, -2.0 * b.cdelt2, 0.0, 0.0, 0.3, COREG_CDELT_INTENDED, &s);
    coreg_wcs2d s2;
    add((double)shift_header(b, 0.0, 0.0, 0.01 * b.cdelt1, 0.01 * b.cdelt2, 0.0, COREG_CDELT_REFERENCE, &s2));
    if (a.proj == COREG_PROJ_TAN) {
        double h[9], hid[9];
        homography(a, s, h);
        homography(a, a, hid);
        const double lag1[3] = {-b.cdelt1, 0.0, 2.0 * b.cdelt1}, lag2[2] = {0.0, 3.0 * b.cdelt2};
        HomographyFamily fam;
        fam.init(a, b, lag1, 3, lag2, 2, true);
        fam.fill_products(0, 2);
        const Mat3d B = HomographyFamily::combo(b);
        double hf[9];
        fam.get(B, 1, 0, hf);  // zero lag on both axes: the map a -> b
        WcslibTan wa, wb;
        wa.init(a);
        wb.init(s);
        for (int k = 0; k < 4; ++k) {
            double x, y, xi, yi, xw, yw, xs, ys;
            apply_h(h, px[k], py[k], &x, &y);
            apply_h(hid, px[k], py[k], &xi, &yi);
            wcslib_pixel_to_pixel(wa, wb, px[k], py[k], &xw, &yw);
            wcslib_pixel_to_pixel(wa, wa, px[k], py[k], &xs, &ys);
            add(x), add(y), add(xw), add(yw);
            if (sane_pair) {
                if (!(std::fabs(xi - px[k]) < 1e-6 && std::fabs(yi - py[k]) < 1e-6)) {
                    std::printf("identity homography off: (%g, %g) -> (%.12g, %.12g)\n", px[k], py[k], xi, yi);
                    ++*failures;
                }
                if (!(std::fabs(xs - px[k]) < 1e-6 && std::fabs(ys - py[k]) < 1e-6)) {
                    std::printf("identity wcslib chain off: (%g, %g) -> (%.12g, %.12g)\n", px[k], py[k], xs, ys);
                    ++*failures;
                }
                if (!(std::fabs(x - xw) < 1e-6 && std::fabs(y - yw) < 1e-6)) {
                    std::printf("homography vs wcslib chain: (%.12g, %.12g) vs (%.12g, %.12g)\n", x, y, xw, yw);
                    ++*failures;
                }
            }
            double xf, yf;
            apply_h(hf, px[k], py[k], &xf, &yf);
            add(xf), add(yf);
        }
        add(ang2pipi_deg(a.crval1 * a.unit_to_deg));
    } else {
        WcslibCar ca, cb;
        ca.init(a);
        cb.init(s);
        CarMapHost m;
        const int bad = m.init(a, s);
        ld lngp, latp, phip;
        add((double)car_euler(a, &lngp, &latp, &phip));
        for (int k = 0; k < 4; ++k) {
            double xw, yw, xs, ys, xm = 0, ym = 0;
            wcslib_pixel_to_pixel(ca, cb, px[k], py[k], &xw, &yw);
            wcslib_pixel_to_pixel(ca, ca, px[k], py[k], &xs, &ys);
            if (!bad) m.apply(px[k], py[k], &xm, &ym);
            add(xw), add(yw), add(xm), add(ym);
            if (sane_pair && ca.valid && k < 3) {
                if (!(std::fabs(xs - px[k]) < 1e-6 && std::fabs(ys - py[k]) < 1e-6)) {
                    std::printf("identity CAR chain off: (%g, %g) -> (%.12g, %.12g)\n", px[k], py[k], xs, ys);
                    ++*failures;
                }
                if (!bad && cb.valid && !(std::fabs(xm - xw) < 1e-5 && std::fabs(ym - yw) < 1e-5)) {
                    std::printf("CAR map vs wcslib chain: (%.12g, %.12g) vs (%.12g, %.12g)\n", xm, ym, xw, yw);
                    ++*failures;
                }
            }
        }
        if (!bad) add(car_pole_sep(m.r));
        add(car_box_c(a, s, 32));
    }
    const CarrCommon c = carr_common(a, 1.004);
    double x0, y0;
    carr_origin(a, &x0, &y0);
    add(c.dist), add(c.cb), add(c.sr), add(x0), add(y0);
    coreg_carr_grid g;
    std::memset(&g, 0, sizeof g);
    g.lon0 = a.crln_obs - 20.0;
    g.lon1 = a.crln_obs + 20.0;
    g.lat0 = -15.0;
    g.lat1 = 25.0;
    g.n_lon = 1 + (int)(rng() % 64);
    g.n_lat = 1 + (int)(rng() % 64);
    CarrTables t;
    carr_tables(g, a.crln_obs, t);
    add(t.sin_lon[g.n_lon - 1]), add(t.cos_lat[g.n_lat - 1]);
    return acc;
}

int main(int argc, char** argv) {
    const long iters = argc > 1 ? std::atol(argv[1]) : 2000;
    rng.seed(argc > 2 ? std::atoll(argv[2]) : 1);
    int failures = 0;
    long sane_pairs = 0;
    for (long it = 0; it < iters; ++it) {
        const int proj = rng() % 3 == 0 ? COREG_PROJ_CAR : COREG_PROJ_TAN;
        const bool ok = rng() % 2 == 0;
        coreg_wcs2d a = ok ? sane(proj) : hostile(proj);
        coreg_wcs2d b = a;
        if (ok) {  // a slightly different header of the same field, as a lagged header is
            b.crval1 += U(-30, 30) * (proj == COREG_PROJ_CAR ? 0.01 : (a.unit_to_deg == 1.0 ? 1.0 / 3600.0 : 1.0));
            b.crpix1 += U(-3, 3);
        } else if (rng() % 2) {
            b = hostile(proj);
        }
        sane_pairs += ok;
        if (ok && (wcs_problem(a, false) || wcs_problem(a, true) || wcs_problem(b, false))) {
            std::printf("a sane header was refused: %s\n", wcs_problem(a, false) ? wcs_problem(a, false) : "(transform / lagged)");
            ++failures;
        }
        (void)wcs_problem(b, rng() % 2 == 0);  // hostile ones: any answer, no report
        const auto state = rng;  // exercise() draws grid sizes: same draws for the second run
        const double c1 = exercise(a, b, ok, &failures);
        rng = state;
        int dummy = 0;
        const double c2 = exercise(a, b, false, &dummy);
        if (std::memcmp(&c1, &c2, sizeof c1) != 0) {
            std::printf("not reproducible at iteration %ld\n", it);
            ++failures;
        }
    }
    if (failures) {
        std::printf("FAILED: %d property violations\n", failures);
        return 1;
    }
    std::printf("ok: %ld iterations (%ld sane header pairs)\n", iters, sane_pairs);
    return 0;
}
