// The writing side of csrc/ricecomp.hpp, host only: cfitsio's RICE_1 encoder (ricecomp.c `fits_rcomp`, `_short`, `_byte`)
// and its float quantization with subtractive dithering (quantize.c / imcompress.c `fits_quantize_float`'s last step:
// q = NINT((v - zero) / scale + r - 0.5)), restated from the published algorithm (cfitsio is third-party code absent from
// /root/reference; the reference WRITES tile-compressed images through astropy's CompImageHDU, utils/Util.py:137-138).
// Used by utils/fits_io.write_compressed_image (test scenes in the format EUI files have, written without astropy) and
// by the round-trip properties of the test-suite: decode(encode(q)) == q for every pixel width, and
// |dequantize(quantize(v)) - v| <= scale / 2.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#include "ricecomp.hpp"

namespace coregrice {

struct BitWriter {
    unsigned char* p;
    unsigned char* end;
    uint64_t acc = 0;
    int cnt = 0;  // bits waiting in acc (< 8 between calls)
    bool overflow = false;
    void put(uint32_t v, int n) {  // the low n bits of v, most significant first (n <= 32)
        if (n == 0) return;
        acc = (acc << n) | (n == 32 ? (uint64_t)v : (uint64_t)(v & ((1u << n) - 1u)));
        cnt += n;
        while (cnt >= 8) {
            cnt -= 8;
            if (p < end) *p++ = (unsigned char)(acc >> cnt);
            else overflow = true;
        }
    }
    void zeros(uint32_t n) {
        for (; n >= 32; n -= 32) put(0u, 32);
        put(0u, (int)n);
    }
    void flush() {
        if (cnt > 0) put(0u, 8 - cnt);
    }
};

// One tile: `q` are the stored integers (already reduced to the pixel width: int8 as 0..255, int16, int32).  Returns the
// number of bytes written, or -1 when `cap` does not suffice.
inline int64_t rice_encode_tile(const int32_t* q, int nx, int nblock, int bytepix, unsigned char* out, int64_t cap) {
    const int fsbits = bytepix == 4 ? 5 : (bytepix == 2 ? 4 : 3);
    const int fsmax = bytepix == 4 ? 25 : (bytepix == 2 ? 14 : 6);
    const int bbits = 1 << fsbits;
    const uint32_t vmask = bytepix == 4 ? 0xffffffffu : (bytepix == 2 ? 0xffffu : 0xffu);
    BitWriter w;
    w.p = out;
    w.end = out + cap;
    uint32_t diff[1024];
    if (nblock > 1024 || nx <= 0) return -1;
    uint32_t lastpix = (uint32_t)q[0] & vmask;
    w.put(lastpix, bbits);
    for (int i = 0; i < nx; i += nblock) {
        const int thisblock = nx - i < nblock ? nx - i : nblock;
        double pixelsum = 0.0;
        for (int j = 0; j < thisblock; ++j) {
            const uint32_t nextpix = (uint32_t)q[i + j] & vmask;
            // the difference in the pixel's own width (wrapping), then zig-zag: non-negative d -> 2 d, negative -> ~(2 d)
            uint32_t d = (nextpix - lastpix) & vmask;
            const bool neg = (d >> (bbits - 1)) & 1u;
            int64_t sd = neg ? (int64_t)d - ((int64_t)vmask + 1) : (int64_t)d;
            const uint64_t twice = (uint64_t)sd << 1;  // (two's complement: the shift of cfitsio's int arithmetic)
            const uint32_t z = (uint32_t)(sd < 0 ? ~twice : twice) & (bytepix == 4 ? 0xffffffffu : (vmask << 1 | 1u));
            diff[j] = z;
            pixelsum += (double)z;
            lastpix = nextpix;
        }
        double dpsum = (pixelsum - (double)(thisblock / 2) - 1.0) / (double)thisblock;
        if (dpsum < 0) dpsum = 0.0;
        uint32_t psum = bytepix == 4 ? ((uint32_t)dpsum) >> 1
                                     : (bytepix == 2 ? (uint32_t)((uint16_t)dpsum >> 1) : (uint32_t)((uint8_t)dpsum >> 1));
        int fs = 0;
        for (; psum > 0; ++fs) psum >>= 1;
        if (fs >= fsmax) {
            w.put((uint32_t)(fsmax + 1), fsbits);
            for (int j = 0; j < thisblock; ++j) w.put(diff[j], bbits);
        } else if (fs == 0 && pixelsum == 0.0) {
            w.put(0u, fsbits);
        } else {
            w.put((uint32_t)(fs + 1), fsbits);
            const uint32_t fsmask = fs == 0 ? 0u : (1u << fs) - 1u;
            for (int j = 0; j < thisblock; ++j) {
                const uint32_t v = diff[j];
                w.zeros(v >> fs);
                w.put(1u, 1);
                w.put(v & fsmask, fs);
            }
        }
        if (w.overflow) return -1;
    }
    w.flush();
    if (w.overflow) return -1;
    return (int64_t)(w.p - out);
}

inline int32_t nint_c(double x) { return x >= 0.0 ? (int32_t)(x + 0.5) : (int32_t)(x - 0.5); }

// Quantize one tile of floating-point pixels: zero = the smallest finite value, q = NINT((v - zero) / scale [+ r - 0.5]).
// NaN -> kNullValue; SUBTRACTIVE_DITHER_2 keeps exact zeros (kZeroValue).  Returns 0, or 1 when the tile's range does
// not fit 32-bit integers at this scale.
template <typename T>
inline int quantize_tile(const T* img, int naxis1, const TileBox& b, int quantize, int iseed0, const float* randoms,
                         double scale, int32_t* q, double* zero_out) {
#pragma clang fp contract(off)
    double vmin = INFINITY, vmax = -INFINITY;
    for (int y = 0; y < b.th; ++y)
        for (int x = 0; x < b.tw; ++x) {
            const double v = (double)img[(int64_t)(b.y0 + y) * naxis1 + b.x0 + x];
            if (std::isfinite(v)) {
                vmin = v < vmin ? v : vmin;
                vmax = v > vmax ? v : vmax;
            }
        }
    const double zero = std::isfinite(vmin) ? vmin : 0.0;
    *zero_out = zero;
    if (std::isfinite(vmax) && (vmax - zero) / scale > 2147483000.0) return 1;
    int iseed = iseed0;
    int nextrand = (int)(randoms[iseed] * 500);
    int i = 0;
    for (int y = 0; y < b.th; ++y)
        for (int x = 0; x < b.tw; ++x, ++i) {
            const double v = (double)img[(int64_t)(b.y0 + y) * naxis1 + b.x0 + x];
            const bool dith = quantize == Q_DITHER_1 || quantize == Q_DITHER_2;
            if (!std::isfinite(v)) q[i] = kNullValue;
            else if (quantize == Q_DITHER_2 && v == 0.0) q[i] = kZeroValue;
            else if (dith) q[i] = nint_c((v - zero) / scale + (double)randoms[nextrand] - 0.5);
            else q[i] = nint_c((v - zero) / scale);
            if (dith) {  // (the sequence advances for null pixels too)
                ++nextrand;
                if (nextrand == kNRandom) {
                    ++iseed;
                    if (iseed == kNRandom) iseed = 0;
                    nextrand = (int)(randoms[iseed] * 500);
                }
            }
        }
    return 0;
}

}  // namespace coregrice
