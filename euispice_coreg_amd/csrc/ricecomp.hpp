// Tile-compressed FITS images (the "tiled image compression convention": the image lives in a binary table, one row per
// tile, the compressed bytes in the table's heap).  EUI level-1 / level-2 files store their image that way; the
// reference reads them through astropy.io.fits (hdrshift/alignment.py:191-208, :299-314 -- `hdul[window].data` on a
// CompImageHDU) and its writer has a CompImageHDU branch (utils/Util.py:137-138).  The codec is third-party code absent
// from /root/reference: cfitsio (bundled with astropy; ricecomp.c `fits_rdecomp*`, imcompress.c `unquantize_i4r4 / _i4r8`,
// `fits_init_randoms`), restated here from its published algorithm:
//   * RICE_1: per tile, the first pixel verbatim (BYTEPIX bytes, big-endian), then blocks of BLOCKSIZE mapped
//     differences, each block introduced by an FS code of 5 / 4 / 3 bits (BYTEPIX 4 / 2 / 1): 0 = all differences zero,
//     fsmax + 1 = differences stored verbatim, else Rice code with FS low bits; differences are zig-zag mapped and taken
//     modulo 2^(8 BYTEPIX);
//   * floating-point images are quantized integers per tile: value = (q - r + 0.5) * ZSCALE + ZZERO with r from a fixed
//     pseudo-random sequence (SUBTRACTIVE_DITHER_1 / _2; _2 keeps exact zeros: q = -2147483646), or q * ZSCALE + ZZERO
//     (NO_DITHER); q = ZBLANK (default -2147483647) is NaN.
// One function decodes one tile, compiled for the GPU (one thread per tile: a tile's bit stream is sequential) and for the
// host (CPU tests against astropy's output, tests/golden/compressed_golden.npz; rare fallbacks).  Every read of the
// compressed stream is bounds-checked: a truncated or corrupt tile yields an error flag, never an out-of-range access.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define COREG_HD __host__ __device__
#else
#define COREG_HD
#endif

namespace coregrice {

constexpr int kNRandom = 10000;            // cfitsio N_RANDOM
constexpr int32_t kNullValue = -2147483647;  // cfitsio NULL_VALUE: default ZBLANK of quantized float images
constexpr int32_t kZeroValue = -2147483646;  // cfitsio ZERO_VALUE (SUBTRACTIVE_DITHER_2)

enum { Q_NONE = 0, Q_NO_DITHER = 1, Q_DITHER_1 = 2, Q_DITHER_2 = 3 };
enum { OUT_F32 = 0, OUT_F64 = 1 };

// cfitsio fits_init_randoms: Park & Miller's minimal standard generator, stored as float
inline void init_randoms(float* r /* [kNRandom] */) {
    const double a = 16807.0, m = 2147483647.0;
    double seed = 1.0;
    for (int i = 0; i < kNRandom; ++i) {
        const double temp = a * seed;
        seed = temp - m * (double)((int)(temp / m));
        r[i] = (float)(seed / m);
    }
}

struct TileImage {
    // geometry
    int32_t naxis1, naxis2, ztile1, ztile2;
    // codec
    int32_t bytepix, blocksize;
    // what the decoded integers mean
    int32_t zbitpix;   // 8 / 16 / 32: integer image (physical = bscale * q + bzero), -32 / -64: quantized floats
    int32_t quantize;  // Q_*
    int32_t dither0;   // ZDITHER0
    int32_t has_blank;
    int32_t blank;     // ZBLANK (integer images: -> NaN when has_blank; float images: default kNullValue)
    int32_t scaled;    // integer images: apply bscale / bzero (two roundings, as NumPy: float64(q) * bscale + bzero)
    double bscale, bzero;
    double zscale0, zzero0;      // when the per-tile tables are null
    const double* zscale;        // [n_tiles] or null
    const double* zzero;         // [n_tiles] or null
    const float* randoms;        // [kNRandom] (dithered float images)
    // the compressed bytes
    const unsigned char* heap;
    int64_t heap_bytes;
    const int64_t* tile_offset;  // [n_tiles]
    const int32_t* tile_nbytes;  // [n_tiles]; <= 0: the tile is not Rice-coded (skipped, flagged in `status`)
    int32_t n_tiles;
    // output image [naxis2][naxis1]
    void* out;
    int32_t out_dtype;  // OUT_F32 / OUT_F64
};

// What a decoded integer means: the physical pixel value.  `rand_idx`: index into cfitsio's random sequence for this pixel
// (dithered float images; see dither_index).
COREG_HD inline double pixel_value(const TileImage& t, double scale, double zero, int32_t q, int rand_idx) {
#pragma clang fp contract(off)  // cfitsio multiplies, rounds, adds, rounds (x86-64 baseline): no FMA here
    if (t.quantize == Q_NONE) {
        if (t.has_blank && q == t.blank) return __builtin_nan("");
        double v = (double)q;
        if (t.scaled) v = v * t.bscale + t.bzero;  // (two roundings, as NumPy's float64(q) * bscale + bzero)
        return v;
    }
    if (q == t.blank) return __builtin_nan("");
    if (t.quantize == Q_DITHER_2 && q == kZeroValue) return 0.0;
    if (t.quantize == Q_DITHER_1 || t.quantize == Q_DITHER_2)
        return ((double)q - (double)t.randoms[rand_idx] + 0.5) * scale + zero;
    return (double)q * scale + zero;
}

// cfitsio's walk through its random sequence, in closed form: pixel i of a tile whose walk starts at (iseed, nextrand =
// (int)(rand[iseed] * 500)) uses rand[nextrand + i]; whenever the index reaches N_RANDOM the walk restarts at
// (int)(rand[++iseed mod N] * 500).  (The sequence advances for null pixels too.)
COREG_HD inline int dither_index(const float* randoms, int iseed, int i) {
    int start = (int)(randoms[iseed] * 500);
    while (start + i >= kNRandom) {
        i -= kNRandom - start;
        iseed = iseed + 1 == kNRandom ? 0 : iseed + 1;
        start = (int)(randoms[iseed] * 500);
    }
    return start + i;
}
COREG_HD inline int dither_seed(const TileImage& t, int n) {
    // cfitsio: unquantize(row = tile number (1-based) + ZDITHER0 - 1): iseed = (row - 1) % N_RANDOM
    const long long row = (long long)(n + 1) + t.dither0 - 1;
    return (int)(((row - 1) % kNRandom + kNRandom) % kNRandom);
}

// Where the decoded integers of a tile go.  PixelSink: straight to the output image, value by value (host; GPU tiles too
// large for the staged path).  QSink: into a buffer of integers (the GPU kernel's lane 0; the other lanes turn them into
// pixel values afterwards).
struct PixelSink {
    const TileImage* im;
    double scale, zero;
    int iseed, nextrand;
    int x0, y0, tw, tx, ty;  // tile origin and width, running position inside the tile
    COREG_HD void put(int32_t q) {
        const TileImage& t = *im;
        const int64_t at = (int64_t)(y0 + ty) * t.naxis1 + (x0 + tx);
        if (++tx == tw) {
            tx = 0;
            ++ty;
        }
        const double v = pixel_value(t, scale, zero, q, nextrand);
        if (t.quantize == Q_DITHER_1 || t.quantize == Q_DITHER_2) {
            ++nextrand;
            if (nextrand == kNRandom) {
                ++iseed;
                if (iseed == kNRandom) iseed = 0;
                nextrand = (int)(t.randoms[iseed] * 500);
            }
        }
        if (t.out_dtype == OUT_F32) ((float*)t.out)[at] = (float)v;
        else ((double*)t.out)[at] = v;
    }
};
struct QSink {
    int32_t* q;
    int i;
    COREG_HD void put(int32_t v) { q[i++] = v; }
};

// position (1 .. 8) of the highest set bit of a non-zero byte: cfitsio's nonzero_count[] table
COREG_HD inline int top_bit(unsigned b) { return 32 - __builtin_clz(b); }

// cfitsio fits_rdecomp / fits_rdecomp_short / fits_rdecomp_byte, one routine: BYTEPIX = 4 / 2 / 1.
// Returns 0, or 1 when the stream ends early / is inconsistent (the remaining pixels of the tile are then NaN-less garbage
// free: they are written as the last good value so that nothing stays uninitialised; the caller reports the error).
template <typename Sink>
COREG_HD inline int rice_decode_tile(const unsigned char* c, int64_t clen, int nx, int nblock, int bytepix, Sink& sink) {
    const int fsbits = bytepix == 4 ? 5 : (bytepix == 2 ? 4 : 3);
    const int fsmax = bytepix == 4 ? 25 : (bytepix == 2 ? 14 : 6);
    const int bbits = 1 << fsbits;
    const unsigned vmask = bytepix == 4 ? 0xffffffffu : (bytepix == 2 ? 0xffffu : 0xffu);
    const unsigned char* const cend = c + clen;
    int err = 0;
    auto next_byte = [&]() -> unsigned {
        if (c < cend) return *c++;
        err = 1;
        return 0u;
    };
    auto emit = [&](unsigned v) {  // the value modulo 2^(8 bytepix), as the signed / unsigned type of the image
        v &= vmask;
        int32_t q;
        if (bytepix == 4) q = (int32_t)v;
        else if (bytepix == 2) q = (int32_t)(int16_t)(uint16_t)v;
        else q = (int32_t)v;  // BITPIX = 8 is unsigned
        sink.put(q);
    };
    if (clen < bytepix + 1) {
        for (int i = 0; i < nx; ++i) emit(0u);
        return 1;
    }
    unsigned lastpix = 0;
    for (int k = 0; k < bytepix; ++k) lastpix = (lastpix << 8) | next_byte();
    unsigned b = next_byte();  // bit buffer
    int nbits = 8;             // bits remaining in b
    for (int i = 0; i < nx;) {
        nbits -= fsbits;
        while (nbits < 0) {
            b = (b << 8) | next_byte();
            nbits += 8;
        }
        const int fs = (int)(b >> nbits) - 1;
        b &= (1u << nbits) - 1u;
        int imax = i + nblock;
        if (imax > nx) imax = nx;
        if (fs < 0) {
            for (; i < imax; ++i) emit(lastpix);  // all differences zero
        } else if (fs == fsmax) {
            for (; i < imax; ++i) {  // differences stored verbatim, bbits bits each
                int k = bbits - nbits;
                unsigned diff = k < 32 ? (b << k) : 0u;
                for (k -= 8; k >= 0; k -= 8) {
                    b = next_byte();
                    diff |= b << k;
                }
                if (nbits > 0) {
                    b = next_byte();
                    diff |= b >> (-k);
                    b &= (1u << nbits) - 1u;
                } else {
                    b = 0;
                }
                diff = (diff & 1u) == 0 ? diff >> 1 : ~(diff >> 1);
                lastpix = (diff + lastpix) & vmask;
                emit(lastpix);
            }
        } else if (fs > fsmax) {
            err = 1;
            for (; i < imax; ++i) emit(lastpix);
        } else {
            for (; i < imax; ++i) {  // Rice code: unary high part, fs low bits
                while (b == 0) {
                    nbits += 8;
                    b = next_byte();
                    if (err) break;
                }
                if (err) {
                    emit(lastpix);
                    continue;
                }
                const int nzero = nbits - top_bit(b);
                nbits -= nzero + 1;
                b ^= 1u << nbits;  // flip the leading one-bit
                nbits -= fs;
                while (nbits < 0) {
                    b = (b << 8) | next_byte();
                    nbits += 8;
                }
                unsigned diff = ((unsigned)nzero << fs) | (b >> nbits);
                b &= (1u << nbits) - 1u;
                diff = (diff & 1u) == 0 ? diff >> 1 : ~(diff >> 1);
                lastpix = (diff + lastpix) & vmask;
                emit(lastpix);
            }
        }
        if (err) {
            for (; i < nx; ++i) emit(lastpix);
            break;
        }
    }
    return err;
}

struct TileBox {
    int x0, y0, tw, th;
};
COREG_HD inline TileBox tile_box(const TileImage& t, int n) {
    const int ntx = (t.naxis1 + t.ztile1 - 1) / t.ztile1;
    const int tyi = n / ntx, txi = n - tyi * ntx;
    TileBox b;
    b.x0 = txi * t.ztile1;
    b.y0 = tyi * t.ztile2;
    b.tw = t.naxis1 - b.x0 < t.ztile1 ? t.naxis1 - b.x0 : t.ztile1;
    b.th = t.naxis2 - b.y0 < t.ztile2 ? t.naxis2 - b.y0 : t.ztile2;
    return b;
}

// Tile `n` of the image: geometry, per-tile scale / zero, dither start, then the Rice stream.  Returns 0 ok, 1 corrupt
// stream, 2 the tile is not Rice-coded (nothing written).
// `stream`: the tile's compressed bytes somewhere faster than the heap (the GPU kernel stages them in LDS), or null
COREG_HD inline int decode_tile(const TileImage& t, int n, const unsigned char* stream = nullptr) {
    const TileBox b = tile_box(t, n);
    PixelSink s;
    s.im = &t;
    s.x0 = b.x0;
    s.y0 = b.y0;
    s.tw = b.tw;
    s.tx = s.ty = 0;
    s.scale = t.zscale ? t.zscale[n] : t.zscale0;
    s.zero = t.zzero ? t.zzero[n] : t.zzero0;
    s.iseed = s.nextrand = 0;
    if (t.quantize == Q_DITHER_1 || t.quantize == Q_DITHER_2) {
        s.iseed = dither_seed(t, n);
        s.nextrand = (int)(t.randoms[s.iseed] * 500);
    }
    const int64_t off = t.tile_offset[n];
    const int64_t len = t.tile_nbytes[n];
    if (len <= 0) return 2;
    if (off < 0 || off + len > t.heap_bytes) return 1;
    return rice_decode_tile(stream ? stream : t.heap + off, len, b.tw * b.th, t.blocksize, t.bytepix, s);
}

}  // namespace coregrice
