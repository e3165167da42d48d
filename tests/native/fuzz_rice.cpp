// Sanitizer harness for the Rice codec the GPU kernel shares with the host (csrc/ricecomp.hpp, csrc/riceenc.hpp), built
// by tests/test_rice_sanitizers_cpu.py with g++ -fsanitize=address,undefined (GPU AddressSanitizer does not exist on
// the pool; the decoder is ONE function for both, so what holds here holds for k_rice_tiles' lane 0).
//   * round trip on random tiles of every pixel width: decode(encode(q)) == q, the decoder writes exactly nx integers;
//   * corrupt input -- truncated, bit-flipped, random bytes -- in buffers allocated to the byte: no read past the
//     stream, no write past the output, the call returns (flagged or not), never loops beyond the stream's end;
//   * the encoder never writes past its capacity;
//   * whole-tile decode (geometry, scale / zero tables, dithering walk) with tile tables that point outside the heap.
// usage: fuzz_rice <iterations> <seed>     exit code 0 = every property held
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../euispice_coreg_amd/csrc/riceenc.hpp"

using namespace coregrice;

static int failures = 0;
#define EXPECT(c, ...)                     \
    do {                                   \
        if (!(c)) {                        \
            if (failures++ < 20) {         \
                std::fprintf(stderr, __VA_ARGS__); \
                std::fputc('\n', stderr);  \
            }                              \
        }                                  \
    } while (0)

struct CountSink {
    int32_t* q;
    int cap, i;
    void put(int32_t v) {
        if (i < cap) q[i] = v;  // (a write past `cap` would be the bug: counted, and ASan guards the allocation)
        ++i;
    }
};

int main(int argc, char** argv) {
    const long iters = argc > 1 ? std::atol(argv[1]) : 2000;
    std::mt19937_64 rng(argc > 2 ? std::atoll(argv[2]) : 1);
    auto rnd = [&](long lo, long hi) { return lo + (long)(rng() % (unsigned long)(hi - lo + 1)); };
    std::vector<float> randoms(kNRandom);
    init_randoms(randoms.data());
    for (long it = 0; it < iters; ++it) {
        const int bytepix = (int[]){1, 2, 4}[rnd(0, 2)];
        const int nblock = (int[]){16, 32, 32, 32, 7, 64}[rnd(0, 5)];
        const int nx = (int)rnd(1, 700);
        const long long lo = bytepix == 1 ? 0 : (bytepix == 2 ? -32768 : -2147483648LL);
        const long long hi = bytepix == 1 ? 255 : (bytepix == 2 ? 32767 : 2147483647LL);
        std::vector<int32_t> q(nx);
        const int kind = (int)rnd(0, 4);
        long long level = rnd(0, 1000) * (hi - lo) / 1000 + lo;
        for (int i = 0; i < nx; ++i) {
            long long v;
            if (kind == 0) v = lo + (long long)(rng() % (unsigned long long)(hi - lo + 1));   // noise: differences wrap
            else if (kind == 1) v = level + rnd(-3, 3);                                        // smooth
            else if (kind == 2) v = (rng() % 17 == 0) ? (rng() & 1 ? hi : lo) : level;         // spikes
            else if (kind == 3) v = level;                                                     // constant
            else v = level + (long long)((double)(hi - lo) * 0.001 * (double)rnd(-100, 100));  // wide
            q[i] = (int32_t)(v < lo ? lo : (v > hi ? hi : v));
        }
        // ---- encode into a capacity known to suffice, then into capacities that do not
        const int64_t cap = (int64_t)nx * bytepix + nx / nblock + 16;
        std::vector<unsigned char> big(cap);
        const int64_t len = rice_encode_tile(q.data(), nx, nblock, bytepix, big.data(), cap);
        EXPECT(len > 0 && len <= cap, "encode failed: it %ld len %lld", it, (long long)len);
        if (len <= 0) continue;
        {
            const int64_t small_cap = rnd(0, len - 1);
            unsigned char* tight = (unsigned char*)std::malloc(small_cap ? small_cap : 1);
            EXPECT(rice_encode_tile(q.data(), nx, nblock, bytepix, tight, small_cap) == -1, "encoder ignored its capacity");
            std::free(tight);
        }
        // ---- decode from a buffer of exactly `len` bytes into exactly nx integers
        unsigned char* stream = (unsigned char*)std::malloc(len);
        std::memcpy(stream, big.data(), len);
        int32_t* out = (int32_t*)std::malloc((size_t)nx * 4);
        CountSink s = {out, nx, 0};
        int e = rice_decode_tile(stream, len, nx, nblock, bytepix, s);
        EXPECT(e == 0 && s.i == nx, "round trip flagged: it %ld e %d n %d / %d", it, e, s.i, nx);
        EXPECT(std::memcmp(out, q.data(), (size_t)nx * 4) == 0, "round trip differs: it %ld bytepix %d kind %d nx %d", it,
               bytepix, kind, nx);
        // ---- corrupt it
        for (int c = 0; c < 6; ++c) {
            int64_t clen = len;
            unsigned char* bad;
            if (c < 2) {  // truncated
                clen = rnd(0, len - 1);
                bad = (unsigned char*)std::malloc(clen ? clen : 1);
                std::memcpy(bad, stream, clen);
            } else if (c < 5) {  // bit flips
                bad = (unsigned char*)std::malloc(len);
                std::memcpy(bad, stream, len);
                for (int k = (int)rnd(1, 4); k > 0; --k) bad[rnd(0, len - 1)] ^= (unsigned char)(1u << rnd(0, 7));
            } else {  // random bytes
                clen = rnd(1, 64);
                bad = (unsigned char*)std::malloc(clen);
                for (int64_t k = 0; k < clen; ++k) bad[k] = (unsigned char)rng();
            }
            CountSink sb = {out, nx, 0};
            (void)rice_decode_tile(bad, clen, nx, nblock, bytepix, sb);
            EXPECT(sb.i == nx, "corrupt stream: %d integers written for a tile of %d", sb.i, nx);
            std::free(bad);
        }
        std::free(stream);
        std::free(out);
        // ---- a whole image of tiles now and then: quantized floats, every dither method, tables that lie
        if (it % 8 == 0) {
            const int ny = (int)rnd(1, 9), nxi = (int)rnd(1, 90), tx = (int)rnd(1, nxi + 3), ty = (int)rnd(1, ny + 1);
            std::vector<double> img((size_t)ny * nxi);
            for (auto& v : img) v = rng() % 23 == 0 ? __builtin_nan("") : (rng() % 19 == 0 ? 0.0 : 1e3 * (double)(rng() % 100000) / 1e5 - 50.0);
            TileImage t{};
            t.naxis1 = nxi;
            t.naxis2 = ny;
            t.ztile1 = tx;
            t.ztile2 = ty;
            t.bytepix = 4;
            t.blocksize = 32;
            t.zbitpix = -64;
            t.quantize = (int)rnd(1, 3);
            t.dither0 = (int)rnd(1, 10000);
            t.has_blank = 1;
            t.blank = kNullValue;
            t.randoms = randoms.data();
            const int ntx = (nxi + tx - 1) / tx, nty = (ny + ty - 1) / ty, nt = ntx * nty;
            std::vector<double> zs(nt), zz(nt);
            std::vector<int64_t> offs(nt);
            std::vector<int32_t> nb(nt);
            std::vector<unsigned char> heap;
            std::vector<int32_t> qt((size_t)tx * ty);
            const double scale = 0.01 * (double)rnd(1, 500);
            for (int n = 0; n < nt; ++n) {
                const TileBox b = tile_box(t, n);
                quantize_tile(img.data(), nxi, b, t.quantize, dither_seed(t, n), randoms.data(), scale, qt.data(), &zz[n]);
                zs[n] = scale;
                std::vector<unsigned char> buf((size_t)b.tw * b.th * 4 + 64);
                const int64_t l = rice_encode_tile(qt.data(), b.tw * b.th, 32, 4, buf.data(), (int64_t)buf.size());
                offs[n] = (int64_t)heap.size();
                nb[n] = (int32_t)l;
                heap.insert(heap.end(), buf.begin(), buf.begin() + l);
            }
            unsigned char* hp = (unsigned char*)std::malloc(heap.size());
            std::memcpy(hp, heap.data(), heap.size());
            double* o = (double*)std::malloc(img.size() * 8);
            t.heap = hp;
            t.heap_bytes = (int64_t)heap.size();
            t.tile_offset = offs.data();
            t.tile_nbytes = nb.data();
            t.zscale = zs.data();
            t.zzero = zz.data();
            t.n_tiles = nt;
            t.out = o;
            t.out_dtype = OUT_F64;
            for (int n = 0; n < nt; ++n) EXPECT(decode_tile(t, n) == 0, "tile %d of a good image flagged", n);
            for (size_t k = 0; k < img.size(); ++k) {
                const bool nan_in = img[k] != img[k];
                EXPECT(nan_in == (o[k] != o[k]), "NaN not kept at %zu", k);
                if (!nan_in) {
                    const double d = o[k] - img[k];
                    EXPECT((d < 0 ? -d : d) <= 0.5 * scale * (1 + 1e-9) + 1e-12, "pixel %zu off by %g (scale %g)", k, d, scale);
                    if (t.quantize == Q_DITHER_2 && img[k] == 0.0) EXPECT(o[k] == 0.0, "exact zero lost");
                }
            }
            // tables that point outside the heap or claim more bytes than there are
            for (int c = 0; c < 4; ++c) {
                const int n = (int)rnd(0, nt - 1);
                const int64_t o0 = offs[n];
                const int32_t n0 = nb[n];
                if (c == 0) offs[n] = (int64_t)heap.size() - rnd(0, 3);
                else if (c == 1) offs[n] = -rnd(1, 100);
                else if (c == 2) nb[n] = n0 + (int32_t)rnd(1, 1000) + (int32_t)heap.size();
                else nb[n] = (int32_t)rnd(-5, 0);
                const int e2 = decode_tile(t, n);
                EXPECT(e2 == 1 || e2 == 2, "lying tile table not flagged (case %d -> %d)", c, e2);
                offs[n] = o0;
                nb[n] = n0;
            }
            std::free(hp);
            std::free(o);
        }
    }
    if (failures) std::fprintf(stderr, "%d failure(s)\n", failures);
    else std::printf("ok: %ld iterations\n", iters);
    return failures ? 1 : 0;
}
