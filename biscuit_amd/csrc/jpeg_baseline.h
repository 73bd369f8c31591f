// jpeg_baseline.h -- baseline (SOF0/SOF1, 8-bit, Huffman, one interleaved scan) JPEG -> RGB for the tile reader.
//
// Slideflow writes tile TFRecords as PNG or JPEG (`img_format`); the reference's tiles are PNG (configure.py:118-124), so
// this is the second format in front of the staging kernel.  TensorFlow's decode_jpeg and Pillow both run libjpeg(-turbo)
// with its defaults -- the slow-but-accurate integer IDCT, "fancy" (triangle-filter) chroma upsampling, the 16-bit
// fixed-point YCbCr->RGB tables -- and a tile must come out with the same bytes here or the network sees another image.
// So the three arithmetic stages below restate those published algorithms step for step (the IJG "islow" IDCT of Loeffler,
// Ligtenberg & Moschytz with CONST_BITS 13 / PASS1_BITS 2; h2v1 and h2v2 triangle upsampling with IJG's alternating
// rounding and its edge replication; ITU-R BT.601 full-range conversion with SCALEBITS 16), and tests/test_jpeg.py holds
// them to Pillow's output bit for bit.  Everything this decoder does not cover -- progressive or arithmetic coding,
// 12-bit, CMYK / Adobe RGB, several scans, sampling other than 4:4:4 / 4:2:2 / 4:2:0, and any stream that is not clean
// (a code that does not exist, data missing before a marker, a restart marker out of place) -- is reported as
// "unsupported", and the caller decodes that record with its own decoder: libjpeg's recovery from damaged data is not
// something to imitate.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace bqjpg {

enum { OK = 0, UNSUPPORTED = 1, WRONG_SIZE = 2 };

static const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

constexpr int LOOK = 9;          // codes up to this length resolve in one table read
constexpr int FAST = 11;         // AC: code and value bits together, when they fit in this many

struct Huff {
    int16_t fast[1 << FAST];     // AC only: (value << 8) | (run << 4) | bits used; 0 = take the long way
    uint16_t look[1 << LOOK];    // (length << 8) | symbol, 0 = longer than LOOK bits
    int32_t maxcode[18];         // largest code of each length (-1: none), [17] = sentinel
    int32_t valoff[17];          // index of the first symbol of a length minus its first code
    uint8_t vals[256];
    bool defined = false;
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// Canonical code assignment (ITU-T T.81 Annex C).  false: the counts describe no prefix code.
inline bool build_huff(Huff& H, const uint8_t counts[16], const uint8_t* vals, int nvals, bool dc) {
    int code = 0, p = 0;
    memset(H.look, 0, sizeof(H.look));
    for (int l = 1; l <= 16; ++l) {
        const int n = counts[l - 1];
        if (n == 0) { H.maxcode[l] = -1; H.valoff[l] = 0; code <<= 1; continue; }
        if (code + n > (1 << l)) return false;
        H.valoff[l] = p - code;
        if (l <= LOOK)
            for (int i = 0; i < n; ++i) {
                const int first = (code + i) << (LOOK - l);
                for (int f = 0; f < (1 << (LOOK - l)); ++f) H.look[first + f] = (uint16_t)((l << 8) | vals[p + i]);
            }
        p += n; code += n;
        H.maxcode[l] = code - 1;
        code <<= 1;
    }
    if (p != nvals) return false;
    H.maxcode[17] = 0x7FFFFFFF;
    memcpy(H.vals, vals, (size_t)nvals);
    if (dc) {
        for (int i = 0; i < nvals; ++i)
            if (vals[i] > 15) return false;
    } else {
        // Most AC coefficients are a short code followed by a few value bits: resolve both with one read.
        for (int i = 0; i < (1 << FAST); ++i) {
            H.fast[i] = 0;
            const uint32_t e = H.look[i >> (FAST - LOOK)];
            if (!e) continue;
            const int l = (int)(e >> 8), run = (e >> 4) & 15, sz = e & 15;
            if (sz == 0) {                                  // end of block / sixteen zeros: no value bits
                if (run == 0 || run == 15) H.fast[i] = (int16_t)((run << 4) | l);
            } else if (l + sz <= FAST && sz <= 7) {
                const int v = extend((i >> (FAST - l - sz)) & ((1 << sz) - 1), sz);
                H.fast[i] = (int16_t)(v * 256 + ((run << 4) | (l + sz)));
            }
        }
    }
    H.defined = true;
    return true;
}

// Entropy-coded bytes of one restart interval with the stuffed zeros removed.  The accumulator is refilled eight bytes at
// a time whether or not they belong to the interval; `clean()` says afterwards whether a bit from beyond its end was
// consumed, which is what a damaged stream does.  A refill moves `p` by at most 7 bytes and a block refills at most 65
// times, so with ECS_PAD readable bytes behind the LAST interval and `overrun()` checked before every block the reader
// never leaves the buffer, however short a damaged scan is (a scan cut down to nothing in front of a valid EOI walked
// `p` hundreds of bytes per block past the data before `clean()` was ever asked: round-3 advisory).
struct Bits {
    const uint8_t *p, *start;
    uint64_t acc = 0;
    int have = 0;
    int64_t nbits = 0;
    void open(const uint8_t* data, size_t n) { p = start = data; acc = 0; have = 0; nbits = (int64_t)n * 8; }
    bool overrun() const { return (int64_t)(p - start) * 8 > nbits + 64; }   // already 8 bytes past the interval's end
    inline void fill() {         // to 56..63 bits
        uint64_t v;
        memcpy(&v, p, 8);
        acc |= __builtin_bswap64(v) >> have;
        p += (63 - have) >> 3;
        have |= 56;
    }
    inline uint32_t peek(int n) const { return (uint32_t)(acc >> (64 - n)); }
    inline void drop(int n) { acc <<= n; have -= n; }
    bool clean() const { return (int64_t)(p - start) * 8 - have <= nbits; }
};

constexpr size_t ECS_PAD = 65 * 7 + 8 + 16 + 64;   // one block's worth of refills behind an interval that just passed overrun()

inline int decode_symbol(Bits& B, const Huff& H) {
    const uint32_t e = H.look[B.peek(LOOK)];
    if (e) { B.drop(e >> 8); return e & 0xFF; }
    int l = LOOK + 1;
    int32_t code = (int32_t)B.peek(l);
    while (code > H.maxcode[l]) { ++l; if (l > 16) return -1; code = (int32_t)B.peek(l); }
    B.drop(l);
    return H.vals[(code + H.valoff[l]) & 0xFF];
}

// ---- inverse DCT (IJG jidctint "islow": 13-bit constants, 2 extra bits kept between the passes) ----------------------
constexpr int CB = 13, P1 = 2;
constexpr int32_t F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633,
                  F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

// One 1-D pass over 8 lanes: in[k][lane] -> out[k][lane], descaled by `shift` (even part exact, odd part per LL&M fig. 1).
template <int SHIFT, typename In, typename Store>
inline void idct_pass(const In (*in)[8], Store store) {
    for (int c = 0; c < 8; ++c) {
        int32_t z2 = in[2][c], z3 = in[6][c];
        int32_t z1 = (z2 + z3) * F_0_541;
        int32_t t2 = z1 - z3 * F_1_847, t3 = z1 + z2 * F_0_765;
        z2 = in[0][c]; z3 = in[4][c];
        int32_t t0 = (z2 + z3) * (1 << CB), t1 = (z2 - z3) * (1 << CB);
        const int32_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        t0 = in[7][c]; t1 = in[5][c]; t2 = in[3][c]; t3 = in[1][c];
        z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
        int32_t z4 = t1 + t3;
        const int32_t z5 = (z3 + z4) * F_1_175;
        t0 *= F_0_298; t1 *= F_2_053; t2 *= F_3_072; t3 *= F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
        constexpr int32_t R = 1 << (SHIFT - 1);
        store(0, c, (t10 + t3 + R) >> SHIFT); store(7, c, (t10 - t3 + R) >> SHIFT);
        store(1, c, (t11 + t2 + R) >> SHIFT); store(6, c, (t11 - t2 + R) >> SHIFT);
        store(2, c, (t12 + t1 + R) >> SHIFT); store(5, c, (t12 - t1 + R) >> SHIFT);
        store(3, c, (t13 + t0 + R) >> SHIFT); store(4, c, (t13 - t0 + R) >> SHIFT);
    }
}

// coef: dequantised, natural order [row][col]; out: 8 rows of 8 samples, `stride` apart
inline bool idct_islow_scalar(const int16_t coef[8][8], uint8_t* out, size_t stride) {
    int32_t ws[8][8], tr[8][8];
    bool ok = true;                      // see idct_islow below
    idct_pass<CB - P1>(coef, [&](int k, int c, int32_t v) { ws[k][c] = v; ok &= v >= -(1 << 14) && v < (1 << 14); });
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) tr[c][r] = ws[r][c];                                // rows: lane = row
    idct_pass<CB + P1 + 3>(tr, [&](int k, int r, int32_t v) {
        out[(size_t)r * stride + k] = clamp8(v + 128);
        ok &= v >= -512 && v < 512;
    });
    return ok;
}

#if defined(__x86_64__)
// The same arithmetic, eight lanes of 16 bits at a time.  Every output of the 1-D transform is a fixed integer
// combination of its inputs, so the sums-then-multiplies of the loop above can be regrouped into pmaddwd pairs
// (a*c1 + b*c2 in 32 bits) with the constants added up beforehand -- the products are exact either way, and the
// rounding happens once, where it did.  Between the passes the values are kept as 16 bits (as libjpeg-turbo's SIMD
// builds keep them); an 8-bit image's never need more.
#define BQJ_PAIR(c1, c2) _mm_set1_epi32((int)(((uint32_t)(uint16_t)(int16_t)(c2) << 16) | (uint16_t)(int16_t)(c1)))
template <int SHIFT>
inline void idct_pass_sse2(__m128i* r) {
    const __m128i rnd = _mm_set1_epi32(1 << (SHIFT - 1));
    // even part
    const __m128i p26l = _mm_unpacklo_epi16(r[2], r[6]), p26h = _mm_unpackhi_epi16(r[2], r[6]);
    const __m128i p04l = _mm_unpacklo_epi16(r[0], r[4]), p04h = _mm_unpackhi_epi16(r[0], r[4]);
    const __m128i k3 = BQJ_PAIR(F_0_541 + F_0_765, F_0_541), k2 = BQJ_PAIR(F_0_541, F_0_541 - F_1_847);
    const __m128i kp = BQJ_PAIR(1 << CB, 1 << CB), km = BQJ_PAIR(1 << CB, -(1 << CB));
    const __m128i t3l = _mm_madd_epi16(p26l, k3), t3h = _mm_madd_epi16(p26h, k3);
    const __m128i t2l = _mm_madd_epi16(p26l, k2), t2h = _mm_madd_epi16(p26h, k2);
    const __m128i t0l = _mm_add_epi32(_mm_madd_epi16(p04l, kp), rnd), t0h = _mm_add_epi32(_mm_madd_epi16(p04h, kp), rnd);
    const __m128i t1l = _mm_add_epi32(_mm_madd_epi16(p04l, km), rnd), t1h = _mm_add_epi32(_mm_madd_epi16(p04h, km), rnd);
    const __m128i e0l = _mm_add_epi32(t0l, t3l), e0h = _mm_add_epi32(t0h, t3h);       // tmp10 (+ rounding)
    const __m128i e3l = _mm_sub_epi32(t0l, t3l), e3h = _mm_sub_epi32(t0h, t3h);       // tmp13
    const __m128i e1l = _mm_add_epi32(t1l, t2l), e1h = _mm_add_epi32(t1h, t2h);       // tmp11
    const __m128i e2l = _mm_sub_epi32(t1l, t2l), e2h = _mm_sub_epi32(t1h, t2h);       // tmp12
    // odd part: inputs 7, 5, 3, 1 (tmp0..tmp3 of the loop above); each result = sum of four products
    const __m128i p75l = _mm_unpacklo_epi16(r[7], r[5]), p75h = _mm_unpackhi_epi16(r[7], r[5]);
    const __m128i p31l = _mm_unpacklo_epi16(r[3], r[1]), p31h = _mm_unpackhi_epi16(r[3], r[1]);
    constexpr int A = F_1_175, Z1 = -F_0_899, Z2 = -F_2_562, Z3 = -F_1_961, Z4 = -F_0_390;
    const __m128i a0 = BQJ_PAIR(F_0_298 + Z1 + Z3 + A, A), b0 = BQJ_PAIR(Z3 + A, Z1 + A);
    const __m128i a1 = BQJ_PAIR(A, F_2_053 + Z2 + Z4 + A), b1 = BQJ_PAIR(Z2 + A, Z4 + A);
    const __m128i a2 = BQJ_PAIR(Z3 + A, Z2 + A), b2 = BQJ_PAIR(F_3_072 + Z2 + Z3 + A, A);
    const __m128i a3 = BQJ_PAIR(Z1 + A, Z4 + A), b3 = BQJ_PAIR(A, F_1_501 + Z1 + Z4 + A);
    const __m128i o0l = _mm_add_epi32(_mm_madd_epi16(p75l, a0), _mm_madd_epi16(p31l, b0));
    const __m128i o0h = _mm_add_epi32(_mm_madd_epi16(p75h, a0), _mm_madd_epi16(p31h, b0));
    const __m128i o1l = _mm_add_epi32(_mm_madd_epi16(p75l, a1), _mm_madd_epi16(p31l, b1));
    const __m128i o1h = _mm_add_epi32(_mm_madd_epi16(p75h, a1), _mm_madd_epi16(p31h, b1));
    const __m128i o2l = _mm_add_epi32(_mm_madd_epi16(p75l, a2), _mm_madd_epi16(p31l, b2));
    const __m128i o2h = _mm_add_epi32(_mm_madd_epi16(p75h, a2), _mm_madd_epi16(p31h, b2));
    const __m128i o3l = _mm_add_epi32(_mm_madd_epi16(p75l, a3), _mm_madd_epi16(p31l, b3));
    const __m128i o3h = _mm_add_epi32(_mm_madd_epi16(p75h, a3), _mm_madd_epi16(p31h, b3));
#define BQJ_OUT(i, j, el, eh, ol, oh)                                                                         \
    r[i] = _mm_packs_epi32(_mm_srai_epi32(_mm_add_epi32(el, ol), SHIFT), _mm_srai_epi32(_mm_add_epi32(eh, oh), SHIFT)); \
    r[j] = _mm_packs_epi32(_mm_srai_epi32(_mm_sub_epi32(el, ol), SHIFT), _mm_srai_epi32(_mm_sub_epi32(eh, oh), SHIFT));
    BQJ_OUT(0, 7, e0l, e0h, o3l, o3h)
    BQJ_OUT(1, 6, e1l, e1h, o2l, o2h)
    BQJ_OUT(2, 5, e2l, e2h, o1l, o1h)
    BQJ_OUT(3, 4, e3l, e3h, o0l, o0h)
#undef BQJ_OUT
}
#undef BQJ_PAIR

inline void transpose8_epi16(__m128i* r) {
    const __m128i a0 = _mm_unpacklo_epi16(r[0], r[1]), a1 = _mm_unpackhi_epi16(r[0], r[1]);
    const __m128i a2 = _mm_unpacklo_epi16(r[2], r[3]), a3 = _mm_unpackhi_epi16(r[2], r[3]);
    const __m128i a4 = _mm_unpacklo_epi16(r[4], r[5]), a5 = _mm_unpackhi_epi16(r[4], r[5]);
    const __m128i a6 = _mm_unpacklo_epi16(r[6], r[7]), a7 = _mm_unpackhi_epi16(r[6], r[7]);
    const __m128i b0 = _mm_unpacklo_epi32(a0, a2), b1 = _mm_unpackhi_epi32(a0, a2);
    const __m128i b2 = _mm_unpacklo_epi32(a1, a3), b3 = _mm_unpackhi_epi32(a1, a3);
    const __m128i b4 = _mm_unpacklo_epi32(a4, a6), b5 = _mm_unpackhi_epi32(a4, a6);
    const __m128i b6 = _mm_unpacklo_epi32(a5, a7), b7 = _mm_unpackhi_epi32(a5, a7);
    r[0] = _mm_unpacklo_epi64(b0, b4); r[1] = _mm_unpackhi_epi64(b0, b4);
    r[2] = _mm_unpacklo_epi64(b1, b5); r[3] = _mm_unpackhi_epi64(b1, b5);
    r[4] = _mm_unpacklo_epi64(b2, b6); r[5] = _mm_unpackhi_epi64(b2, b6);
    r[6] = _mm_unpacklo_epi64(b3, b7); r[7] = _mm_unpackhi_epi64(b3, b7);
}

// false: the block leaves the range an 8-bit image's blocks stay in (intermediate beyond 15 bits, sample beyond 10).  Out
// there libjpeg's builds stop agreeing with each other -- the C code wraps its range-limit table, the SIMD code wraps
// 16-bit sums -- so there is no "same bytes" to produce, and the stream goes back to the caller.
inline bool idct_islow(const int16_t coef[8][8], uint8_t* out, size_t stride) {
    __m128i r[8];
    for (int i = 0; i < 8; ++i) r[i] = _mm_load_si128(reinterpret_cast<const __m128i*>(coef[i]));
    idct_pass_sse2<CB - P1>(r);          // down the columns (lane = column)
    const __m128i b14 = _mm_set1_epi16(1 << 14), b9 = _mm_set1_epi16(1 << 9);
    __m128i wide = _mm_setzero_si128();  // sign bit set in a lane: some |value| > 2^14
    for (int i = 0; i < 8; ++i) wide = _mm_or_si128(wide, _mm_add_epi16(r[i], b14));
    transpose8_epi16(r);
    idct_pass_sse2<CB + P1 + 3>(r);      // along the rows (lane = row, r[k] = output column k)
    __m128i far = _mm_setzero_si128();   // bits 10..15 set in a lane: some sample outside -512..511
    for (int i = 0; i < 8; ++i) far = _mm_or_si128(far, _mm_add_epi16(r[i], b9));
    if ((_mm_movemask_epi8(wide) & 0xAAAA) | _mm_movemask_epi8(_mm_cmpgt_epi16(_mm_srli_epi16(far, 10), _mm_setzero_si128())))
        return false;
    transpose8_epi16(r);
    const __m128i bias = _mm_set1_epi8((char)0x80);
    for (int i = 0; i < 8; i += 2) {
        const __m128i v = _mm_add_epi8(_mm_packs_epi16(r[i], r[i + 1]), bias);          // saturate to -128..127, then + 128
        _mm_storel_epi64(reinterpret_cast<__m128i*>(out + (size_t)i * stride), v);
        _mm_storel_epi64(reinterpret_cast<__m128i*>(out + (size_t)(i + 1) * stride), _mm_srli_si128(v, 8));
    }
    return true;
}
#else
inline bool idct_islow(const int16_t coef[8][8], uint8_t* out, size_t stride) { return idct_islow_scalar(coef, out, stride); }
#endif

// ---- the decoder ------------------------------------------------------------------------------------------------------
struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int stride = 0, rows = 0;            // plane size (whole MCUs)
    int dw = 0, dh = 0;                  // samples that exist: ceil(W*h/hmax), ceil(H*v/vmax)
    int pred = 0;
    std::vector<uint8_t> plane;
};

struct Scratch {
    Huff dc[4], ac[4];
    uint16_t q[4][64];                   // natural order
    bool qdef[4];
    Comp comp[3];
    std::vector<uint8_t> ecs;            // unstuffed entropy-coded data
    std::vector<size_t> seg;             // start of each restart interval in ecs (+ end)
    std::vector<int16_t> sum[2];         // 3*near + far per chroma column
    std::vector<uint8_t> up[2];          // one upsampled chroma row each
    int32_t cr_r[256], cb_b[256], cr_g[256], cb_g[256];
    bool tables = false;
};

inline uint32_t be16(const uint8_t* p) { return ((uint32_t)p[0] << 8) | p[1]; }

inline void color_tables(Scratch& S) {
    if (S.tables) return;
    for (int i = 0; i < 256; ++i) {
        const int x = i - 128;
        S.cr_r[i] = (91881 * x + 32768) >> 16;           // 1.40200
        S.cb_b[i] = (116130 * x + 32768) >> 16;          // 1.77200
        S.cr_g[i] = -46802 * x;                          // 0.71414
        S.cb_g[i] = -22554 * x + 32768;                  // 0.34414, carries the rounding for the sum
    }
    S.tables = true;
}

// Triangle-filter upsampling, 1 -> 2 samples along a row: each output is 3/4 of the nearer input and 1/4 of the further one,
// rounded alternately down and up so that the pair does not drift; the first and last outputs have no further neighbour
// and copy the nearer.  Vertically (h2v2) the same weights are applied to the rows first, at full precision, and the
// result is rounded once: w[i] = 3 near[i] + far[i], out = (3 w[i] + w[i -+ 1] + 8 | 7) >> 4.  Writing the edge rule as
// "the missing neighbour is the sample itself" gives the same bytes ((4 x + 1) >> 2 = x, (4 w + 8) >> 4 as specified), so
// the row is padded by one on each side and one loop serves every position.
//   w: n values with w[-1] and w[n] writable; out: 2 n bytes (+ up to 16 of slack).
template <int SHIFT, int RND_EVEN, int RND_ODD>
inline void triangle_row(int16_t* w, int n, uint8_t* out) {
    w[-1] = w[0];
    w[n] = w[n - 1];
    int i = 0;
#if defined(__x86_64__)
    const __m128i re = _mm_set1_epi16(RND_EVEN), ro = _mm_set1_epi16(RND_ODD);
    for (; i < n; i += 8) {              // reads w[i-1 .. i+8], writes out[2i .. 2i+15]: both buffers carry the slack
        const __m128i t = _mm_loadu_si128(reinterpret_cast<const __m128i*>(w + i));
        const __m128i t3 = _mm_add_epi16(_mm_add_epi16(t, t), t);
        const __m128i e = _mm_srli_epi16(_mm_add_epi16(_mm_add_epi16(t3, _mm_loadu_si128(reinterpret_cast<const __m128i*>(w + i - 1))), re), SHIFT);
        const __m128i o = _mm_srli_epi16(_mm_add_epi16(_mm_add_epi16(t3, _mm_loadu_si128(reinterpret_cast<const __m128i*>(w + i + 1))), ro), SHIFT);
        _mm_storeu_si128(reinterpret_cast<__m128i*>(out + 2 * i), _mm_unpacklo_epi8(_mm_packus_epi16(e, e), _mm_packus_epi16(o, o)));
    }
#else
    for (; i < n; ++i) {
        out[2 * i] = (uint8_t)((3 * w[i] + w[i - 1] + RND_EVEN) >> SHIFT);
        out[2 * i + 1] = (uint8_t)((3 * w[i] + w[i + 1] + RND_ODD) >> SHIFT);
    }
#endif
}

inline void upsample_h2(const uint8_t* in, int n, int16_t* w, uint8_t* out) {
    for (int i = 0; i < n; ++i) w[i] = in[i];
    triangle_row<2, 1, 2>(w, n, out);
}

inline void upsample_h2v2(const uint8_t* near, const uint8_t* far, int n, int16_t* w, uint8_t* out) {
    for (int i = 0; i < n; ++i) w[i] = (int16_t)(near[i] * 3 + far[i]);
    triangle_row<4, 8, 7>(w, n, out);
}

// YCbCr (ITU-R BT.601, full range, chroma centred on 128) -> RGB in 16-bit fixed point, one table read per term:
//   R = Y + 1.40200 Cr,  G = Y - 0.34414 Cb - 0.71414 Cr,  B = Y + 1.77200 Cb, constants scaled by 2^16 and rounded,
//   R and B rounded per term, G once for the sum.
inline void ycc_row_scalar(const Scratch& S, const uint8_t* yy, const uint8_t* cb, const uint8_t* cr, int x0, int x1, uint8_t* o) {
    for (int x = x0; x < x1; ++x) {
        const int Y = yy[x];
        o[3 * x] = clamp8(Y + S.cr_r[cr[x]]);
        o[3 * x + 1] = clamp8(Y + ((S.cb_g[cb[x]] + S.cr_g[cr[x]]) >> 16));
        o[3 * x + 2] = clamp8(Y + S.cb_b[cb[x]]);
    }
}

#if defined(__x86_64__)
// The same values without tables, 16 pixels a step.  With x = C - 128 in 16-bit lanes:
//   (91881 x + 32768) >> 16  = x   + ((26345 x + 32768) >> 16)            (91881  = 65536 + 26345)
//   (116130 x + 32768) >> 16 = 2 x + ((-14942 x + 32768) >> 16)           (116130 = 131072 - 14942)
//   (c x + 32768) >> 16      = (mulhi(2 x, c) + 1) >> 1                    (floor of a floor)
//   (-22554 cb - 46802 cr + 32768) >> 16 = -cr + ((-22554 cb + 18734 cr + 32768) >> 16), the sum by pmaddwd in 32 bits.
__attribute__((target("ssse3"))) inline void ycc_row_ssse3(const uint8_t* yy, const uint8_t* cb, const uint8_t* cr, int W,
                                                             uint8_t* o) {
    const __m128i zero = _mm_setzero_si128(), c128 = _mm_set1_epi16(128), one = _mm_set1_epi16(1);
    const __m128i k_r = _mm_set1_epi16(26345), k_b = _mm_set1_epi16(-14942);
    const __m128i k_g = _mm_set1_epi32((int)(((uint32_t)18734 << 16) | (uint16_t)(int16_t)-22554));
    const __m128i half = _mm_set1_epi32(32768);
    // byte i of R, G, B -> bytes 3i, 3i+1, 3i+2 of the 48 output bytes
    const __m128i r0 = _mm_setr_epi8(0, -1, -1, 1, -1, -1, 2, -1, -1, 3, -1, -1, 4, -1, -1, 5);
    const __m128i g0 = _mm_setr_epi8(-1, 0, -1, -1, 1, -1, -1, 2, -1, -1, 3, -1, -1, 4, -1, -1);
    const __m128i b0 = _mm_setr_epi8(-1, -1, 0, -1, -1, 1, -1, -1, 2, -1, -1, 3, -1, -1, 4, -1);
    const __m128i r1 = _mm_setr_epi8(-1, -1, 6, -1, -1, 7, -1, -1, 8, -1, -1, 9, -1, -1, 10, -1);
    const __m128i g1 = _mm_setr_epi8(5, -1, -1, 6, -1, -1, 7, -1, -1, 8, -1, -1, 9, -1, -1, 10);
    const __m128i b1 = _mm_setr_epi8(-1, 5, -1, -1, 6, -1, -1, 7, -1, -1, 8, -1, -1, 9, -1, -1);
    const __m128i r2 = _mm_setr_epi8(-1, 11, -1, -1, 12, -1, -1, 13, -1, -1, 14, -1, -1, 15, -1, -1);
    const __m128i g2 = _mm_setr_epi8(-1, -1, 11, -1, -1, 12, -1, -1, 13, -1, -1, 14, -1, -1, 15, -1);
    const __m128i b2 = _mm_setr_epi8(10, -1, -1, 11, -1, -1, 12, -1, -1, 13, -1, -1, 14, -1, -1, 15);
    auto half8 = [&](__m128i y, __m128i b, __m128i r, __m128i& R, __m128i& G, __m128i& B) {       // 8 pixels, 16-bit lanes
        b = _mm_sub_epi16(b, c128); r = _mm_sub_epi16(r, c128);
        const __m128i r2x = _mm_add_epi16(r, r), b2x = _mm_add_epi16(b, b);
        R = _mm_add_epi16(_mm_add_epi16(y, r), _mm_srai_epi16(_mm_add_epi16(_mm_mulhi_epi16(r2x, k_r), one), 1));
        B = _mm_add_epi16(_mm_add_epi16(y, b2x), _mm_srai_epi16(_mm_add_epi16(_mm_mulhi_epi16(b2x, k_b), one), 1));
        const __m128i gl = _mm_srai_epi32(_mm_add_epi32(_mm_madd_epi16(_mm_unpacklo_epi16(b, r), k_g), half), 16);
        const __m128i gh = _mm_srai_epi32(_mm_add_epi32(_mm_madd_epi16(_mm_unpackhi_epi16(b, r), k_g), half), 16);
        G = _mm_add_epi16(_mm_sub_epi16(y, r), _mm_packs_epi32(gl, gh));
    };
    for (int x = 0;; x += 16) {
        if (x > W - 16) {                 // the last step overlaps the one before it
            if (x >= W) break;
            x = W - 16;
        }
        const __m128i y = _mm_loadu_si128(reinterpret_cast<const __m128i*>(yy + x));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(cb + x));
        const __m128i r = _mm_loadu_si128(reinterpret_cast<const __m128i*>(cr + x));
        __m128i Rl, Gl, Bl, Rh, Gh, Bh;
        half8(_mm_unpacklo_epi8(y, zero), _mm_unpacklo_epi8(b, zero), _mm_unpacklo_epi8(r, zero), Rl, Gl, Bl);
        half8(_mm_unpackhi_epi8(y, zero), _mm_unpackhi_epi8(b, zero), _mm_unpackhi_epi8(r, zero), Rh, Gh, Bh);
        const __m128i R = _mm_packus_epi16(Rl, Rh), G = _mm_packus_epi16(Gl, Gh), B = _mm_packus_epi16(Bl, Bh);
        uint8_t* d = o + 3 * (size_t)x;
        _mm_storeu_si128(reinterpret_cast<__m128i*>(d),
                         _mm_or_si128(_mm_or_si128(_mm_shuffle_epi8(R, r0), _mm_shuffle_epi8(G, g0)), _mm_shuffle_epi8(B, b0)));
        _mm_storeu_si128(reinterpret_cast<__m128i*>(d + 16),
                         _mm_or_si128(_mm_or_si128(_mm_shuffle_epi8(R, r1), _mm_shuffle_epi8(G, g1)), _mm_shuffle_epi8(B, b1)));
        _mm_storeu_si128(reinterpret_cast<__m128i*>(d + 32),
                         _mm_or_si128(_mm_or_si128(_mm_shuffle_epi8(R, r2), _mm_shuffle_epi8(G, g2)), _mm_shuffle_epi8(B, b2)));
    }
}
#endif

inline void ycc_row(const Scratch& S, const uint8_t* yy, const uint8_t* cb, const uint8_t* cr, int W, uint8_t* o) {
#if defined(__x86_64__)
    static const bool have = __builtin_cpu_supports("ssse3");
    if (have && W >= 16) { ycc_row_ssse3(yy, cb, cr, W, o); return; }
#endif
    ycc_row_scalar(S, yy, cb, cr, 0, W, o);
}

// One 8x8 block: Huffman-decode, dequantise, inverse-transform into the component plane.  false: not a clean stream
// (or coefficients no 8-bit image produces, where libjpeg's builds differ in how they wrap).
inline bool decode_block(Bits& B, const Huff& dc, const Huff& ac, const uint16_t* q, int& pred, uint8_t* out, size_t stride) {
    alignas(16) int16_t coef[8][8];
    int16_t* cf = &coef[0][0];
    if (B.overrun()) return false;       // the stream ended blocks ago: refuse before reading further (see Bits)
    B.fill();
    int s = decode_symbol(B, dc);
    if (s < 0) return false;
    if (s) { const int r = (int)B.peek(s); B.drop(s); pred += extend(r, s); }
    int last = 0;
    memset(coef, 0, sizeof(coef));
    int32_t dq = pred * q[0];
    int32_t big = dq + 16384;            // collects any product outside 15 bits
    cf[0] = (int16_t)dq;
    for (int k = 1; k < 64;) {
        B.fill();
        const int f = ac.fast[B.peek(FAST)];
        int v;
        if (f) {
            B.drop(f & 15);
            v = f >> 8;
            k += (f >> 4) & 15;
            if (v == 0) {                                   // no value: sixteen zeros (k advanced by 15 already) or the end
                if ((f & 0xF0) == 0) break;
                ++k;
                continue;
            }
        } else {
            const int rs = decode_symbol(B, ac);
            if (rs < 0) return false;
            s = rs & 15;
            if (s == 0) {
                if ((rs >> 4) != 15) break;
                k += 16;
                continue;
            }
            k += rs >> 4;
            v = extend((int)B.peek(s), s);
            B.drop(s);
        }
        if (k > 63) return false;
        const int nat = ZIGZAG[k];
        dq = v * q[nat];
        big |= dq + 16384;
        cf[nat] = (int16_t)dq;
        last = k++;
    }
    if (big & ~0x7FFF) return false;
    if (last == 0) {                     // DC only: both passes reduce to one descale
        const uint8_t v = clamp8(((cf[0] * 4 + 16) >> 5) + 128);
        for (int r = 0; r < 8; ++r) memset(out + (size_t)r * stride, v, 8);
        return cf[0] >= -4096 && cf[0] < 4096;              // samples within -512..511, as idct_islow asks
    }
    return idct_islow(coef, out, stride);
}

// data[0..n): a whole JPEG file.  out: px*px*3 RGB bytes.  probe_only: stop in front of the entropy decoder -- everything the
// markers and the scan's structure can refuse (progressive / arithmetic / lossless frames, sampling factors, colour spaces, restart
// structure, size) has been checked by then, at the cost of one pass over the bytes; `out` is not written.
inline int decode(const uint8_t* data, size_t n, int px, uint8_t* out, Scratch& S, bool probe_only = false) {
    if (n < 4 || data[0] != 0xFF || data[1] != 0xD8) return UNSUPPORTED;
    color_tables(S);
    for (int i = 0; i < 4; ++i) { S.dc[i].defined = S.ac[i].defined = false; S.qdef[i] = false; }
    size_t p = 2;
    int W = 0, H = 0, ncomp = 0, restart = 0;
    bool jfif = false, adobe = false, sof = false;
    int adobe_transform = 0;
    size_t scan = 0;
    while (!scan) {
        if (p + 4 > n || data[p] != 0xFF) return UNSUPPORTED;
        while (p < n && data[p] == 0xFF) ++p;              // fill bytes before a marker are legal
        if (p >= n) return UNSUPPORTED;
        const int m = data[p++];
        if (m < 0xC0 || (m >= 0xD0 && m <= 0xD9) || p + 2 > n) return UNSUPPORTED;      // nothing that belongs in a header
        const size_t len = be16(data + p);
        if (len < 2 || p + len > n) return UNSUPPORTED;
        const uint8_t* d = data + p + 2;
        const size_t dl = len - 2;
        switch (m) {
            case 0xC0: case 0xC1: {                         // baseline / extended sequential, Huffman
                if (sof || dl < 6 || d[0] != 8) return UNSUPPORTED;
                H = (int)be16(d + 1); W = (int)be16(d + 3); ncomp = d[5];
                if ((ncomp != 1 && ncomp != 3) || dl != (size_t)(6 + 3 * ncomp) || W == 0 || H == 0) return UNSUPPORTED;
                for (int c = 0; c < ncomp; ++c) {
                    Comp& K = S.comp[c];
                    K.id = d[6 + 3 * c]; K.h = d[7 + 3 * c] >> 4; K.v = d[7 + 3 * c] & 15; K.tq = d[8 + 3 * c];
                    if (K.tq > 3) return UNSUPPORTED;
                }
                sof = true;
                break;
            }
            case 0xC4: {                                    // Huffman tables
                size_t o = 0;
                while (o < dl) {
                    if (o + 17 > dl) return UNSUPPORTED;
                    const int tc = d[o] >> 4, th = d[o] & 15;
                    int nv = 0;
                    for (int i = 0; i < 16; ++i) nv += d[o + 1 + i];
                    if (tc > 1 || th > 3 || nv > 256 || o + 17 + (size_t)nv > dl) return UNSUPPORTED;
                    if (!build_huff(tc ? S.ac[th] : S.dc[th], d + o + 1, d + o + 17, nv, tc == 0)) return UNSUPPORTED;
                    o += 17 + (size_t)nv;
                }
                break;
            }
            case 0xDB: {                                    // quantisation tables
                size_t o = 0;
                while (o < dl) {
                    const int pq = d[o] >> 4, tq = d[o] & 15;
                    if (pq != 0 || tq > 3 || o + 65 > dl) return UNSUPPORTED;
                    for (int k = 0; k < 64; ++k) S.q[tq][ZIGZAG[k]] = d[o + 1 + k];
                    S.qdef[tq] = true;
                    o += 65;
                }
                break;
            }
            case 0xDD:
                if (dl != 2) return UNSUPPORTED;
                restart = (int)be16(d);
                break;
            case 0xE0:
                if (dl >= 5 && !memcmp(d, "JFIF", 5)) jfif = true;
                break;
            case 0xEE:
                if (dl >= 12 && !memcmp(d, "Adobe", 5)) { adobe = true; adobe_transform = d[11]; }
                break;
            case 0xDA: {                                    // the scan
                if (!sof || dl != (size_t)(4 + 2 * ncomp) || d[0] != ncomp) return UNSUPPORTED;
                for (int c = 0; c < ncomp; ++c) {
                    if (d[1 + 2 * c] != S.comp[c].id) return UNSUPPORTED;           // components in frame order
                    S.comp[c].td = d[2 + 2 * c] >> 4; S.comp[c].ta = d[2 + 2 * c] & 15;
                    if (S.comp[c].td > 3 || S.comp[c].ta > 3) return UNSUPPORTED;
                }
                if (d[1 + 2 * ncomp] != 0 || d[2 + 2 * ncomp] != 63 || d[3 + 2 * ncomp] != 0) return UNSUPPORTED;
                scan = p + len;
                break;
            }
            case 0xC2: case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE:
            case 0xCF: case 0xCC: case 0xDC:
                return UNSUPPORTED;                         // progressive, lossless, arithmetic, DNL
            default: break;                                 // APPn, COM: skipped
        }
        p += len;
    }
    if (W != px || H != px) return WRONG_SIZE;
    // colour space: what libjpeg would assume for these markers
    if (ncomp == 3) {
        if (!jfif && adobe && adobe_transform != 1) return UNSUPPORTED;
        if (!jfif && !adobe && S.comp[0].id == 'R' && S.comp[1].id == 'G' && S.comp[2].id == 'B') return UNSUPPORTED;
    }
    int hmax = 1, vmax = 1;
    if (ncomp == 1) { S.comp[0].h = S.comp[0].v = 1; }
    else {
        hmax = S.comp[0].h; vmax = S.comp[0].v;
        if (hmax < 1 || hmax > 2 || vmax < 1 || vmax > 2 || (hmax == 1 && vmax == 2)) return UNSUPPORTED;
        for (int c = 1; c < 3; ++c)
            if (S.comp[c].h != 1 || S.comp[c].v != 1) return UNSUPPORTED;
    }
    const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
    for (int c = 0; c < ncomp; ++c) {
        Comp& K = S.comp[c];
        if (!S.qdef[K.tq] || !S.dc[K.td].defined || !S.ac[K.ta].defined) return UNSUPPORTED;
        K.stride = mcux * K.h * 8; K.rows = mcuy * K.v * 8;
        K.dw = (W * K.h + hmax - 1) / hmax; K.dh = (H * K.v + vmax - 1) / vmax;
        K.plane.resize((size_t)K.stride * K.rows + 16);             // the row loops read whole vectors
        K.pred = 0;
    }
    if (ncomp == 3 && hmax == 2 && S.comp[1].dw <= 2) return UNSUPPORTED;       // libjpeg replicates instead of filtering

    // ---- entropy-coded segment: drop the stuffed zeros, split at the restart markers ---------------------------------
    S.ecs.clear(); S.seg.clear();
    S.ecs.reserve(n - scan + ECS_PAD);
    S.seg.push_back(0);
    {
        size_t i = scan;
        int expect = 0;
        bool ended = false;
        while (i < n) {
            const uint8_t* f = (const uint8_t*)memchr(data + i, 0xFF, n - i);
            const size_t j = f ? (size_t)(f - data) : n;
            S.ecs.insert(S.ecs.end(), data + i, data + j);
            if (!f || j + 1 >= n) return UNSUPPORTED;                              // ran off the end without EOI
            const int m = data[j + 1];
            if (m == 0x00) { S.ecs.push_back(0xFF); i = j + 2; }
            else if (m == 0xFF) return UNSUPPORTED;         // fill bytes inside the scan: libjpeg's two decode loops part ways
            else if (m >= 0xD0 && m <= 0xD7) {
                if (!restart || m != 0xD0 + expect) return UNSUPPORTED;
                expect = (expect + 1) & 7;
                S.seg.push_back(S.ecs.size());
                i = j + 2;
            } else if (m == 0xD9) { ended = true; break; }
            else return UNSUPPORTED;                                                // another scan or table: not baseline-simple
        }
        if (!ended) return UNSUPPORTED;
    }
    S.seg.push_back(S.ecs.size());
    S.ecs.insert(S.ecs.end(), ECS_PAD, (uint8_t)0);
    const int64_t total_mcu = (int64_t)mcux * mcuy;
    const int64_t per_seg = restart ? restart : total_mcu;
    if ((int64_t)(S.seg.size() - 1) != (total_mcu + per_seg - 1) / per_seg) return UNSUPPORTED;
    if (probe_only) return OK;

    Bits B;
    int64_t mcu = 0;
    for (size_t sg = 0; sg + 1 < S.seg.size(); ++sg) {
        B.open(S.ecs.data() + S.seg[sg], S.seg[sg + 1] - S.seg[sg]);
        for (int c = 0; c < ncomp; ++c) S.comp[c].pred = 0;
        const int64_t stop = mcu + per_seg < total_mcu ? mcu + per_seg : total_mcu;
        for (; mcu < stop; ++mcu) {
            const int mx = (int)(mcu % mcux), my = (int)(mcu / mcux);
            for (int c = 0; c < ncomp; ++c) {
                Comp& K = S.comp[c];
                for (int by = 0; by < K.v; ++by)
                    for (int bx = 0; bx < K.h; ++bx) {
                        uint8_t* o = K.plane.data() + (size_t)(my * K.v + by) * 8 * K.stride + (size_t)(mx * K.h + bx) * 8;
                        if (!decode_block(B, S.dc[K.td], S.ac[K.ta], S.q[K.tq], K.pred, o, (size_t)K.stride)) return UNSUPPORTED;
                    }
            }
        }
        if (!B.clean()) return UNSUPPORTED;
    }

    // ---- upsample + colour, one output row at a time -------------------------------------------------------------------
    if (ncomp == 1) {
        for (int y = 0; y < H; ++y) {
            const uint8_t* g = S.comp[0].plane.data() + (size_t)y * S.comp[0].stride;
            uint8_t* o = out + (size_t)y * W * 3;
            for (int x = 0; x < W; ++x) { o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = g[x]; }
        }
        return OK;
    }
    const int cw = S.comp[1].dw, ch = S.comp[1].dh;
    for (int c = 0; c < 2; ++c) { S.up[c].resize((size_t)2 * cw + 32); S.sum[c].resize((size_t)cw + 32); }
    for (int y = 0; y < H; ++y) {
        const uint8_t* yy = S.comp[0].plane.data() + (size_t)y * S.comp[0].stride;
        const uint8_t* cc[2];
        for (int c = 0; c < 2; ++c) {
            const Comp& K = S.comp[1 + c];
            if (hmax == 1) {
                cc[c] = K.plane.data() + (size_t)y * K.stride;
            } else if (vmax == 1) {
                upsample_h2(K.plane.data() + (size_t)y * K.stride, cw, S.sum[c].data() + 8, S.up[c].data());
                cc[c] = S.up[c].data();
            } else {
                const int nr = y >> 1;
                int fr = (y & 1) ? nr + 1 : nr - 1;              // the row above for the upper output row, below for the lower
                fr = fr < 0 ? 0 : fr > ch - 1 ? ch - 1 : fr;     // past the first / last real row: that row again
                upsample_h2v2(K.plane.data() + (size_t)nr * K.stride, K.plane.data() + (size_t)fr * K.stride, cw,
                              S.sum[c].data() + 8, S.up[c].data());
                cc[c] = S.up[c].data();
            }
        }
        ycc_row(S, yy, cc[0], cc[1], W, out + (size_t)y * W * 3);
    }
    return OK;
}

}  // namespace bqjpg
