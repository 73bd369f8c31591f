// zlib-stream (RFC 1950 / 1951) decompressor for the tile reader: PNG tiles are ~270 KB of scanlines each and zlib's
// inflate() was ~85 % of the decode time of a tile (1.3 ms of 1.5 ms on the bench host; the GPU consumes a tile every
// 40 us).  Same results as zlib's uncompress() -- every stream it accepts gives the same bytes, a stream it rejects is
// rejected (the Adler-32 trailer is checked) -- at 3-5x its speed on these streams:
//   * 64-bit bit buffer, refilled with one unaligned 8-byte load (the caller pads the input with 16 zero bytes);
//   * two-level decode tables: 11 bits direct for literal/length codes, 8 bits for distance codes, a table entry carries
//     the code length, the extra-bit count and the base value, so a symbol is one load, one shift and one add;
//   * literals are stored straight from the entry, up to three per refill;
//   * matches are copied in 8-byte steps (the output buffer has 8 bytes of slack, checked against) unless the distance
//     is shorter than that.
// Written from RFC 1951; no code taken from zlib or libdeflate.  Host code only.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace bqinf {

constexpr int LT_BITS = 11, DT_BITS = 8;                 // primary table widths
constexpr int LT_SIZE = (1 << LT_BITS) + 1024;           // + room for the sub-tables of codes longer than LT_BITS
constexpr int DT_SIZE = (1 << DT_BITS) + 512;
constexpr uint32_t F_LITERAL = 0x2000u, F_SUB = 0x4000u, F_SPECIAL = 0x8000u;   // SPECIAL: value 0 = end of block, 1 = invalid
constexpr int OUT_SLACK = 8;                             // bytes the caller provides behind the expected output size

struct Tables {
    uint32_t lt[LT_SIZE];
    uint32_t dt[DT_SIZE];
};

// entry = value << 16 | flags | extra << 8 | nbits      (nbits: bits of the code itself; sub-table pointer: the primary width)
inline uint32_t entry(unsigned value, uint32_t flags, unsigned extra, unsigned nbits) {
    return ((uint32_t)value << 16) | flags | ((uint32_t)extra << 8) | nbits;
}

inline unsigned reverse_bits(unsigned code, int len) {
    unsigned r = 0;
    for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1); code >>= 1; }
    return r;
}

// Canonical Huffman decode table from code lengths (RFC 1951 3.2.2).  sym_entry(sym) gives the entry without its length.
// Returns false for an over-subscribed set of lengths, for one that does not fit the table, and -- as zlib does -- for an
// incomplete set unless it is empty or a single code of length 1 in a literal/length or distance table (unused codes
// decode as invalid).
template <typename F>
inline bool build_table(const uint8_t* lens, int nsym, uint32_t* table, int table_bits, int table_cap, bool precode,
                        F&& sym_entry) {
    int count[16] = {0};
    for (int s = 0; s < nsym; ++s) ++count[lens[s]];
    count[0] = 0;
    unsigned next_code[16];
    {
        unsigned code = 0;
        long long space = 1;                              // Kraft: every length must leave room
        for (int l = 1; l <= 15; ++l) {
            space = space * 2 - count[l];
            if (space < 0) return false;                  // over-subscribed
            code = (code + (unsigned)count[l - 1]) << 1;
            next_code[l] = code;
        }
        if (space > 0) {                                  // incomplete
            int total = 0, longest = 0;
            for (int l = 1; l <= 15; ++l) if (count[l]) { total += count[l]; longest = l; }
            if (total != 0 && (precode || longest != 1)) return false;
        }
    }
    const uint32_t invalid = entry(1, F_SPECIAL, 0, 1);
    const int primary = 1 << table_bits;
    for (int i = 0; i < primary; ++i) table[i] = invalid;
    // sub-table sizes: for every primary prefix of a long code, the longest code below it
    uint8_t sub_bits[1 << LT_BITS];                       // (DT_BITS <= LT_BITS)
    bool any_long = false;
    for (int l = table_bits + 1; l <= 15; ++l) any_long |= count[l] != 0;
    if (any_long) memset(sub_bits, 0, (size_t)primary);
    unsigned codes[320];
    for (int s = 0; s < nsym; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const unsigned rev = reverse_bits(next_code[l]++, l);
        codes[s] = rev;
        if (l > table_bits) {
            const unsigned prefix = rev & (primary - 1);
            if (l - table_bits > sub_bits[prefix]) sub_bits[prefix] = (uint8_t)(l - table_bits);
        }
    }
    int used = primary;
    if (any_long) {
        for (int pfx = 0; pfx < primary; ++pfx) {
            if (!sub_bits[pfx]) continue;
            const int size = 1 << sub_bits[pfx];
            if (used + size > table_cap) return false;
            table[pfx] = entry((unsigned)used, F_SUB, sub_bits[pfx], (unsigned)table_bits);
            for (int i = 0; i < size; ++i) table[used + i] = invalid;
            used += size;
        }
    }
    for (int s = 0; s < nsym; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const unsigned rev = codes[s];
        const uint32_t e = sym_entry(s) | (uint32_t)l;
        if (l <= table_bits) {
            for (unsigned i = rev; i < (unsigned)primary; i += 1u << l) table[i] = e;
        } else {
            const unsigned prefix = rev & (primary - 1);
            const uint32_t pe = table[prefix];
            const unsigned base = pe >> 16, sb = (pe >> 8) & 0x1F;
            for (unsigned i = rev >> table_bits; i < (1u << sb); i += 1u << (l - table_bits)) table[base + i] = e;
        }
    }
    return true;
}

inline uint32_t litlen_entry(int sym) {
    static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    if (sym < 256) return entry((unsigned)sym, F_LITERAL, 0, 0);
    if (sym == 256) return entry(0, F_SPECIAL, 0, 0);
    if (sym > 285) return entry(1, F_SPECIAL, 0, 0);
    return entry(base[sym - 257], 0, extra[sym - 257], 0);
}

inline uint32_t dist_entry(int sym) {
    static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
                                      4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (sym > 29) return entry(1, F_SPECIAL, 0, 0);
    return entry(base[sym], 0, extra[sym], 0);
}

inline uint32_t adler32_scalar(const uint8_t* p, size_t n, uint32_t a, uint32_t b) {
    while (n) {
        size_t k = n < 5552 ? n : 5552;                   // the largest run that cannot overflow 32 bits
        n -= k;
        while (k >= 8) {
            a += p[0]; b += a; a += p[1]; b += a; a += p[2]; b += a; a += p[3]; b += a;
            a += p[4]; b += a; a += p[5]; b += a; a += p[6]; b += a; a += p[7]; b += a;
            p += 8; k -= 8;
        }
        while (k--) { a += *p++; b += a; }
        a %= 65521u; b %= 65521u;
    }
    return (b << 16) | a;
}

#if defined(__x86_64__)
// 16 bytes per step: a += sum(x); b += 16 a_before + sum((16 - i) x_i), the weighted sum by pmaddubsw.  ~0.25 cycles per
// byte against ~1.3 for the loop above (268 KB per tile: 8 % of a tile's decode time).
__attribute__((target("ssse3"))) inline uint32_t adler32_ssse3(const uint8_t* p, size_t n) {
    uint32_t a = 1, b = 0;
    const __m128i weights = _mm_setr_epi8(16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1);
    const __m128i ones16 = _mm_set1_epi16(1), zero = _mm_setzero_si128();
    while (n >= 16) {
        size_t blocks = n / 16 < 5552 / 16 ? n / 16 : 5552 / 16;      // 347 steps: b's lanes stay below 2^32
        n -= blocks * 16;
        __m128i va = zero, vb = zero, va_before = zero;     // va: 2 x 64-bit byte sums; vb, va_before: 4 x 32-bit
        const uint32_t a0 = a;
        b += a0 * (uint32_t)(blocks * 16) % 65521u;         // every byte position adds the block's starting a once
        for (size_t i = 0; i < blocks; ++i, p += 16) {
            const __m128i x = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p));
            va_before = _mm_add_epi32(va_before, va);       // sum over steps of (byte sum so far), low halves of the 64-bit lanes
            va = _mm_add_epi64(va, _mm_sad_epu8(x, zero));
            vb = _mm_add_epi32(vb, _mm_madd_epi16(_mm_maddubs_epi16(x, weights), ones16));
        }
        auto hsum32 = [](__m128i v) {
            v = _mm_add_epi32(v, _mm_shuffle_epi32(v, 0x4E));
            v = _mm_add_epi32(v, _mm_shuffle_epi32(v, 0xB1));
            return (uint32_t)_mm_cvtsi128_si32(v);
        };
        const uint32_t sum_a = (uint32_t)_mm_cvtsi128_si32(va) + (uint32_t)_mm_cvtsi128_si32(_mm_shuffle_epi32(va, 0x4E));
        // va_before's lanes 0 and 2 hold the running byte sums (lanes 1, 3 are the zero high halves of va)
        const uint64_t before = (uint64_t)hsum32(va_before);
        b = (uint32_t)((b + 16 * before + hsum32(vb)) % 65521u);
        a = (a0 + sum_a) % 65521u;
    }
    return n ? adler32_scalar(p, n, a, b) : ((b << 16) | a);
}
#endif

inline uint32_t adler32(const uint8_t* p, size_t n) {
#if defined(__x86_64__)
    static const bool have = __builtin_cpu_supports("ssse3");
    if (have) return adler32_ssse3(p, n);
#endif
    return adler32_scalar(p, n, 1, 0);
}

// ---- one stream's decoder state ---------------------------------------------------------------------------------
// The symbol loop keeps its state in a Cursor (a local object whose address does not escape: after inlining the fields
// live in registers); a Stream carries a Cursor from block to block.
struct Cursor {
    uint64_t buf;                                        // bit buffer, LSB first
    unsigned cnt;                                        // valid bits in buf
    const uint8_t* in;                                   // next byte to load
    const uint8_t* in_end;                               // end of the deflate data (the Adler-32 trailer follows)
    uint8_t* out;
    uint8_t* out0;
    uint8_t* out_end;
    const uint32_t* lt;
    const uint32_t* dt;

    void refill() {                                      // afterwards 56 <= cnt <= 63
        uint64_t w;
        memcpy(&w, in, 8);                               // little-endian host (x86-64)
        buf |= w << cnt;
        in += (63 - cnt) >> 3;
        cnt |= 56;
    }
    unsigned peek(int n) const { return (unsigned)(buf & ((1ull << n) - 1)); }
    void drop(unsigned n) { buf >>= (n & 63); cnt -= n; }    // & 63: what the shift instruction does anyway, so no mask is emitted
    unsigned take(int n) { const unsigned v = peek(n); drop((unsigned)n); return v; }
    const uint8_t* position() const { return in - (cnt >> 3); }   // first byte not consumed
    void restart(const uint8_t* p) { in = p; buf = 0; cnt = 0; }
    uint32_t lookup_lt() const {
        uint32_t e = lt[buf & ((1u << LT_BITS) - 1)];
        if (e & F_SUB) e = lt[(e >> 16) + ((buf >> LT_BITS) & ((1u << ((e >> 8) & 0x1F)) - 1))];
        return e;
    }
    uint32_t lookup_dt() const {
        uint32_t d = dt[buf & ((1u << DT_BITS) - 1)];
        if (d & F_SUB) d = dt[(d >> 16) + ((buf >> DT_BITS) & ((1u << ((d >> 8) & 0x1F)) - 1))];
        return d;
    }
};

enum { STEP_GO = 0, STEP_EOB = 1, STEP_BAD = 2 };

// One refill's worth of symbols: up to three literals, or literals and then a match, or the end of the block.
// Bits used after a refill (>= 56 available): 3 x 15 for three literals; otherwise the bit buffer is refilled again in
// front of the length/distance pair (15 + 5 + 15 + 13 = 48).
static inline __attribute__((always_inline)) int step(Cursor& c) {
    if (c.position() > c.in_end || c.out > c.out_end) return STEP_BAD;
    c.refill();
    uint32_t e = c.lookup_lt();
    if (e & F_LITERAL) {
        c.drop(e & 0xFF);
        *c.out++ = (uint8_t)(e >> 16);
        e = c.lookup_lt();
        if (e & F_LITERAL) {
            c.drop(e & 0xFF);
            *c.out++ = (uint8_t)(e >> 16);
            e = c.lookup_lt();
            if (e & F_LITERAL) {
                c.drop(e & 0xFF);
                *c.out++ = (uint8_t)(e >> 16);
                return STEP_GO;
            }
        }
        if (c.out > c.out_end) return STEP_BAD;
        c.refill();
    }
    if (e & F_SPECIAL) {
        if (e >> 16) return STEP_BAD;                     // invalid code
        c.drop(e & 0xFF);
        return STEP_EOB;
    }
    c.drop(e & 0xFF);
    const unsigned len = (e >> 16) + c.take((int)((e >> 8) & 0x1F));
    const uint32_t d = c.lookup_dt();
    if (d & F_SPECIAL) return STEP_BAD;
    c.drop(d & 0xFF);
    const size_t dist = (d >> 16) + c.take((int)((d >> 8) & 0x1F));
    if (dist > (size_t)(c.out - c.out0) || (size_t)(c.out_end - c.out) < len) return STEP_BAD;
    const uint8_t* src = c.out - dist;
    uint8_t* dst = c.out;
    c.out += len;
    if (dist >= 8) {
        do { uint64_t w; memcpy(&w, src, 8); memcpy(dst, &w, 8); src += 8; dst += 8; } while (dst < c.out);
    } else {
        do { *dst++ = *src++; } while (dst < c.out);
    }
    return STEP_GO;
}

struct Stream {
    enum State { HEADER, SYMBOLS, DONE, FAILED };
    Cursor c;
    Tables* tables;
    size_t out_n;
    bool final_block;
    State state;
};

// `in` readable for n + 16 bytes (zero padding), `out` writable for out_n + OUT_SLACK bytes.
inline void begin(Stream& s, const uint8_t* in, size_t n, uint8_t* out, size_t out_n, Tables& T) {
    s.tables = &T; s.out_n = out_n; s.final_block = false; s.state = Stream::FAILED;
    if (n < 6) return;
    const unsigned cmf = in[0], flg = in[1];
    if ((cmf & 0x0F) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) return;
    s.c.restart(in + 2);
    s.c.in_end = in + n - 4;                              // the trailer is not deflate data
    s.c.out = s.c.out0 = out;
    s.c.out_end = out + out_n;
    s.c.lt = T.lt; s.c.dt = T.dt;
    s.state = Stream::HEADER;
}

// From the end of a block to the next symbol loop (state SYMBOLS: tables built), through stored blocks on the way; or
// to the end of the stream (DONE: length, position and Adler-32 verified) or its failure.
inline void next_block(Stream& s) {
    Cursor& c = s.c;
    s.state = Stream::FAILED;
    for (;;) {
        if (s.final_block) {
            if (c.out != c.out_end) return;
            c.drop(c.cnt & 7);
            const uint8_t* p = c.position();
            if (p != c.in_end) return;                    // zlib: trailing garbage / truncated stream
            const uint32_t want = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
            if (adler32(c.out0, s.out_n) == want) s.state = Stream::DONE;
            return;
        }
        if (c.position() > c.in_end) return;
        c.refill();
        s.final_block = c.take(1) != 0;
        const unsigned type = c.take(2);
        if (type == 0) {
            c.drop(c.cnt & 7);
            const uint8_t* p = c.position();
            if (c.in_end - p < 4) return;
            const unsigned len = p[0] | (p[1] << 8), nlen = p[2] | (p[3] << 8);
            if ((len ^ nlen) != 0xFFFFu) return;
            p += 4;
            if ((size_t)(c.in_end - p) < len || (size_t)(c.out_end - c.out) < len) return;
            memcpy(c.out, p, len);
            c.out += len;
            c.restart(p + len);
            continue;
        }
        if (type == 3) return;
        uint8_t lens[320];
        int hlit, hdist;
        if (type == 1) {
            hlit = 288; hdist = 32;
            for (int i = 0; i < 144; ++i) lens[i] = 8;
            for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7;
            for (int i = 280; i < 288; ++i) lens[i] = 8;
            for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
        } else {
            hlit = (int)c.take(5) + 257; hdist = (int)c.take(5) + 1;
            const int hclen = (int)c.take(4) + 4;
            if (hlit > 286 || hdist > 30) return;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            c.refill();
            for (int i = 0; i < hclen; ++i) {
                if (c.cnt < 3) c.refill();
                cl[order[i]] = (uint8_t)c.take(3);
            }
            uint32_t pre[128 + 1];
            if (!build_table(cl, 19, pre, 7, 128, true, [](int sym) { return entry((unsigned)sym, 0, 0, 0); })) return;
            int i = 0;
            while (i < hlit + hdist) {
                if (c.position() > c.in_end) return;
                c.refill();
                const uint32_t e = pre[c.peek(7)];
                if (e & F_SPECIAL) return;
                c.drop(e & 0xFF);
                const unsigned sym = e >> 16;
                if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                unsigned rep, val = 0;
                if (sym == 16) {
                    if (i == 0) return;
                    val = lens[i - 1]; rep = 3 + c.take(2);
                } else if (sym == 17) rep = 3 + c.take(3);
                else rep = 11 + c.take(7);
                if (i + (int)rep > hlit + hdist) return;
                memset(lens + i, (int)val, rep);
                i += (int)rep;
            }
            if (lens[256] == 0) return;                   // no end-of-block code
        }
        if (!build_table(lens, hlit, s.tables->lt, LT_BITS, LT_SIZE, false, litlen_entry)) return;
        if (!build_table(lens + hlit, hdist, s.tables->dt, DT_BITS, DT_SIZE, false, dist_entry)) return;
        s.state = Stream::SYMBOLS;
        return;
    }
}

// The symbols of one block.
inline void run_symbols(Stream& s) {
    Cursor c = s.c;
    int r;
    do { r = step(c); } while (r == STEP_GO);
    s.c = c;
    s.state = r == STEP_EOB ? Stream::HEADER : Stream::FAILED;
}

// The symbols of two streams in lock-step, until either leaves its block.  A Huffman-coded stream is one dependent
// chain (table load -> shift -> next table load, ~6 cycles per symbol however wide the machine); two independent chains
// in one loop give the out-of-order core twice the work per cycle.
inline void run_symbols2(Stream& sa, Stream& sb) {
    Cursor a = sa.c, b = sb.c;
    int ra = STEP_GO, rb = STEP_GO;
    for (;;) {
        ra = step(a);
        if (ra != STEP_GO) break;
        rb = step(b);
        if (rb != STEP_GO) break;
    }
    sa.c = a; sb.c = b;
    if (ra != STEP_GO) sa.state = ra == STEP_EOB ? Stream::HEADER : Stream::FAILED;
    if (rb != STEP_GO) sb.state = rb == STEP_EOB ? Stream::HEADER : Stream::FAILED;
}

inline void finish_alone(Stream& s) {
    for (;;) {
        if (s.state == Stream::HEADER) next_block(s);
        if (s.state != Stream::SYMBOLS) return;
        run_symbols(s);
    }
}

// Decompress a zlib stream of n bytes; `in` must be readable for n + 16 bytes (zero padding), `out` writable for
// out_n + OUT_SLACK bytes.  Returns true iff the stream is well-formed, inflates to exactly out_n bytes and its Adler-32
// matches.
inline bool inflate_zlib(const uint8_t* in, size_t n, uint8_t* out, size_t out_n, Tables& T) {
    Stream s;
    begin(s, in, n, out, out_n, T);
    finish_alone(s);
    return s.state == Stream::DONE;
}

// Two streams at once (same contract each, separate tables): ok_a / ok_b as inflate_zlib would return them.
inline void inflate_zlib2(const uint8_t* in_a, size_t n_a, uint8_t* out_a, size_t out_n_a, Tables& Ta, bool& ok_a,
                          const uint8_t* in_b, size_t n_b, uint8_t* out_b, size_t out_n_b, Tables& Tb, bool& ok_b) {
    Stream a, b;
    begin(a, in_a, n_a, out_a, out_n_a, Ta);
    begin(b, in_b, n_b, out_b, out_n_b, Tb);
    for (;;) {
        if (a.state == Stream::HEADER) next_block(a);
        if (b.state == Stream::HEADER) next_block(b);
        if (a.state != Stream::SYMBOLS || b.state != Stream::SYMBOLS) break;
        run_symbols2(a, b);
    }
    finish_alone(a);
    finish_alone(b);
    ok_a = a.state == Stream::DONE;
    ok_b = b.state == Stream::DONE;
}

}  // namespace bqinf
