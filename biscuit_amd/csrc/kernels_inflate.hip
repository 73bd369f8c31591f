// zlib-stream (RFC 1950 / 1951) decompression on the device: the PNG tiles of a Slideflow TFRecord (configure.py:118-124:
// img_format='png') arrive as ~155-226 KB of DEFLATE each, and inflating them is what the host cores spend their time on (1.0-1.3 k
// tiles/s per core: eight ranks x 33 k tiles/s would want ~240 cores).  Here the host only walks the record framing and copies the
// IDAT payloads (bqio_extract_z); the GPU inflates, reverses the scanline filters (kernels_png.hip) and standardises.
//
// Huffman decoding is one dependent chain per stream -- table load -> shift -> next table load -- so the parallelism is ACROSS
// streams: one lane per stream, 64 streams per wave, thousands of streams in flight.
//  * bit buffer: 64 bits per lane, refilled with one ALIGNED dword whenever 32 bits or fewer are left (streams start on 16-byte
//    boundaries in the packed input, so a lane's loads are aligned dwords all the way);
//  * decode tables: two levels, 9 bits direct for literal/length codes and 6 for distance codes, sub-tables behind them for the
//    longer codes; an entry carries the code length, the extra-bit count and the base value (the layout of the host decoder,
//    csrc/inflate_fast.h).  A stream's tables live in its 8 KB slice of a caller-provided scratch buffer; zlib's deflate closes a
//    block every 16 383 symbols whatever the data, so the lanes of a wave reach their block headers -- and rebuild their tables --
//    together;
//  * the symbol loop runs in ROUNDS (the round kernel below): a literal-only fast phase out of a small per-lane table in LDS and a
//    general phase through the global tables that only the stalled lanes enter -- a trip to L2 costs ~1 200 cycles, and with 64
//    streams in lock step the plain loop pays it per symbol (profiles/r05_inflate.txt);
//  * the next input dword is loaded one refill ahead of its use;
//  * output: literals are collected into a dword per lane and stored when it is full; a match copies from the lane's own output.
// Results are zlib's: a stream is accepted iff it is well-formed and inflates to exactly the expected number of bytes; the Adler-32
// trailer is checked by a second small kernel (one wave per stream) so that "accepted" means what uncompress() means.
#include "bq_common.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int LT_BITS = 9, DT_BITS = 6;
constexpr int LT_CAP = 1024;             // 512 direct + sub-tables (zlib's bound for 9 root bits: 852)
constexpr int DT_CAP = 640;              // 64 direct + sub-tables (592)
constexpr unsigned F_LITERAL = 0x2000u, F_SUB = 0x4000u, F_SPECIAL = 0x8000u;   // SPECIAL: value 0 = end of block, 1 = invalid
// per-stream scratch, in 32-bit words: lt | dt | lens (320 bytes) | codes (320 x 16 bit) | sub_bits (512 bytes)
constexpr int SCR_LT = 0, SCR_DT = LT_CAP, SCR_LENS = SCR_DT + DT_CAP, SCR_CODES = SCR_LENS + 80, SCR_SUB = SCR_CODES + 160,
              SCR_WORDS = SCR_SUB + 128;

enum { INF_OK = 0, INF_BAD_HEADER = 1, INF_BAD_BLOCK = 2, INF_BAD_TABLE = 3, INF_BAD_CODE = 4, INF_BAD_DISTANCE = 5, INF_OVERRUN = 6,
       INF_LENGTH = 7, INF_TRAILING = 8, INF_ADLER = 9, INF_FILTER = 10 };

__device__ __forceinline__ unsigned entry(unsigned value, unsigned flags, unsigned extra, unsigned nbits) {
    return (value << 16) | flags | (extra << 8) | nbits;
}

__device__ const unsigned short kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const unsigned char kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const unsigned short kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
                                                 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const unsigned char kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const unsigned char kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// kind: 0 = code-length code (the entry is the symbol), 1 = literal/length, 2 = distance
__device__ __forceinline__ unsigned sym_entry(int kind, int sym) {
    if (kind == 0) return entry((unsigned)sym, 0, 0, 0);
    if (kind == 1) {
        if (sym < 256) return entry((unsigned)sym, F_LITERAL, 0, 0);
        if (sym == 256) return entry(0, F_SPECIAL, 0, 0);
        if (sym > 285) return entry(1, F_SPECIAL, 0, 0);
        return entry(kLenBase[sym - 257], 0, kLenExtra[sym - 257], 0);
    }
    if (sym > 29) return entry(1, F_SPECIAL, 0, 0);
    return entry(kDistBase[sym], 0, kDistExtra[sym], 0);
}

// Canonical Huffman decode table from code lengths (RFC 1951 3.2.2), two levels.  False for an over-subscribed set of lengths, for
// one that does not fit the table, and -- as zlib -- for an incomplete set unless it is empty or a single code of length 1 in a
// literal/length or distance table (unused codes decode as invalid).  The arithmetic of csrc/inflate_fast.h: build_table.
__device__ __noinline__ bool build_table(const unsigned char* lens, int nsym, unsigned* table, int table_bits, int table_cap, int kind,
                                         unsigned short* codes, unsigned char* sub_bits) {
    int count[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (int s = 0; s < nsym; ++s) ++count[lens[s] & 15];
    count[0] = 0;
    unsigned next_code[16];
    {
        unsigned code = 0;
        int space = 1;
        bool over = false;
#pragma unroll
        for (int l = 1; l <= 15; ++l) {
            space = space * 2 - count[l];
            over |= space < 0;
            code = (code + (unsigned)count[l - 1]) << 1;
            next_code[l] = code;
        }
        if (over) return false;
        if (space > 0) {
            int total = 0, longest = 0;
#pragma unroll
            for (int l = 1; l <= 15; ++l) if (count[l]) { total += count[l]; longest = l; }
            if (total != 0 && (kind == 0 || longest != 1)) return false;
        }
    }
    const unsigned invalid = entry(1, F_SPECIAL, 0, 1);
    const int primary = 1 << table_bits;
    for (int i = 0; i < primary; ++i) table[i] = invalid;
    bool any_long = false;
    for (int l = table_bits + 1; l <= 15; ++l) any_long |= count[l] != 0;
    if (any_long)
        for (int i = 0; i < primary; ++i) sub_bits[i] = 0;
    for (int s = 0; s < nsym; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const unsigned rev = __brev(next_code[l]++) >> (32 - l);
        codes[s] = (unsigned short)rev;
        if (l > table_bits) {
            const unsigned prefix = rev & (unsigned)(primary - 1);
            if (l - table_bits > sub_bits[prefix]) sub_bits[prefix] = (unsigned char)(l - table_bits);
        }
    }
    int used = primary;
    if (any_long) {
        for (int pfx = 0; pfx < primary; ++pfx) {
            const int sb = sub_bits[pfx];
            if (!sb) continue;
            const int size = 1 << sb;
            if (used + size > table_cap) return false;
            table[pfx] = entry((unsigned)used, F_SUB, (unsigned)sb, (unsigned)table_bits);
            for (int i = 0; i < size; ++i) table[used + i] = invalid;
            used += size;
        }
    }
    for (int s = 0; s < nsym; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const unsigned rev = codes[s];
        const unsigned e = sym_entry(kind, s) | (unsigned)l;
        if (l <= table_bits) {
            for (unsigned i = rev; i < (unsigned)primary; i += 1u << l) table[i] = e;
        } else {
            const unsigned pe = table[rev & (unsigned)(primary - 1)];
            const unsigned base = pe >> 16, sb = (pe >> 8) & 0x1F;
            for (unsigned i = rev >> table_bits; i < (1u << sb); i += 1u << (l - table_bits)) table[base + i] = e;
        }
    }
    return true;
}

struct InflateParams {
    const unsigned char* z;        // packed zlib streams; stream i = z[off[i] .. off[i] + len[i]), off[i] a multiple of 16, at least
                                   // 32 readable bytes behind every stream
    const unsigned* off;           // [n]
    const unsigned* len;           // [n]
    unsigned char* out;            // [n][out_stride]
    unsigned out_len, out_stride;  // bytes every stream must inflate to; distance between outputs (>= out_len + 4, multiple of 4)
    unsigned* scratch;             // [n][SCR_WORDS]
    int* status;                   // [n]: INF_*
    int n;
    unsigned row_len;              // PNG scanline length (1 + 3 px) whose filter-type bytes are checked, or 0
};

// the bit reader of one lane: 64 bits, refilled a dword at a time; the dword after next is already on its way when one is used
struct Bits {
    const unsigned* p;             // the dword after `ahead`
    unsigned ahead;                // the next dword (loaded one refill early)
    unsigned long long buf;
    int cnt;
    __device__ __forceinline__ void open(const unsigned* q) { buf = 0; cnt = 0; ahead = q[0]; p = q + 1; refill(); }
    __device__ __forceinline__ void refill() {      // call with cnt <= 32: afterwards 33 <= cnt <= 64
        buf |= (unsigned long long)ahead << cnt;
        cnt += 32;
        ahead = *p++;
    }
    __device__ __forceinline__ void need() { if (cnt <= 32) refill(); }
    __device__ __forceinline__ unsigned peek(int n) const { return (unsigned)buf & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(int n) { buf >>= n; cnt -= n; }
    __device__ __forceinline__ unsigned take(int n) { const unsigned v = peek(n); drop(n); return v; }
    // the first byte not consumed yet, given the stream's first byte
    __device__ __forceinline__ long long consumed(const unsigned char* z0) const {
        return (long long)(reinterpret_cast<const unsigned char*>(p) - z0) - 4 - (cnt >> 3);
    }
};

constexpr int INF_NT = 64;                                    // one wave per workgroup

// The kernel WITHOUT LDS (option inflate_variant = 0): every table access goes to the stream's scratch slice in global memory (L2) and
// as many waves per CU as the registers allow -- throughput comes from occupancy, not latency.  It is the fallback for a device
// whose LDS the round kernel below cannot have (hipFuncSetAttribute refused), and the simplest statement of the algorithm.
__global__ void __launch_bounds__(INF_NT) inflate_kernel(const InflateParams p) {
    const int i = blockIdx.x * INF_NT + threadIdx.x;
    if (i >= p.n) return;
    unsigned* const scr = p.scratch + (size_t)i * SCR_WORDS;
    unsigned* const lt = scr + SCR_LT;
    unsigned* const dt = scr + SCR_DT;
    unsigned char* const lens = reinterpret_cast<unsigned char*>(scr + SCR_LENS);
    unsigned short* const codes = reinterpret_cast<unsigned short*>(scr + SCR_CODES);
    unsigned char* const sub_bits = reinterpret_cast<unsigned char*>(scr + SCR_SUB);
    const unsigned char* const zin = p.z + p.off[i];
    const unsigned zlen = p.len[i];
    unsigned char* const out0 = p.out + (size_t)i * p.out_stride;
    const unsigned out_len = p.out_len;
    int status = INF_OK;

    Bits b;
    b.open(reinterpret_cast<const unsigned*>(zin));
    // the dwords a well-formed stream may touch: its own bytes (header, data, 4 bytes of trailer) and up to 12 bytes of look-ahead
    const unsigned* const p_limit = reinterpret_cast<const unsigned*>(zin) + (zlen + 3) / 4 + 4;
    {
        const unsigned cmf = b.take(8), flg = b.take(8);
        if (zlen < 6 || (cmf & 0x0F) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) status = INF_BAD_HEADER;
    }
    unsigned op = 0;                // bytes produced
    unsigned acc = 0;               // the bytes of the output dword in progress (positions op & ~3 .. op - 1)
    bool final_block = false;
    while (status == INF_OK && !final_block) {
        if (b.p > p_limit) { status = INF_OVERRUN; break; }
        b.need();
        final_block = b.take(1) != 0;
        const unsigned type = b.take(2);
        if (type == 0) {
            // stored: the rest of the current byte is skipped, LEN / NLEN, then LEN bytes
            b.drop(b.cnt & 7);
            b.need();
            const unsigned len = b.take(16);
            b.need();
            const unsigned nlen = b.take(16);
            if ((len ^ nlen) != 0xFFFFu) { status = INF_BAD_BLOCK; break; }
            if (len > out_len - op) { status = INF_LENGTH; break; }
            for (unsigned k = 0; k < len; ++k) {
                b.need();
                const unsigned v = b.take(8);
                acc |= v << (8 * (op & 3));
                ++op;
                if ((op & 3) == 0) { *reinterpret_cast<unsigned*>(out0 + op - 4) = acc; acc = 0; }
                if (b.p > p_limit) { status = INF_OVERRUN; break; }
            }
            continue;
        }
        if (type == 3) { status = INF_BAD_BLOCK; break; }
        int hlit, hdist;
        if (type == 1) {
            hlit = 288; hdist = 32;
            for (int k = 0; k < 144; ++k) lens[k] = 8;
            for (int k = 144; k < 256; ++k) lens[k] = 9;
            for (int k = 256; k < 280; ++k) lens[k] = 7;
            for (int k = 280; k < 288; ++k) lens[k] = 8;
            for (int k = 0; k < 32; ++k) lens[288 + k] = 5;
        } else {
            b.need();
            hlit = (int)b.take(5) + 257; hdist = (int)b.take(5) + 1;
            const int hclen = (int)b.take(4) + 4;
            if (hlit > 286 || hdist > 30) { status = INF_BAD_BLOCK; break; }
            // the code-length code's 7-bit table borrows the distance table's words (nothing lives there between blocks), its 19
            // lengths the upper half of sub_bits (which a 7-bit table without long codes does not touch)
            unsigned char* const cl_mem = sub_bits + 256;
            for (int k = 0; k < 19; ++k) cl_mem[k] = 0;
            for (int k = 0; k < hclen; ++k) {
                b.need();
                cl_mem[kClOrder[k]] = (unsigned char)b.take(3);
            }
            if (!build_table(cl_mem, 19, dt, 7, 128, 0, codes, sub_bits)) { status = INF_BAD_TABLE; break; }
            int k = 0;
            const int total = hlit + hdist;
            bool bad = false;
            while (k < total) {
                if (b.p > p_limit) { bad = true; break; }
                b.need();
                const unsigned e = dt[b.peek(7)];
                if (e & F_SPECIAL) { bad = true; break; }
                b.drop((int)(e & 0xFF));
                const unsigned sym = e >> 16;
                if (sym < 16) { lens[k++] = (unsigned char)sym; continue; }
                unsigned rep, val = 0;
                if (sym == 16) {
                    if (k == 0) { bad = true; break; }
                    val = lens[k - 1]; rep = 3 + b.take(2);
                } else if (sym == 17) rep = 3 + b.take(3);
                else rep = 11 + b.take(7);
                if (k + (int)rep > total) { bad = true; break; }
                for (unsigned r = 0; r < rep; ++r) lens[k + r] = (unsigned char)val;
                k += (int)rep;
            }
            if (bad || lens[256] == 0) { status = INF_BAD_TABLE; break; }
        }
        if (!build_table(lens, hlit, lt, LT_BITS, LT_CAP, 1, codes, sub_bits) ||
            !build_table(lens + hlit, hdist, dt, DT_BITS, DT_CAP, 2, codes, sub_bits)) { status = INF_BAD_TABLE; break; }
        // ---- the symbols of the block
        for (;;) {
            if (b.p > p_limit) { status = INF_OVERRUN; break; }
            b.need();
            // literal / length symbol: the 16-bit direct entry from LDS, a longer code through its link into the global sub-table
            // -- or both levels from the global table
            unsigned kind, nb, val, lbase = 0, lextra = 0;
            unsigned e = lt[(unsigned)b.buf & ((1u << LT_BITS) - 1u)];
            if (e & F_SUB) e = lt[(e >> 16) + (((unsigned)(b.buf >> LT_BITS)) & ((1u << ((e >> 8) & 0x1F)) - 1u))];
            nb = e & 0xFFu;
            val = e >> 16;
            if (e & F_LITERAL) kind = 0u;
            else if (e & F_SPECIAL) kind = 3u;
            else { kind = 1u; lbase = val; lextra = (e >> 8) & 0x1Fu; }
        
            if (kind == 0u) {
                b.drop((int)nb);
                if (op >= out_len) { status = INF_LENGTH; break; }
                acc |= val << (8 * (op & 3));
                ++op;
                if ((op & 3) == 0) { *reinterpret_cast<unsigned*>(out0 + op - 4) = acc; acc = 0; }
                continue;
            }
            if (kind == 3u) {
                if (val & 1u) { status = INF_BAD_CODE; break; }
                b.drop((int)nb);
                break;                                      // end of block
            }
            b.drop((int)nb);
            const unsigned len = lbase + b.take((int)lextra);
            b.need();
            unsigned dnb, dbase, dextra;
            unsigned d = dt[(unsigned)b.buf & ((1u << DT_BITS) - 1u)];
            if (d & F_SUB) d = dt[(d >> 16) + (((unsigned)(b.buf >> DT_BITS)) & ((1u << ((d >> 8) & 0x1F)) - 1u))];
            if (d & F_SPECIAL) { status = INF_BAD_CODE; break; }
            dnb = d & 0xFFu; dbase = d >> 16; dextra = (d >> 8) & 0x1Fu;
        
            b.drop((int)dnb);
            const unsigned dist = dbase + b.take((int)dextra);
            if (dist > op) { status = INF_BAD_DISTANCE; break; }
            if (len > out_len - op) { status = INF_LENGTH; break; }
            // the copy reads this lane's own output: the bytes of the dword in progress go out first
            if (op & 3) *reinterpret_cast<unsigned*>(out0 + (op & ~3u)) = acc;
            for (unsigned k = 0; k < len; ++k) {
                const unsigned v = out0[op - dist];
                out0[op] = (unsigned char)v;
                ++op;
            }
            // re-read the dword in progress (the copy may have written part of it)
            acc = (op & 3) ? (*reinterpret_cast<const unsigned*>(out0 + (op & ~3u)) & ((1u << (8 * (op & 3))) - 1u)) : 0u;
        }
    }
    if (status == INF_OK) {
        if (op & 3) *reinterpret_cast<unsigned*>(out0 + (op & ~3u)) = acc;
        if (op != out_len) status = INF_LENGTH;
        else {
            // the deflate data must end where the Adler-32 trailer begins: zlen - 4 bytes in (zlib refuses trailing garbage)
            b.drop(b.cnt & 7);
            if (b.consumed(zin) != (long long)zlen - 4) status = INF_TRAILING;
        }
    }
    p.status[i] = status;
}

// ---- the round kernel (option inflate_variant = 5, the default) -------------------------------------------------------------------------
// Lanes of a wave share one program counter: in the kernel above a symbol that needs the slow path in ANY of the 64 streams -- a code
// longer than the direct table, a match, a refill from memory -- costs all 64 a trip to L2 (~1 000 cycles), and with 64 streams that
// is every symbol.  Here the two kinds of work are separated in time:
//   fast phase   up to FAST symbols per lane with NO memory access but LDS: a per-lane table of 2^L0 16-bit entries answers "is the
//                next code a literal of at most L0 bits" (93 % of the symbols of a nearly incompressible tile, 99 % of a photo-like
//                one); the literal goes to a 32-byte ring in LDS with one ds_write_b8, the bits come out of the 64-bit buffer; the
//                fast step is a table read, a compare and a shift.  A lane whose next symbol is anything else idles until the phase ends;
//   slow phase   once per round: the ring's complete dwords are stored, the bit buffer is refilled from memory, and ONLY the lanes
//                that stopped on a non-literal decode one symbol the general way (two-level tables in the scratch buffer, matches,
//                end of block, block headers and table construction).
// (Rounds 5's other forms -- direct tables mirrored in LDS, a 128-bit pending register, other table widths and phase lengths -- were
// measured slower and are gone: profiles/r05_inflate.txt has their figures.)
template <int L0, int FAST>
__global__ void __launch_bounds__(INF_NT) inflate_rounds2_kernel(const InflateParams p) {
    constexpr int RING = 32;                                   // bytes of output ring per lane
    constexpr int PER_LANE = (1 << L0) * 2 + RING;             // bytes of LDS per lane
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int i = blockIdx.x * INF_NT + threadIdx.x;
    const bool mine = i < p.n;
    const int ii = mine ? i : 0;
    unsigned short* const l0 = reinterpret_cast<unsigned short*>(lds_raw + threadIdx.x * PER_LANE);
    unsigned char* const ring = lds_raw + threadIdx.x * PER_LANE + (1 << L0) * 2;
    unsigned* const scr = p.scratch + (size_t)ii * SCR_WORDS;
    unsigned* const lt = scr + SCR_LT;
    unsigned* const dt = scr + SCR_DT;
    unsigned char* const lens = reinterpret_cast<unsigned char*>(scr + SCR_LENS);
    unsigned short* const codes = reinterpret_cast<unsigned short*>(scr + SCR_CODES);
    unsigned char* const sub_bits = reinterpret_cast<unsigned char*>(scr + SCR_SUB);
    const unsigned char* const zin = p.z + p.off[ii];
    const unsigned zlen = p.len[ii];
    unsigned char* const out0 = p.out + (size_t)ii * p.out_stride;
    const unsigned out_len = p.out_len;
    int status = INF_OK;

    Bits b;
    b.open(reinterpret_cast<const unsigned*>(zin));
    const unsigned* const p_limit = reinterpret_cast<const unsigned*>(zin) + (zlen + 3) / 4 + 4;
    {
        const unsigned cmf = b.take(8), flg = b.take(8);
        if (zlen < 6 || (cmf & 0x0F) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) status = INF_BAD_HEADER;
    }
    unsigned op = 0;                            // bytes produced; byte k of the output waits at ring[k & (RING - 1)] until it is stored
    unsigned flushed = 0;                       // a multiple of 4: bytes below it are in memory
    bool final_block = false;
    enum { ST_HEADER = 0, ST_SYMBOLS = 1, ST_DONE = 2 };
    int state = mine ? ST_HEADER : ST_DONE;
    if (status != INF_OK) state = ST_DONE;
    // complete dwords of the ring -> memory (partial: the dword in progress too, as a dword -- the bytes behind op are overwritten
    // later, the output stride leaves room behind the last one)
    auto flush = [&](bool partial) {
        const unsigned end = partial ? (op + 3u) & ~3u : op & ~3u;
        for (unsigned k = flushed; k < end; k += 4)
            *reinterpret_cast<unsigned*>(out0 + k) = *reinterpret_cast<const unsigned*>(ring + (k & (RING - 1)));
        flushed = op & ~3u;
    };

    while (__builtin_amdgcn_ballot_w64(state != ST_DONE) != 0ull) {
        bool stalled = false;                   // stopped on something that is not a short literal
        // ---- fast phase: no memory access but the table
        if (state == ST_SYMBOLS) {
            bool go = out_len - op >= (unsigned)FAST;      // (the last few bytes of a stream take the general path)
#pragma unroll
            for (int f = 0; f < FAST; ++f) {
                const unsigned e = l0[(unsigned)b.buf & ((1u << L0) - 1u)];
                const int nb = (int)(e >> 8);              // 0 for "not a short literal"
                const bool hit = nb != 0;
                stalled = stalled || (go && !hit);
                go = go && hit && b.cnt >= nb;
                if (go) {
                    ring[op & (RING - 1)] = (unsigned char)e;
                    ++op;
                    b.drop(nb);
                }
            }
        }
        // ---- slow phase
        if (state != ST_DONE) {
            flush(false);
            if (b.p > p_limit) { status = INF_OVERRUN; state = ST_DONE; }
            else b.need();
        }
        if (state == ST_SYMBOLS && (stalled || out_len - op < (unsigned)FAST)) {
            unsigned e = lt[(unsigned)b.buf & ((1u << LT_BITS) - 1u)];
            if (e & F_SUB) e = lt[(e >> 16) + (((unsigned)(b.buf >> LT_BITS)) & ((1u << ((e >> 8) & 0x1F)) - 1u))];
            if (e & F_LITERAL) {
                if (op >= out_len) { status = INF_LENGTH; state = ST_DONE; }
                else { b.drop((int)(e & 0xFF)); ring[op & (RING - 1)] = (unsigned char)(e >> 16); ++op; }
            } else if (e & F_SPECIAL) {
                if (e >> 16) { status = INF_BAD_CODE; state = ST_DONE; }
                else { b.drop((int)(e & 0xFF)); state = final_block ? ST_DONE : ST_HEADER; }
            } else {
                b.drop((int)(e & 0xFF));
                const unsigned len = (e >> 16) + b.take((int)((e >> 8) & 0x1F));
                b.need();
                unsigned d = dt[(unsigned)b.buf & ((1u << DT_BITS) - 1u)];
                if (d & F_SUB) d = dt[(d >> 16) + (((unsigned)(b.buf >> DT_BITS)) & ((1u << ((d >> 8) & 0x1F)) - 1u))];
                if (d & F_SPECIAL) { status = INF_BAD_CODE; state = ST_DONE; }
                else {
                    b.drop((int)(d & 0xFF));
                    const unsigned dist = (d >> 16) + b.take((int)((d >> 8) & 0x1F));
                    if (dist > op) { status = INF_BAD_DISTANCE; state = ST_DONE; }
                    else if (len > out_len - op) { status = INF_LENGTH; state = ST_DONE; }
                    else {
                        flush(true);                        // the copy reads this lane's own output
                        unsigned k = 0;
                        if (dist >= 4) {
                            for (; k + 4 <= len; k += 4) {  // four loads, then four stores
                                const unsigned b0 = out0[op - dist], b1 = out0[op - dist + 1], b2 = out0[op - dist + 2], b3 = out0[op - dist + 3];
                                out0[op] = (unsigned char)b0; out0[op + 1] = (unsigned char)b1; out0[op + 2] = (unsigned char)b2; out0[op + 3] = (unsigned char)b3;
                                ring[op & (RING - 1)] = (unsigned char)b0; ring[(op + 1) & (RING - 1)] = (unsigned char)b1;
                                ring[(op + 2) & (RING - 1)] = (unsigned char)b2; ring[(op + 3) & (RING - 1)] = (unsigned char)b3;
                                op += 4;
                            }
                        }
                        for (; k < len; ++k) {
                            const unsigned v = out0[op - dist];
                            out0[op] = (unsigned char)v;
                            ring[op & (RING - 1)] = (unsigned char)v;      // (the ring stays the truth for the dword in progress)
                            ++op;
                        }
                        flushed = op & ~3u;                 // everything below is in memory through the copy's own stores
                    }
                }
            }
        } else if (state == ST_HEADER) {
            final_block = b.take(1) != 0;
            const unsigned type = b.take(2);
            if (type == 0) {
                b.drop(b.cnt & 7);
                b.need();
                const unsigned len = b.take(16);
                b.need();
                const unsigned nlen = b.take(16);
                if ((len ^ nlen) != 0xFFFFu) { status = INF_BAD_BLOCK; state = ST_DONE; }
                else if (len > out_len - op) { status = INF_LENGTH; state = ST_DONE; }
                else {
                    for (unsigned k = 0; k < len && status == INF_OK; ++k) {
                        b.need();
                        ring[op & (RING - 1)] = (unsigned char)b.take(8);
                        ++op;
                        if (op - flushed >= (unsigned)RING - 4) flush(false);
                        if (b.p > p_limit) status = INF_OVERRUN;
                    }
                    if (status != INF_OK) state = ST_DONE;
                    else if (final_block) state = ST_DONE;
                }
            } else if (type == 3) { status = INF_BAD_BLOCK; state = ST_DONE; }
            else {
                int hlit = 288, hdist = 32;
                bool bad = false;
                if (type == 1) {
                    for (int k = 0; k < 144; ++k) lens[k] = 8;
                    for (int k = 144; k < 256; ++k) lens[k] = 9;
                    for (int k = 256; k < 280; ++k) lens[k] = 7;
                    for (int k = 280; k < 288; ++k) lens[k] = 8;
                    for (int k = 0; k < 32; ++k) lens[288 + k] = 5;
                } else {
                    b.need();
                    hlit = (int)b.take(5) + 257; hdist = (int)b.take(5) + 1;
                    const int hclen = (int)b.take(4) + 4;
                    if (hlit > 286 || hdist > 30) bad = true;
                    else {
                        unsigned char* const cl_mem = sub_bits + 256;
                        for (int k = 0; k < 19; ++k) cl_mem[k] = 0;
                        for (int k = 0; k < hclen; ++k) {
                            b.need();
                            cl_mem[kClOrder[k]] = (unsigned char)b.take(3);
                        }
                        if (!build_table(cl_mem, 19, dt, 7, 128, 0, codes, sub_bits)) bad = true;
                        int k = 0;
                        const int total = hlit + hdist;
                        while (!bad && k < total) {
                            if (b.p > p_limit) { bad = true; break; }
                            b.need();
                            const unsigned e = dt[b.peek(7)];
                            if (e & F_SPECIAL) { bad = true; break; }
                            b.drop((int)(e & 0xFF));
                            const unsigned sym = e >> 16;
                            if (sym < 16) { lens[k++] = (unsigned char)sym; continue; }
                            unsigned rep, val = 0;
                            if (sym == 16) {
                                if (k == 0) { bad = true; break; }
                                val = lens[k - 1]; rep = 3 + b.take(2);
                            } else if (sym == 17) rep = 3 + b.take(3);
                            else rep = 11 + b.take(7);
                            if (k + (int)rep > total) { bad = true; break; }
                            for (unsigned r = 0; r < rep; ++r) lens[k + r] = (unsigned char)val;
                            k += (int)rep;
                        }
                        if (!bad && lens[256] == 0) bad = true;
                    }
                }
                if (!bad && (!build_table(lens, hlit, lt, LT_BITS, LT_CAP, 1, codes, sub_bits) ||
                             !build_table(lens + hlit, hdist, dt, DT_BITS, DT_CAP, 2, codes, sub_bits))) bad = true;
                if (bad) { status = status == INF_OK ? INF_BAD_TABLE : status; state = ST_DONE; }
                else {
                    // the fast table: (code length << 8) | value for the literals whose codes fit L0 bits, 0 for everything else
                    for (int k = 0; k < (1 << L0); ++k) {
                        const unsigned e = lt[k];
                        const unsigned nb = e & 0xFFu;
                        l0[k] = ((e & F_LITERAL) && !(e & F_SUB) && nb <= (unsigned)L0) ? (unsigned short)((nb << 8) | ((e >> 16) & 0xFFu)) : (unsigned short)0;
                    }
                    state = ST_SYMBOLS;
                }
            }
        }
    }
    if (mine) {
        if (status == INF_OK) {
            flush(true);
            if (op != out_len) status = INF_LENGTH;
            else {
                b.drop(b.cnt & 7);
                if (b.consumed(zin) != (long long)zlen - 4) status = INF_TRAILING;
            }
        }
        p.status[i] = status;
    }
}

// Adler-32 of every stream's output against its trailer: one wave per stream.  a = 1 + sum(x) ; b = n + sum((n - k) x_k), mod 65521.
__global__ void __launch_bounds__(64) adler_kernel(const InflateParams p) {
    const int i = blockIdx.x;
    if (i >= p.n) return;
    const int lane = threadIdx.x;
    if (p.status[i] != INF_OK) return;
    const unsigned char* out = p.out + (size_t)i * p.out_stride;
    const unsigned n = p.out_len;
    // lane handles bytes k = 4 (lane + 64 j) .. + 3: sums of x and of (n - k) x in 64 bits (n < 2^24, x < 2^8, 2^16 terms: < 2^48)
    unsigned long long sa = 0, sb = 0;
    for (unsigned k = 4u * lane; k < n; k += 256u) {
        const unsigned w = *reinterpret_cast<const unsigned*>(out + k);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (k + q < n) {
                const unsigned x = (w >> (8 * q)) & 0xFF;
                sa += x;
                sb += (unsigned long long)(n - (k + q)) * x;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sa += __shfl_down(sa, o, 64);
        sb += __shfl_down(sb, o, 64);
    }
    if (lane == 0) {
        const unsigned a = (unsigned)((1ull + sa) % 65521ull);
        const unsigned bb = (unsigned)(((unsigned long long)n + sb) % 65521ull);
        const unsigned char* t = p.z + p.off[i] + p.len[i] - 4;
        const unsigned want = ((unsigned)t[0] << 24) | ((unsigned)t[1] << 16) | ((unsigned)t[2] << 8) | t[3];
        if (((bb << 16) | a) != want) p.status[i] = INF_ADLER;
    }
    // PNG scanlines (row_len = 1 + 3 px): a filter-type byte above 4 is a damaged file for every PNG decoder -- the host reader
    // refuses it (tfrecord_reader.cpp: png_finish), the un-filter kernel would read it as "None"
    if (p.row_len) {
        bool bad = false;
        for (unsigned r = lane; r * p.row_len < n; r += 64) bad |= out[(size_t)r * p.row_len] > 4;
        if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0 && p.status[i] == INF_OK) p.status[i] = INF_FILTER;
    }
}

}  // namespace

size_t inflate_scratch_bytes(int n) { return (size_t)(n > 0 ? n : 0) * SCR_WORDS * 4; }

// n zlib streams -> n x out_len bytes.  d_z / d_off / d_len: the packed input (see InflateParams); d_out: [n][out_stride] with
// out_stride >= out_len + 4 and a multiple of 4; d_scratch: inflate_scratch_bytes(n); d_status: [n] (0 = inflated and verified).
int launch_inflate(const unsigned char* d_z, const unsigned* d_off, const unsigned* d_len, int n, unsigned char* d_out, unsigned out_len,
                   unsigned out_stride, void* d_scratch, int* d_status, hipStream_t s, int variant, unsigned row_len) {
    if (n <= 0) return 0;
    if (out_stride < out_len + 4 || (out_stride & 3)) return (int)hipErrorInvalidValue;
    InflateParams p;
    p.z = d_z; p.off = d_off; p.len = d_len; p.out = d_out; p.out_len = out_len; p.out_stride = out_stride; p.row_len = row_len;
    p.scratch = reinterpret_cast<unsigned*>(d_scratch); p.status = d_status; p.n = n;
    if (variant != 0) {
        // option inflate_variant = 5 (default): rounds of a literal-only fast phase (7-bit table + a 32-byte output ring per lane in
        // LDS, 8 fast steps) and a general phase only stalled lanes enter
        constexpr size_t lds = (size_t)INF_NT * ((1u << 7) * 2 + 32);
        static BqLdsAttr at;
        if (const int e = at.ensure(reinterpret_cast<const void*>(inflate_rounds2_kernel<7, 8>), lds)) return e;
        hipLaunchKernelGGL((inflate_rounds2_kernel<7, 8>), dim3((n + INF_NT - 1) / INF_NT), dim3(INF_NT), lds, s, p);
    } else {
        hipLaunchKernelGGL(inflate_kernel, dim3((n + INF_NT - 1) / INF_NT), dim3(INF_NT), 0, s, p);
    }
    hipLaunchKernelGGL(adler_kernel, dim3(n), dim3(64), 0, s, p);
    return (int)hipGetLastError();
}
