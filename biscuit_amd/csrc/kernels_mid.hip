// Persistent SeparableConv2D kernel for the middle flow (728 -> 728 @ 19x19, 25 of the 34
// separable convolutions, ~60 % of the network's FLOPs), bf16.
//
// Compared with kernels_pipe.hip (same math, bit-identical results) it removes what that
// kernel was measured to be bound by:
//   * LDS bandwidth of the depthwise stage (3 ds_read_b128 per tap and 16-byte piece: data plus
//     two fp32 tap vectors).  Here a wave owns ONE 8-channel piece, so its 72 taps are
//     wave-uniform and live in scalar registers (s_load through the constant address space),
//     and a lane owns two vertically adjacent output pixels, so the 3x3 windows share rows:
//     12 ds_read_b128 per lane and chunk instead of ~40, 96 unpack + 144 v_fma instead of ~400
//     vector instructions.  No per-tap padding masks: the halo lives in LDS as a zero-padded
//     2-D image (pad columns and out-of-image rows are stored as zeros).
//   * lock-step HBM phases (every CU loading, then every CU computing, then every CU storing).
//     Workgroups are persistent: each walks several tiles, the store drain of tile t and the halo
//     loads of tile t+1 run under compute, and the B (weight) register ring keeps streaming
//     across tile boundaries.
//   * the short weight prefetch distance: the ring is one full chunk (4 k-blocks) deep.
//
// Tile = TR full image rows of one image (TR*W <= 96 pixels; 19x19 maps: 5 rows, 4 tiles per
// image, 1024 tiles per 256-image batch = 4 per CU, no ragged last round).  One workgroup of
// 8 waves; every wave keeps a 96 x 96 fp32 accumulator block (3 x 3 MFMA tiles, 144 VGPRs).
// Per 64-channel chunk c, one barrier:
//     load  raw(c+2) halo rows -> registers (coalesced 16-byte loads, 128 B per pixel)
//     D(c+1) depthwise on the vector ALU from raw[(c+1)&1] -> A[(c+1)&1] (bf16)
//     store raw(c+2) registers -> raw[c&1]
//     G(c)  36 v_mfma_f32_32x32x16_bf16 on A[c&1], B fragments from the register ring
#include "gemm_common.h"

#include <stdlib.h>

namespace {
using namespace bqk;

constexpr int KC = 64;             // channels per chunk
constexpr int SLOT = 144;          // bytes per halo pixel slot: 128 B of data + 16 B pad (odd number of
                                   // 16-byte units -> conflict-free ds_read_b128 / ds_write_b128)
constexpr int A_STR = 144;         // A chunk row stride, same reasoning
constexpr int MF = 3, RN = 3, WN = 8;
constexpr int NT = 64 * WN;        // 512 threads
constexpr int MT = 32 * MF;        // 96 rows of the M tile
constexpr int KBC = KC / 16;       // k-blocks per chunk

struct Raw3 { uint4 a, b, c; };

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu2(unsigned x) {           // ReLU on two packed bf16 (v_pk_max_i16)
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, x), z));
}

// per-thread, per-kernel constants of the halo loader (3 slots of 16 bytes per chunk)
struct Loader {
    int goff0, goff1, goff2;   // element offset of the slot's pixel from the tile's halo origin
    int loff0, loff1, loff2;   // LDS byte offset inside a raw buffer
    unsigned bits;             // halo row of slot i in bits [4i, 4i+4), "slot exists" in bit 12+i
    __device__ __forceinline__ bool act(int i) const { return (bits >> (12 + i)) & 1u; }
    __device__ __forceinline__ int ry(int i) const { return (bits >> (4 * i)) & 15u; }
};

struct TileInfo {
    const bf16_t* origin;   // pixel (img, y0-1, 0), channel 0
    int y0;                 // first output row
    int m0;                 // flattened index of the first output pixel
    int mvalid;             // output pixels of this tile
};

template <bool RELU>
__device__ __forceinline__ uint4 fix(uint4 v, bool rowok) {
    if (RELU) { v.x = relu2(v.x); v.y = relu2(v.y); v.z = relu2(v.z); v.w = relu2(v.w); }
    if (!rowok) v = make_uint4(0u, 0u, 0u, 0u);
    return v;
}

// The loader's per-thread constants live in LDS between uses (two 16-byte reads per use): the depthwise
// and matrix stages need every vector register.  The offset goes through an empty asm so the reads are
// redone at each use instead of being kept in registers.
__device__ __forceinline__ void park_loader(const Loader& L, unsigned char* smem, int off) {
    *reinterpret_cast<uint4*>(smem + off) = make_uint4(L.goff0, L.goff1, L.goff2, L.bits);
    *reinterpret_cast<uint4*>(smem + off + 16) = make_uint4(L.loff0, L.loff1, L.loff2, 0u);
}
__device__ __forceinline__ Loader fetch_loader(const unsigned char* smem, int off) {
    asm volatile("" : "+v"(off));
    const uint4 a = *reinterpret_cast<const uint4*>(smem + off);
    const uint4 b = *reinterpret_cast<const uint4*>(smem + off + 16);
    Loader L;
    L.goff0 = a.x; L.goff1 = a.y; L.goff2 = a.z; L.bits = a.w;
    L.loff0 = b.x; L.loff1 = b.y; L.loff2 = b.z;
    return L;
}

// rows outside the image load a valid pixel instead (first tile row) and are stored as zeros
__device__ __forceinline__ Raw3 raw_load(const Loader& L, const TileInfo& t, int c, int K, int H, int gclamp,
                                         int tid) {
    int coff = c * KC + (tid & 7) * 8;
    coff = coff < K - 8 ? coff : K - 8;
    Raw3 r;
    const bf16_t* base = t.origin + coff;
    const int y = t.y0 - 1;
    r.a = *reinterpret_cast<const uint4*>(base + ((unsigned)(y + L.ry(0)) < (unsigned)H ? L.goff0 : gclamp));
    r.b = *reinterpret_cast<const uint4*>(base + ((unsigned)(y + L.ry(1)) < (unsigned)H ? L.goff1 : gclamp));
    // unconditional (threads without a third slot re-load a valid pixel): a load under a branch makes
    // the number of outstanding loads path-dependent and every later s_waitcnt vmcnt conservative
    r.c = *reinterpret_cast<const uint4*>(
        base + ((L.act(2) && (unsigned)(y + L.ry(2)) < (unsigned)H) ? L.goff2 : gclamp));
    return r;
}

template <bool RELU>
__device__ __forceinline__ void raw_store(const Raw3& r, const Loader& L, const TileInfo& t, unsigned char* smem,
                                          int buf_off, int c, int K, int H, int tid) {
    if (c * KC + (tid & 7) * 8 >= K) return;       // channel tail of the last chunk: never read
    const int y = t.y0 - 1;
    if (L.act(0))
        *reinterpret_cast<uint4*>(smem + buf_off + L.loff0) = fix<RELU>(r.a, (unsigned)(y + L.ry(0)) < (unsigned)H);
    if (L.act(1))
        *reinterpret_cast<uint4*>(smem + buf_off + L.loff1) = fix<RELU>(r.b, (unsigned)(y + L.ry(1)) < (unsigned)H);
    if (L.act(2))
        *reinterpret_cast<uint4*>(smem + buf_off + L.loff2) = fix<RELU>(r.c, (unsigned)(y + L.ry(2)) < (unsigned)H);
}

#define STAMP(ev) do { if (stp && lane == 0) stp[ev] = __builtin_amdgcn_s_memtime(); } while (0)

// D stage on the matrix cores.  The depthwise conv of 16 channels x 16 pixels is nine MFMAs
//     D[ch][px] += Wdiag_t[ch][k] * X_t[k][px],   k = (term, channel'),  t = tap
// with v_mfma_f32_16x16x32_bf16: K = 32 holds the 16 channels twice, once for the high and once for the
// low bf16 half of the fp32 tap (w = hi + lo to 16 mantissa bits; products are exact in fp32), the A
// operand is the diagonal matrix of the tap and the B operand is simply the 16-byte piece of the shifted
// halo pixel (lane = pixel, 8 consecutive channels), read straight from the zero-padded LDS image.
// Measured motivation: the vector-ALU form of this stage (260 instructions per wave and chunk) does not
// overlap the partner wave's MFMAs - both contend for the SIMD's one vector issue port - and made the
// stage as long as the matrix stage itself; here it is 27 MFMAs of 16 cycles and ~80 vector instructions.
struct DwLane {
    int pbase[3];      // LDS byte offset of the lane's pixel for the wave's three 16-pixel blocks (+ channel piece)
    int aout;          // A-chunk byte offset of the lane's 4-channel crumb for the first block (next: + 16 rows)
    int tapoff;        // byte offset of the lane's channel in the LDS tap table (row 0)
    unsigned s0, s1, s2, s3;   // v_perm selectors building the diagonal A operand: the lane's one non-zero
                               // (its term's bf16 half of the tap) in the right half of the right dword
};

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void depthwise(unsigned char* smem, int raw_off, int a_off, int taps_off, int c, int Kp,
                                          int RW, const DwLane& dl, unsigned long long* stp = nullptr, int lane = 1) {
    f32x4v acc[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[b] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    const int tsrc = taps_off + dl.tapoff + c * KC * 4;
    const int src = raw_off;
    // operands one tap ahead of the MFMAs that consume them (the scheduling groups below keep it so)
    unsigned wv[2];
    uint4 x[2][3];
    wv[0] = *reinterpret_cast<const unsigned*>(smem + tsrc);
#pragma unroll
    for (int b = 0; b < 3; ++b) x[0][b] = *reinterpret_cast<const uint4*>(smem + src + dl.pbase[b]);
    STAMP(64);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (t + 1 < 9) {
            const int toff = (((t + 1) / 3) * RW + (t + 1) % 3) * SLOT;
            wv[(t + 1) & 1] = *reinterpret_cast<const unsigned*>(smem + tsrc + (t + 1) * Kp * 4);
#pragma unroll
            for (int b = 0; b < 3; ++b)
                x[(t + 1) & 1][b] = *reinterpret_cast<const uint4*>(smem + src + dl.pbase[b] + toff);
        }
        const unsigned w = wv[t & 1];
        const uint4 afrag = make_uint4(__builtin_amdgcn_perm(w, w, dl.s0), __builtin_amdgcn_perm(w, w, dl.s1),
                                       __builtin_amdgcn_perm(w, w, dl.s2), __builtin_amdgcn_perm(w, w, dl.s3));
#pragma unroll
        for (int b = 0; b < 3; ++b)
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afrag),
                                                             __builtin_bit_cast(bf16x8, x[t & 1][b]), acc[b], 0, 0, 0);
        STAMP(65 + t);
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const uint2 o = make_uint2(pack_bf16x2(acc[b][0], acc[b][1]), pack_bf16x2(acc[b][2], acc[b][3]));
        *reinterpret_cast<uint2*>(smem + a_off + dl.aout + b * 16 * A_STR) = o;
    }
    STAMP(77);
}

// G stage: one 64-channel chunk of A against the B ring; the ring refills itself PF k-blocks ahead and
// wraps into the next tile's first k-blocks (same weights).  Straight-line on purpose: all four k-blocks
// always run (k-blocks past K multiply zero rows of A by a valid, unused weight block), because a load
// under a branch makes the s_waitcnt vmcnt counts path-dependent and the compiler then drains the whole
// ring before every k-block.
template <int PF>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[MF][RN], uint4 (&bq)[PF][RN], const unsigned char* smem,
                                          int a_base, const uint4* __restrict__ bp0, int c, int KB, int KBtot, int dbg) {
    uint4 a[2][MF];                                // A fragments, one k-block ahead
#pragma unroll
    for (int i = 0; i < MF; ++i) a[0][i] = *reinterpret_cast<const uint4*>(smem + a_base + i * 32 * A_STR);
#pragma unroll
    for (int d = 0; d < KBC; ++d) {
        const int kb = c * KBC + d;
        if (d + 1 < KBC) {
#pragma unroll
            for (int i = 0; i < MF; ++i)
                a[(d + 1) & 1][i] = *reinterpret_cast<const uint4*>(smem + a_base + i * 32 * A_STR + (d + 1) * 32);
        }
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) mma<bf16_t>(acc[i][j], bq[d & (PF - 1)][j], a[d & 1][i]);
        const int nx = kb + PF;
        int idx = nx < KB ? nx : (d & (PF - 1));   // wrap: the slot next holds that k-block of the next tile
        idx = (dbg & 8) ? 0 : idx;                 // timing experiment: every fetch hits the same L1-resident block
#pragma unroll
        for (int j = 0; j < RN; ++j) bq[d & (PF - 1)][j] = bp0[((size_t)j * KBtot + idx) * 64];
    }
}

// accumulators -> folded BN (+ residual) (+ ReLU) -> bf16 -> LDS staging rows
__device__ __forceinline__ void acc_to_lds(const GemmParams& p, const f32x16 (&acc)[MF][RN], int nfb, int m0,
                                           int mvalid, int r32, int h, unsigned char* smem) {
    const bf16_t* __restrict__ res = reinterpret_cast<const bf16_t*>(p.residual);
    const int sstride = stage_stride<bf16_t>(p.Nstore);
#pragma unroll
    for (int j = 0; j < RN; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n0 = (nfb + j) * 32 + g * 8 + h * 4;
            if (n0 < p.Nstore) {
                const float4 sc = *reinterpret_cast<const float4*>(p.scale + n0);
                const float4 bi = *reinterpret_cast<const float4*>(p.bias + n0);
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    const int rl = i * 32 + r32;
                    float v[4];
                    v[0] = fmaf(acc[i][j][g * 4 + 0], sc.x, bi.x);
                    v[1] = fmaf(acc[i][j][g * 4 + 1], sc.y, bi.y);
                    v[2] = fmaf(acc[i][j][g * 4 + 2], sc.z, bi.z);
                    v[3] = fmaf(acc[i][j][g * 4 + 3], sc.w, bi.w);
                    if (res) {                     // staged by residual_to_lds at the crumb's own address
                        float rv[4];
                        load4<bf16_t>(reinterpret_cast<const bf16_t*>(smem + (size_t)rl * sstride) + n0, rv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rv[e];
                    }
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    store4<bf16_t>(reinterpret_cast<bf16_t*>(smem + (size_t)rl * sstride) + n0, v);
                }
            }
        }
    }
}

// residual rows -> LDS staging rows with coalesced 16-byte loads (reading the residual straight into the
// accumulator layout is 36 loads of 8-byte crumbs from 32 different rows each: measured 42k cycles per tile)
__device__ __forceinline__ void residual_to_lds(const GemmParams& p, int m0, int mvalid, int tid, unsigned char* smem) {
    const int sstride = stage_stride<bf16_t>(p.Nstore);
    const int ppr = p.Nstore / 8;
    const unsigned char* __restrict__ res = reinterpret_cast<const unsigned char*>(p.residual) + (size_t)m0 * p.ldo * 2;
    const size_t row_bytes = (size_t)p.ldo * 2;
    int row = tid / ppr, pc = tid - row * ppr;
    const int drow = NT / ppr, dpc = NT - drow * ppr;
    constexpr int BATCH = 6;
    const int rounds = (mvalid * ppr + NT - 1) / NT;
    for (int k0 = 0; k0 < rounds; k0 += BATCH) {
        uint4 v[BATCH];
        int off[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const bool ok = row < mvalid;
            const int rr = ok ? row : 0, pp = ok ? pc : 0;
            v[k] = *reinterpret_cast<const uint4*>(res + (size_t)rr * row_bytes + pp * 16);
            off[k] = ok ? rr * sstride + pp * 16 : -1;
            row += drow; pc += dpc;
            if (pc >= ppr) { pc -= ppr; ++row; }
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
            if (off[k] >= 0) *reinterpret_cast<uint4*>(smem + off[k]) = v[k];
    }
}

// staging rows -> global, whole 16-byte pieces, every thread busy (flattened (row, piece) walk)
__device__ __forceinline__ void lds_to_global(const GemmParams& p, int m0, int mvalid, int tid,
                                              const unsigned char* smem) {
    const int sstride = stage_stride<bf16_t>(p.Nstore);
    const int ppr = p.Nstore / 8;                  // 16-byte pieces per row
    unsigned char* __restrict__ out = reinterpret_cast<unsigned char*>(p.out) + (size_t)m0 * p.ldo * 2;
    const size_t row_bytes = (size_t)p.ldo * 2;
    int row = tid / ppr, pc = tid - row * ppr;
    const int drow = NT / ppr, dpc = NT - drow * ppr;
    while (row < mvalid) {
        *reinterpret_cast<uint4*>(out + (size_t)row * row_bytes + pc * 16) =
            *reinterpret_cast<const uint4*>(smem + (size_t)row * sstride + pc * 16);
        row += drow; pc += dpc;
        if (pc >= ppr) { pc -= ppr; ++row; }
    }
}

__device__ __forceinline__ TileInfo tile_info(int v, int ntiles, int TPI, int TR, int H, int W, int ldi,
                                               const bf16_t* in) {
    TileInfo t;
    const int tile = xcd_tile(v, ntiles);
    const int img = tile / TPI, part = tile - img * TPI;
    t.y0 = part * TR;
    const int rows = H - t.y0 < TR ? H - t.y0 : TR;
    t.m0 = (img * H + t.y0) * W;
    t.mvalid = rows * W;
    t.origin = in + (ptrdiff_t)(t.m0 - W) * ldi;
    return t;
}

// pad columns of both halo buffers = 0 ('same' padding in x); rows outside the image are stored as zeros
__device__ __forceinline__ void zero_pads(unsigned char* smem, int tid, int RR, int RW, int raw_bytes) {
    if (tid < 2 * RR * 2 * 8) {
        const int per = RR * 2 * 8;
        const int b = tid / per, rem = tid - b * per;
        const int k = rem >> 3, ry = k >> 1, rx = (k & 1) ? RW - 1 : 0;
        *reinterpret_cast<uint4*>(smem + b * raw_bytes + (ry * RW + rx) * SLOT + (rem & 7) * 16) =
            make_uint4(0u, 0u, 0u, 0u);
    }
}

template <bool RELU, bool PERSIST, int PF>
__global__ void __launch_bounds__(NT) sepconv_mid_kernel(const GemmParams p, const int TR, const int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = p.W, H = p.H;
    const int RW = W + 2, RR = TR + 2;
    const int raw_bytes = RR * RW * SLOT;
    const int a_off0 = 2 * raw_bytes;
    const int K = p.K, KB = K / 16, NC = (K + KC - 1) / KC;
    const int ldi = p.ldi;
    const int TPI = (H + TR - 1) / TR;             // tiles per image
    const bf16_t* __restrict__ in = reinterpret_cast<const bf16_t*>(p.in);

    // ---- per-thread constants ----
    Loader L;
    {
        int go[3], lo[3];
        unsigned bits = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int sl = (tid >> 3) + 64 * i;
            const bool act = sl < RR * W;
            const int sc = act ? sl : 0;
            const int ry = sc / W, x = sc - ry * W;
            bits |= (unsigned)ry << (4 * i);
            bits |= (act ? 1u : 0u) << (12 + i);
            go[i] = (ry * W + x) * ldi;
            lo[i] = (ry * RW + x + 1) * SLOT + (tid & 7) * 16;
        }
        L.goff0 = go[0]; L.goff1 = go[1]; L.goff2 = go[2];
        L.loff0 = lo[0]; L.loff1 = lo[1]; L.loff2 = lo[2];
        L.bits = bits;
    }
    const int lpark = a_off0 + 2 * MT * A_STR + 9 * (NC * KC) * 4 + tid * 32;   // behind the tap table
    park_loader(L, smem, lpark);
    const int gclamp = W * ldi;                    // pixel (y0, 0): always inside the image
    // depthwise lane map: wave -> 16-channel block (wave & 3) and three 16-pixel blocks; lane -> pixel, k-group
    const int Kp = NC * KC;                        // row length of the tap table: whole chunks, zero beyond the layer's channels
    const int taps_off = a_off0 + 2 * MT * A_STR;
    DwLane dl;
    {
        const int cb = wave & 3, pxb0 = (wave >> 2) * 3;
        const int px = lane & 15, g = lane >> 4;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int P = (pxb0 + b) * 16 + px;    // pixel of the M tile (row-major over TR rows of W)
            const int row = P / W, x = P - row * W;
            dl.pbase[b] = (row * RW + x) * SLOT + (cb * 2 + (g & 1)) * 16;
        }
        dl.aout = (pxb0 * 16 + px) * A_STR + cb * 32 + g * 8;
        const int ch = lane & 15;                  // A operand row = channel of the block
        dl.tapoff = (cb * 16 + ch) * 4;
        // lane (ch, g) holds k = 8g..8g+7 = (term g>>1, channels 8(g&1)..+7): non-zero only at its own channel
        const bool active = (ch >> 3) == (g & 1);
        const int j = ch & 7;
        // table entry = hi | lo << 16: bytes (1,0) are the high term, (3,2) the low term; 0x0c selects zero
        const unsigned src2 = (g >> 1) ? 0x0302u : 0x0100u;
        const unsigned sel = (j & 1) ? (src2 << 16) | 0x0c0cu : 0x0c0c0000u | src2;
        dl.s0 = (active && (j >> 1) == 0) ? sel : 0x0c0c0c0cu;
        dl.s1 = (active && (j >> 1) == 1) ? sel : 0x0c0c0c0cu;
        dl.s2 = (active && (j >> 1) == 2) ? sel : 0x0c0c0c0cu;
        dl.s3 = (active && (j >> 1) == 3) ? sel : 0x0c0c0c0cu;
    }
    // MFMA lane map
    const int r32 = lane & 31, h = lane >> 5;
    const int nfb = wave * RN;
    const bool first_half = (p.dbg & 32) ? true : ((p.dbg & 64) ? (wave & 1) == 0 : wave < WN / 2);
    const uint4* __restrict__ wp = reinterpret_cast<const uint4*>(p.wp);
    const uint4* bp0 = wp + ((size_t)nfb * p.KBtot + p.kb0) * 64 + lane;

    int v = blockIdx.x;
    if (v >= ntiles) return;
    unsigned long long* stp = (p.stamps && blockIdx.x < 64) ? p.stamps + ((size_t)blockIdx.x * 8 + wave) * 128 : nullptr;
    STAMP(0);
    TileInfo T = tile_info(v, ntiles, TPI, TR, H, W, ldi, in);
    Raw3 r0 = raw_load(L, T, 0, K, H, gclamp, tid);
    Raw3 r1 = raw_load(L, T, 1, K, H, gclamp, tid);
    // first B fragments
    uint4 bq[PF][RN];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int idx = d < KB ? d : KB - 1;
#pragma unroll
        for (int j = 0; j < RN; ++j) bq[d][j] = bp0[((size_t)j * p.KBtot + idx) * 64];
    }
    // zero the pad columns of both halo buffers and the unused A rows (once; see the tile loop for pads)
    zero_pads(smem, tid, RR, RW, raw_bytes);
    // tap table in LDS: fp32 tap split into two bf16 terms, hi | lo << 16, [9][Kp]
    for (int i = tid; i < 9 * Kp; i += NT) {
        const int t = i / Kp, ch = i - t * Kp;
        const float w = ch < ldi ? p.dw[t * ldi + ch] : 0.f;
        const bf16_t hi = (bf16_t)w;
        const float r = w - (float)hi;
        const bf16_t lo = (bf16_t)r;
        *reinterpret_cast<unsigned*>(smem + taps_off + i * 4) =
            (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
    }
    for (int i = tid; i < 2 * (MT - TR * W) * 9; i += NT) {
        const int b = i / ((MT - TR * W) * 9), rem = i - b * (MT - TR * W) * 9;
        *reinterpret_cast<uint4*>(smem + a_off0 + b * MT * A_STR + TR * W * A_STR + rem * 16) =
            make_uint4(0u, 0u, 0u, 0u);
    }

    for (;;) {
        // ---- tile prologue: raw(0), raw(1) are in r0, r1 ----
        raw_store<RELU>(r0, L, T, smem, 0, 0, K, H, tid);
        raw_store<RELU>(r1, L, T, smem, raw_bytes, 1, K, H, tid);
        Raw3 rr = raw_load(L, T, 2, K, H, gclamp, tid);
        f32x16 acc[MF][RN];
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        STAMP(1);
        __syncthreads();
        STAMP(2);
        depthwise(smem, 0, a_off0, taps_off, 0, Kp, RW, dl);
        STAMP(3);
        __syncthreads();
        STAMP(4);

        for (int c = 0; c < NC; ++c) {
            const int cur = c & 1, nxt = cur ^ 1;
            // raw[cur]'s last reader D(c) finished before the previous barrier
            if (c + 2 < NC && !(p.dbg & 16))
                raw_store<RELU>(rr, fetch_loader(smem, lpark), T, smem, cur * raw_bytes, c + 2, K, H, tid);
            // D (vector ALU) and G (matrix cores) of one wave are independent; the two waves of a SIMD run
            // them in opposite order so one's depthwise overlaps the other's MFMAs.  The halo loads of chunk
            // c+3 are issued right before the wave's own D stage, which never waits on vmcnt: issued before
            // G they would sit in front of the B ring's loads in the in-order vmcnt queue and every k-block
            // would wait out their HBM latency.
            const bool do_d = c + 1 < NC && !(p.dbg & 1);
            if (first_half) {
                if (!(p.dbg & 16)) rr = raw_load(fetch_loader(smem, lpark), T, c + 3, K, H, gclamp, tid);
                if (do_d)
                    depthwise(smem, nxt * raw_bytes, a_off0 + nxt * MT * A_STR, taps_off, c + 1, Kp, RW, dl,
                              c == 5 ? stp : nullptr, lane);
            }
            STAMP(5 + 4 * c);
            const int a_base = a_off0 + cur * MT * A_STR + r32 * A_STR + h * 16;
            if (!(p.dbg & 2)) mma_chunk<PF>(acc, bq, smem, a_base, bp0, c, KB, p.KBtot, p.dbg);
            STAMP(6 + 4 * c);
            if (!first_half) {
                if (!(p.dbg & 16)) rr = raw_load(fetch_loader(smem, lpark), T, c + 3, K, H, gclamp, tid);
                if (do_d)
                    depthwise(smem, nxt * raw_bytes, a_off0 + nxt * MT * A_STR, taps_off, c + 1, Kp, RW, dl,
                              c == 5 ? stp : nullptr, lane);
            }
            STAMP(7 + 4 * c);
            __syncthreads();
            STAMP(8 + 4 * c);
        }

        // ---- tile epilogue; the next tile's first halo chunks load underneath its store pass ----
        const int vn = v + gridDim.x;
        const bool more = PERSIST && vn < ntiles;
        if (!(p.dbg & 4)) {
            if (p.residual) {
                STAMP(60);
                residual_to_lds(p, T.m0, T.mvalid, tid, smem);
                STAMP(61);
                __syncthreads();
                STAMP(62);
            }
            acc_to_lds(p, acc, nfb, T.m0, T.mvalid, r32, h, smem);
        }
        TileInfo Tn = T;
        if (more) {                                // the accumulators are dead: registers to spare
            Tn = tile_info(vn, ntiles, TPI, TR, H, W, ldi, in);
            r0 = raw_load(L, Tn, 0, K, H, gclamp, tid);
            r1 = raw_load(L, Tn, 1, K, H, gclamp, tid);
        }
        STAMP(57);
        __syncthreads();
        STAMP(58);
        if (!(p.dbg & 4)) lds_to_global(p, T.m0, T.mvalid, tid, smem);
        STAMP(59);
        if (!more) break;
        __syncthreads();
        zero_pads(smem, tid, RR, RW, raw_bytes);   // the staging rows overwrote them
        T = Tn;
        v = vn;
    }
}

template <bool RELU>
int launch_mid(const GemmParams& p, int TR, int num_cus, hipStream_t s) {
    // One tile per workgroup.  The persistent form (PERSIST = true, several tiles per workgroup) is kept for
    // experiments only: its epilogue staging rows overwrite the LDS tap table and the parked loader
    // constants, so it must not be launched until the epilogue stages 32 rows at a time.
    static const bool persist = false;
    static const int pf = getenv("BQ_MID_PF") ? atoi(getenv("BQ_MID_PF")) : 4;
    auto kern = persist ? (pf == 2 ? sepconv_mid_kernel<RELU, true, 2> : sepconv_mid_kernel<RELU, true, 4>)
                        : (pf == 2 ? sepconv_mid_kernel<RELU, false, 2> : sepconv_mid_kernel<RELU, false, 4>);
    const int RW = p.W + 2, RR = TR + 2;
    size_t lds = (size_t)2 * RR * RW * SLOT + 2 * MT * A_STR + (size_t)9 * ((p.K + KC - 1) / KC * KC) * 4 + NT * 32;
    const size_t stage = (size_t)MT * (p.Nstore * 2 + 16);
    if (stage > lds) lds = stage;
    static size_t lds_set = 0;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        lds_set = lds;
    }
    const int TPI = (p.H + TR - 1) / TR;
    const int n_img = p.M / (p.H * p.W);
    const int ntiles = n_img * TPI;
    static const int env_wgs = getenv("BQ_MID_WGS") ? atoi(getenv("BQ_MID_WGS")) : 0;
    int grid = env_wgs > 0 ? env_wgs : num_cus;
    if (grid > ntiles || !persist) grid = ntiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, p, TR, ntiles);
    return (int)hipGetLastError();
}

}  // namespace

// bf16 SeparableConv2D with 768 (padded) output channels on maps whose rows tile 96 pixels well
static int mid_rows(int H, int W) {
    const int TR = MT / W;
    return TR > H ? H : TR;
}

bool mid_supported(int dtype, int prod, int nfp, int H, int W, int K, int M, int Nstore) {
    if (dtype != 1 || (prod != PROD_DW && prod != PROD_DW_RELU) || nfp != WN * RN || K % 16 != 0) return false;
    if (W > 31 || W < 8 || M % (H * W) != 0) return false;     // W <= 31: >= 3 rows per tile, lanes for row pairs
    const int TR = mid_rows(H, W);
    if (TR < 2 || ((TR + 1) / 2) * W > 64) return false;
    if ((TR + 2) * W > 3 * 64) return false;                   // loader: 3 slots per thread
    const size_t lds = (size_t)2 * (TR + 2) * (W + 2) * SLOT + 2 * MT * A_STR;
    const size_t stage = (size_t)MT * (Nstore * 2 + 16);
    return lds <= 160 * 1024 && stage <= 160 * 1024 && K >= 3 * KC;
}

int launch_sepconv_mid(int prod, const GemmParams& p, int num_cus, hipStream_t s) {
    const int TR = mid_rows(p.H, p.W);
    return prod == PROD_DW_RELU ? launch_mid<true>(p, TR, num_cus, s) : launch_mid<false>(p, TR, num_cus, s);
}
