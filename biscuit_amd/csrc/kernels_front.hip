// The front of the network in one streaming kernel (round 4): uint8 tile -> per-image standardisation -> block1_conv1
// (3x3 / stride 2 / valid, 3 -> 32) + BN + ReLU -> block1_conv2 (3x3 / valid, 32 -> 64) + BN + ReLU.
//
// As three kernels (stage_u8, stem1, the conv2 tile kernel) the 299x299x3 tile went to HBM as planar 16-bit (137 MB per batch
// of 256), came back, went out again as 149x149x32 (364 MB), came back again: 1.78 GB and 0.59 ms for 69 MB of input and
// 708 MB of output.  Here the two intermediate tensors never leave the CU:
//  * the per-tile statistics (exact integer sums, kernels_misc.hip: stage_stats_kernel) are the only pre-pass;
//  * a work item is a strip of <= 16 conv2 output columns x a band of rows; the wave walks down, one conv2 row per step,
//    two conv2 rows per step, and keeps the four stem rows they need in a 4 KB ring in LDS;
//  * a step fetches the uint8 rows of its two new stem rows (three rows of 99 bytes each, as aligned dwords, one step ahead), and
//    builds the stem convolution's MFMA operand from them: lane (pixel slot, k-group) picks its eight bytes of the
//    27-byte window (k = (dy, dx, c), the order of the weights), standardises them with the arithmetic of the staging
//    kernel -- ((float)u8 - mean) * inv, rounded to the storage type: the same values the planar tensor held -- ;
//  * block1_conv1 on the matrix cores: K = 27 (padded to 32), the fp32 weights split into two IEEE halves, w = hi + lo /
//    2^11 (22 significand bits, as the MC head does): two v_mfma_f32_16x16x32_f16 per 16 pixels and 16 channels, fp32
//    accumulation -- the products are exact, only the accumulation order differs from the vector-ALU kernel it replaces
//    (0.18 ms of 864 fmas per pixel became 4 MFMAs per 16 pixels: strips are 14 conv2 columns wide so that their 16 stem
//    columns are exactly one MFMA pixel tile);
//  * its accumulator layout (lane = pixel slot, 8 consecutive channels with the fragment-pair interleave) IS the operand
//    layout of the next convolution: BN + ReLU, one ds_write_b128 into the ring, and block1_conv2's nine taps are nine
//    ds_read_b128 at (row + dy, slot + dx) -- no transposition, no im2col buffer; 36 MFMAs per row;
//  * epilogue straight from the accumulators: BN + ReLU, two 16-byte stores per lane (64 contiguous bytes of 16 pixels each).
// 11 independent waves per CU, no workgroup barrier after the weights are in LDS.
#include "gemm_common.h"

// ablation switches of tools/ubench/stream_bench.hip (timing only, wrong results): 1 no stores, 4 no conv2 MFMAs, 8 no uint8
// loads, 32 no stem (operand build + MFMAs + epilogue).  The product build has none of them.
#ifndef FRONT_ABL
#define FRONT_ABL 0
#endif

namespace {
using namespace bqk;

typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

constexpr int PX = 299, SO = 149, CO = 147;     // tile, stem output, conv2 output edge
constexpr int ROWB = PX * 3;                    // bytes per uint8 row
#ifndef FRONT_NW
#define FRONT_NW 11
#endif
constexpr int NWF = FRONT_NW;                   // waves per workgroup (11 = strips per image row: at batch 256 every wave gets 6 items)
constexpr int SW = 14;                          // conv2 columns per strip: 16 stem columns = ONE MFMA pixel tile for the stem
constexpr float LO_SCALE = 1.0f / 2048.0f;      // weights.py: HEAD_SPLIT_SCALE

template <typename T>
struct FrontParams {
    const uint8_t* tiles;               // [n][299][299][3]
    const unsigned long long* stats;    // [n][2]: integer sum and sum of squares of the tile's bytes
    const uint4* ws16;                  // stem weights, f16: [hi, lo][2 fragments][64 lanes] x 16 B
    const float* s_scale;               // [32] folded BN of block1_conv1
    const float* s_bias;
    const uint4* wc16;                  // conv2 weights in T: [9 taps][4 fragments][64 lanes] x 16 B
    const float* c_scale;               // [64] folded BN of block1_conv2
    const float* c_bias;
    T* out;                             // [n][147][147][64]
    int n, nstrips, nbands, items;
};

__device__ __forceinline__ void span(int s, int n, int ns, int& x0, int& nc) {
    const int base = n / ns, rem = n - base * ns;
    x0 = s * base + (s < rem ? s : rem);
    nc = base + (s < rem ? 1 : 0);
}

template <typename T> __device__ __forceinline__ f32x4 mmaT(const uint4& a, const uint4& b, const f32x4& c);
template <> __device__ __forceinline__ f32x4 mmaT<f16_t>(const uint4& a, const uint4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mmaT<bf16_t>(const uint4& a, const uint4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// the staged value of one byte as an IEEE half (the stem's MFMAs are f16 for both storage types): T(((float)u - mean) * inv).
// A bf16 value of this size (|v| from ~1e-3 to ~6) is exactly representable in f16.
template <typename T> __device__ __forceinline__ _Float16 staged_f16(unsigned byte, float mean, float inv) {
    const float v = ((float)byte - mean) * inv;
    if constexpr (H16<T>::F16) return (_Float16)v;
    else return (_Float16)(float)(bf16_t)v;
}

template <typename T>
__global__ void __launch_bounds__(NWF * 64) front_stream_kernel(const FrontParams<T> p) {
    bq_f16_saturate();                           // (the stem's operands are f16 whatever T is)
    constexpr int NT = NWF * 64;
    constexpr int WS_BYTES = 2 * 2 * 1024;      // stem weights hi | lo
    constexpr int WC_BYTES = 9 * 4 * 1024;      // conv2 weights
    constexpr int WC_OFF = WS_BYTES;
    constexpr int SB_OFF = WC_OFF + WC_BYTES;   // s_scale[32] s_bias[32] c_scale[64] c_bias[64], fp32
    constexpr int PRIV_OFF = SB_OFF + 192 * 4;
    constexpr int U8_PITCH = 112;               // bytes per staged uint8 row (99 + up to 3 of alignment, as 26 dwords)
    constexpr int U8_DW = 26;
    constexpr int PXP = 96;                     // bytes per pixel of a ring row: 32 channels + 32 of padding -- 6 slots of 16 B, = 2 (mod 4):
                                                // the conv2 operand's ds_read_b128 (lane -> pixel l & 15 (+ dx), k-group l >> 4) is conflict-free
                                                // over its four 16-lane groups and the ring's ds_write_b128 2-way; at 64 B (4 slots) the reads
                                                // were 2-way and the writes 4-way: 27 % of the kernel's LDS cycles (profiles/r04_sq_counters.txt)
    constexpr int RING_PITCH = 16 * PXP;        // one stem row of the strip: 16 pixels x 32 channels
    constexpr int NRING = 4;                    // stem rows y .. y + 3: two conv2 rows per step
    constexpr int PRIV = 3 * U8_PITCH + NRING * RING_PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < WS_BYTES / 16; i += NT) *reinterpret_cast<uint4*>(smem + i * 16) = p.ws16[i];
    for (int i = tid; i < WC_BYTES / 16; i += NT) *reinterpret_cast<uint4*>(smem + WC_OFF + i * 16) = p.wc16[i];
    if (tid < 32) {
        reinterpret_cast<float*>(smem + SB_OFF)[tid] = p.s_scale[tid];
        reinterpret_cast<float*>(smem + SB_OFF)[32 + tid] = p.s_bias[tid];
    }
    if (tid >= 64 && tid < 128) {
        reinterpret_cast<float*>(smem + SB_OFF)[64 + tid - 64] = p.c_scale[tid - 64];
        reinterpret_cast<float*>(smem + SB_OFF)[128 + tid - 64] = p.c_bias[tid - 64];
    }
    __syncthreads();                            // the only workgroup barrier

    unsigned char* const u8l = smem + PRIV_OFF + wave * PRIV;          // three staged uint8 rows
    unsigned char* const ring = u8l + 3 * U8_PITCH;                    // four stem rows
    const int px = lane & 15, g = lane >> 4;
    const float* const sbs = reinterpret_cast<const float*>(smem + SB_OFF) + 8 * g;         // stem scale (+32: bias)
    const float* const sbc = reinterpret_cast<const float*>(smem + SB_OFF) + 64 + 8 * g;    // conv2 scale (+64: bias)
    // this lane's eight window bytes k = 8 g + j -> (window row k / 9, byte k % 9); k >= 27 is padding (zero weights): byte 0 of row 0
    int koff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * g + j;
        koff[j] = k < 27 ? (k / 9) * U8_PITCH + (k % 9) : 0;
    }
    const int wgx = xcd_tile(blockIdx.x, gridDim.x);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.out, 0, (int)((size_t)p.n * CO * CO * 64 * sizeof(T)), 0x00020000);

    for (int it = 0;; ++it) {
        const int item = __builtin_amdgcn_readfirstlane((it * (int)gridDim.x + wgx) * NWF + wave);
        if (item >= p.items) break;
        const int strip = item % p.nstrips;
        const int t1 = item / p.nstrips;
        const int band = t1 % p.nbands;
        const int img = t1 / p.nbands;
        int x0, nc, y0, nr;
        span(strip, CO, p.nstrips, x0, nc);      // conv2 columns [x0, x0 + nc), nc <= 14; stem columns x0 .. x0 + 15
        span(band, CO, p.nbands, y0, nr);        // conv2 rows [y0, y0 + nr); stem rows y0 .. y0 + nr + 1
        const int y1 = y0 + nr;
        // statistics of the tile: the arithmetic of stage_apply_kernel (float64 from the exact integer sums)
        float mean, inv;
        {
            const double nb = (double)(PX * PX * 3);
            const double mean_d = (double)p.stats[2 * img] / nb;
            double var = (double)p.stats[2 * img + 1] / nb - mean_d * mean_d;
            if (var < 0) var = 0;
            const double sd = sqrt(var), floor_sd = 1.0 / sqrt(nb);
            mean = (float)mean_d;
            inv = (float)(1.0 / (sd > floor_sd ? sd : floor_sd));
        }
        // The three uint8 rows of stem row sy, as 26 ALIGNED dwords each from (row start + 6 x0) & ~3: lanes 0..25 row 0,
        // 32..57 row 1 (one instruction for two rows), a second instruction row 2.  An aligned dword may begin up to three bytes
        // in front of the buffer (a tile view that is not 4-byte aligned) or end up to three bytes behind it (the last tile's
        // last row): such a load is moved onto the buffer's first / last four bytes and shifted, so that the bytes that exist
        // arrive where the aligned dword would have had them.  Branch-free, one load per row group: under `if` the compiler
        // made them flat loads with a vmcnt(0) each -- no prefetch at all, 3 us per row step.
        const long long tile_off = (long long)img * (PX * ROWB);
        const uintptr_t tb = reinterpret_cast<uintptr_t>(p.tiles);
        const long long last4 = (long long)p.n * (PX * ROWB) - 4;
        // (the shifts are applied where the dword is USED, a step later: applied here they would be a use right behind the load)
        auto load_dw = [&](long long first_byte, int dw, int& shr_out, int& shl_out) -> unsigned {
            long long a4 = (long long)(((tb + (uintptr_t)first_byte) & ~(uintptr_t)3) - tb) + 4 * dw;
            const int shl = a4 < 0 ? (int)(-a4) * 8 : 0;                       // (1..3 bytes in front of the buffer)
            long long over = a4 - last4;                                        // 1..3: straddles the end; more: wholly behind it --
            over = over < 0 ? 0 : (over > 3 ? 3 : over);                        // only bytes of pixel slots past the image, any value will do
            a4 = a4 < 0 ? 0 : (a4 > last4 ? last4 : a4);
            shr_out = (int)over * 8;
            shl_out = shl;
            if constexpr (FRONT_ABL & 8) return (unsigned)a4;
            return *reinterpret_cast<const unsigned*>(p.tiles + a4);
        };
        const int lrow = lane >> 5, ldw = lane & 31;
        struct U8 { unsigned a, b; int a_r, a_l, b_r, b_l; };
        U8 uA{}, uB{};                              // the uint8 rows of the two stem rows of the next step
        auto load_u8 = [&](int sy, U8& u) {
            const int syc = sy > SO - 1 ? SO - 1 : sy;
            const long long r0 = tile_off + (long long)(2 * syc) * ROWB + 6 * x0;
            u.a = load_dw(r0 + lrow * ROWB, ldw, u.a_r, u.a_l);
            u.b = load_dw(r0 + 2 * ROWB, ldw, u.b_r, u.b_l);
        };
        const uint8_t* const tile = p.tiles + tile_off;
        // stem row sy -> ring slot `slot` (its uint8 rows are in u)
        auto stem_row = [&](int sy, int slot, const U8& u) {
            const unsigned ua = u.a, ub = u.b;
            const int ua_r = u.a_r, ua_l = u.a_l, ub_r = u.b_r, ub_l = u.b_l;
            const int syc = sy > SO - 1 ? SO - 1 : sy;
            const uint8_t* r0 = tile + (size_t)(2 * syc) * ROWB + 6 * x0;
            // byte offset of the window's first byte inside each staged row (rows are 897 bytes apart: the shifts differ)
            const int sh0 = (int)(reinterpret_cast<uintptr_t>(r0) & 3), sh1 = (int)(reinterpret_cast<uintptr_t>(r0 + ROWB) & 3),
                      sh2 = (int)(reinterpret_cast<uintptr_t>(r0 + 2 * ROWB) & 3);
            {   // every lane stores (lanes past the 26 dwords into the row's padding, both halves the same third row): under
                // `if (ldw < 26)` the compiler loses count of what is in flight and waits for the previous step's stores too
                if constexpr (FRONT_ABL & 32) {
                if (ua == 0x12345u) *reinterpret_cast<unsigned*>(ring + slot * RING_PITCH + lane * 4) = ua ^ ub;
                return;
            }
            const int dwc = ldw < U8_DW ? ldw : U8_PITCH / 4 - 1;
                *reinterpret_cast<unsigned*>(u8l + lrow * U8_PITCH + 4 * dwc) = (ua >> ua_r) << ua_l;
                *reinterpret_cast<unsigned*>(u8l + 2 * U8_PITCH + 4 * dwc) = (ub >> ub_r) << ub_l;
            }
            {
                int s = px;                                     // pixel slot = stem column x0 + px
                if (x0 + s > SO - 1) s = SO - 1 - x0;           // (slots past the image: a valid pixel, never used)
                _Float16 h[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 8 * g + j;                    // (koff[j] holds row and byte; the shift depends on the row)
                    const int sh = k < 9 ? sh0 : (k < 18 ? sh1 : sh2);
                    const unsigned byte = u8l[koff[j] + sh + 6 * s];
                    h[j] = staged_f16<T>(byte, mean, inv);
                }
                uint4 b;
                b.x = __builtin_bit_cast(unsigned short, h[0]) | ((unsigned)__builtin_bit_cast(unsigned short, h[1]) << 16);
                b.y = __builtin_bit_cast(unsigned short, h[2]) | ((unsigned)__builtin_bit_cast(unsigned short, h[3]) << 16);
                b.z = __builtin_bit_cast(unsigned short, h[4]) | ((unsigned)__builtin_bit_cast(unsigned short, h[5]) << 16);
                b.w = __builtin_bit_cast(unsigned short, h[6]) | ((unsigned)__builtin_bit_cast(unsigned short, h[7]) << 16);
                f32x4 ah[2], al[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const uint4 wh = *reinterpret_cast<const uint4*>(smem + (f * 64 + lane) * 16);
                    const uint4 wl = *reinterpret_cast<const uint4*>(smem + ((2 + f) * 64 + lane) * 16);
                    ah[f] = mmaT<f16_t>(wh, b, (f32x4){0.f, 0.f, 0.f, 0.f});
                    al[f] = mmaT<f16_t>(wl, b, (f32x4){0.f, 0.f, 0.f, 0.f});
                }
                // lane (slot, g): channels 8 g .. 8 g + 7 (fragment 0: + 0..3, fragment 1: + 4..7); BN + ReLU, rounded
                const float4 s0 = *reinterpret_cast<const float4*>(sbs), s1 = *reinterpret_cast<const float4*>(sbs + 4);
                const float4 c0 = *reinterpret_cast<const float4*>(sbs + 32), c1 = *reinterpret_cast<const float4*>(sbs + 36);
                float v[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = fmaf(al[0][i], LO_SCALE, ah[0][i]);
                    v[4 + i] = fmaf(al[1][i], LO_SCALE, ah[1][i]);
                }
                uint4 o;
                o.x = H16<T>::pack2(fmaxf(fmaf(v[0], s0.x, c0.x), 0.f), fmaxf(fmaf(v[1], s0.y, c0.y), 0.f));
                o.y = H16<T>::pack2(fmaxf(fmaf(v[2], s0.z, c0.z), 0.f), fmaxf(fmaf(v[3], s0.w, c0.w), 0.f));
                o.z = H16<T>::pack2(fmaxf(fmaf(v[4], s1.x, c1.x), 0.f), fmaxf(fmaf(v[5], s1.y, c1.y), 0.f));
                o.w = H16<T>::pack2(fmaxf(fmaf(v[6], s1.z, c1.z), 0.f), fmaxf(fmaf(v[7], s1.w, c1.w), 0.f));
                *reinterpret_cast<uint4*>(ring + slot * RING_PITCH + px * PXP + g * 16) = o;
            }
        };

        load_u8(y0, uA);
        load_u8(y0 + 1, uB);
        stem_row(y0, 0, uA);
        stem_row(y0 + 1, 1, uB);
        load_u8(y0 + 2, uA);
        load_u8(y0 + 3, uB);
        // Four stores the hardware drops (offset past the buffer), so that the vector-memory queue looks the same on entry
        // to the loop as on its back edge -- four loads, then four stores: the compiler then waits for the LOADS at the top of
        // a step; with nothing here it merged "loads only" and "loads + stores" into the weaker wait and every step waited for
        // the previous step's stores to be acknowledged.
#pragma unroll
        for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b128((u32x4s){0u, 0u, 0u, 0u}, orsrc, (int)(0xffffff00u + 16 * q), 0, 0);
        int s0 = 0, s1 = 1, s2 = 2, s3 = 3;      // ring slots of stem rows y .. y + 3
        unsigned ooff = px < nc ? (unsigned)(((((size_t)img * CO + y0) * CO + x0 + px) * 64 + 8 * g) * sizeof(T)) : 0xffffff00u;
        const unsigned ostep = px < nc ? (unsigned)(CO * 64 * sizeof(T)) : 0u;
        // TWO conv2 rows per step: they share two of their three stem rows and every weight fragment, so a step reads 36 + 12
        // KB from LDS for two rows where one row per step read 36 + 9 per row -- the conv2 stage ran at ~45 % of both the LDS
        // bandwidth and the matrix pipe, each waiting for the other.
        for (int y = y0; y < y1; y += 2, ooff += 2 * ostep) {
            stem_row(y + 2, s2, uA);
            stem_row(y + 3, s3, uB);
            {   // the next step's stem rows, one step ahead (past the band: re-reads of its last row)
                const int last = y1 + 1;
                load_u8(y + 4 <= last ? y + 4 : last, uA);
                load_u8(y + 5 <= last ? y + 5 : last, uB);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- block1_conv2: nine taps, each one k-step of 32 channels; accumulators [row][fragment]
            f32x4 acc[2][4];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[r][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
            {   // an explicit pipeline: the twelve B operands first, the weight fragments one tap (four fragments) ahead of their
                // MFMAs -- LDS returns in order, so the wait in front of a tap is lgkmcnt(4); left as `read; mfma` the compiler
                // waited lgkmcnt(0) in front of most MFMAs, an LDS round trip each
                uint4 b[4][3];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int slot = r == 0 ? s0 : (r == 1 ? s1 : (r == 2 ? s2 : s3));
                    const unsigned char* rp = ring + slot * RING_PITCH + (px < SW ? px : SW - 1) * PXP + g * 16;   // (slots 14, 15 are not outputs)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) b[r][dx] = *reinterpret_cast<const uint4*>(rp + dx * PXP);
                }
                uint4 w[2][4];
                auto fetch = [&](int t, uint4 (&dst)[4]) {
#pragma unroll
                    for (int f = 0; f < 4; ++f) dst[f] = *reinterpret_cast<const uint4*>(smem + WC_OFF + ((t * 4 + f) * 64 + lane) * 16);
                };
                fetch(0, w[0]);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (t + 1 < 9) fetch(t + 1, w[(t + 1) & 1]);
                    const int dy = t / 3, dx = t % 3;
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int f = 0; f < 4; ++f) {
                            if constexpr (FRONT_ABL & 4) { acc[r][f][0] += __uint_as_float(w[t & 1][f].x ^ b[dy + r][dx].x); acc[r][f][1] += __uint_as_float(w[t & 1][f].y ^ b[dy + r][dx].w); }
                            else acc[r][f] = mmaT<T>(w[t & 1][f], b[dy + r][dx], acc[r][f]);
                        }
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const unsigned off = r == 0 ? ooff : (y + 1 < y1 ? ooff + ostep : 0xffffff00u);     // (an odd band: the last step's second row is dropped)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float4 a0 = *reinterpret_cast<const float4*>(sbc + 32 * q), a1 = *reinterpret_cast<const float4*>(sbc + 32 * q + 4);
                    const float4 b0 = *reinterpret_cast<const float4*>(sbc + 64 + 32 * q), b1 = *reinterpret_cast<const float4*>(sbc + 64 + 32 * q + 4);
                    const f32x4 u = acc[r][2 * q], v = acc[r][2 * q + 1];
                    u32x4s o;
                    o[0] = H16<T>::pack2(fmaxf(fmaf(u[0], a0.x, b0.x), 0.f), fmaxf(fmaf(u[1], a0.y, b0.y), 0.f));
                    o[1] = H16<T>::pack2(fmaxf(fmaf(u[2], a0.z, b0.z), 0.f), fmaxf(fmaf(u[3], a0.w, b0.w), 0.f));
                    o[2] = H16<T>::pack2(fmaxf(fmaf(v[0], a1.x, b1.x), 0.f), fmaxf(fmaf(v[1], a1.y, b1.y), 0.f));
                    o[3] = H16<T>::pack2(fmaxf(fmaf(v[2], a1.z, b1.z), 0.f), fmaxf(fmaf(v[3], a1.w, b1.w), 0.f));
                    if (!(FRONT_ABL & 1) || o[0] == 0x12345678u) __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, (int)off + 64 * q, 0, 0);
                }
            }
            { const int t0 = s0, t1 = s1; s0 = s2; s1 = s3; s2 = t0; s3 = t1; }
        }
    }
}

template <typename T>
int launch_front_t(const uint8_t* tiles, const unsigned long long* stats, const void* ws16, const float* s_scale, const float* s_bias,
                   const void* wc16, const float* c_scale, const float* c_bias, void* out, int n, int num_cus, hipStream_t s) {
    FrontParams<T> p;
    p.tiles = tiles; p.stats = stats;
    p.ws16 = reinterpret_cast<const uint4*>(ws16); p.s_scale = s_scale; p.s_bias = s_bias;
    p.wc16 = reinterpret_cast<const uint4*>(wc16); p.c_scale = c_scale; p.c_bias = c_bias;
    p.out = reinterpret_cast<T*>(out);
    p.n = n;
    p.nstrips = (CO + SW - 1) / SW;
    const long long base_items = (long long)n * p.nstrips;
    const int waves = num_cus * NWF;
    int nb = (CO + 24) / 25;
    if (base_items * nb < waves) {
        nb = (int)((waves + base_items - 1) / base_items);
        if (nb > CO / 4) nb = CO / 4;
        if (nb < 1) nb = 1;
    }
    p.nbands = nb;
    p.items = (int)(base_items * nb);
    constexpr size_t lds = 2 * 2 * 1024 + 9 * 4 * 1024 + 192 * 4 + (size_t)NWF * (3 * 112 + 4 * 16 * 96);
    static_assert(lds <= 160 * 1024, "front kernel LDS budget");
    auto kern = front_stream_kernel<T>;
    static BqLdsAttr attr;
    if (const int e = attr.ensure(reinterpret_cast<const void*>(kern), lds)) return e;
    int grid = (p.items + NWF - 1) / NWF;
    if (grid > num_cus) grid = num_cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NWF * 64), lds, s, p);
    return (int)hipGetLastError();
}

}  // namespace

// stats: [n][2] 64-bit integer sums of the tiles' bytes (launch_stage_stats); ws16: "block1_conv1/w16"; wc16: "block1_conv2/wp16"
int launch_front(int dtype, const uint8_t* tiles, const unsigned long long* stats, const void* ws16, const float* s_scale,
                 const float* s_bias, const void* wc16, const float* c_scale, const float* c_bias, void* out, int n, int num_cus,
                 hipStream_t s) {
    if (n <= 0) return 0;
    if (dtype == 0 || (long long)n * CO * CO * 64 * 2 > 0xffffff00ll) return (int)hipErrorInvalidValue;
    return dtype == 2 ? launch_front_t<f16_t>(tiles, stats, ws16, s_scale, s_bias, wc16, c_scale, c_bias, out, n, num_cus, s)
                      : launch_front_t<bf16_t>(tiles, stats, ws16, s_scale, s_bias, wc16, c_scale, c_bias, out, n, num_cus, s);
}
