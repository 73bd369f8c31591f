// WHO STILL RUNS THIS FILE (round 5).  The headline path (bq_mc_infer / bq_backbone_u8 in a 16-bit context) does not: its front is
// kernels_front.hip, block 2 and block3_sepconv1 are kernels_stream.hip.  run_conv (biscuit_hip.hip) comes here for
//   * block1_conv2 (kind 0) on the float / planar entry: bq_backbone (UncertaintyInterface: standardised float tiles),
//     bq_debug_activation, and a blob without "block1_conv1/w16" / "block1_conv2/wp16";
//   * block2_sepconv1 / block2_sepconv2 (kinds 1, 2) and block3_sepconv1 (kind 3) only when the streaming kernel refuses the launch:
//     a blob without "<layer>/wp16", or an output tensor beyond the 32-bit byte offsets of its raw buffer stores (n x H x W x C x 2
//     >= 4 GiB: batches of several thousand tiles).
// Kept bit-compatible with the streaming kernels (same tap order, same rounding points): tests/test_gpu_parity.py runs both entries.
//
// Persistent 2-D tile kernel for the big, HBM-bound entry-flow layers (block1_conv2 at 147x147,
// the block2/block3 separable convolutions at 147x147 / 74x74), bf16.
//
// These layers move 1-3 GB per batch through tiny GEMMs (K, N <= 256): what matters is that
// every input byte is fetched once, in coalesced 16-byte pieces, with enough loads in flight,
// and that nothing but the final activations is written.  Design:
//  * Persistent workgroups (4 waves) walk 8x16-pixel output tiles of the image grid.
//  * The layer's weights are copied ONCE per workgroup into LDS in MFMA fragment order
//    (16-64 KB) and stay there: no per-tile weight stream from L2.
//  * Per tile the (8+2)x(16+2) input halo is loaded with all of a thread's 16-byte loads in
//    flight at once, zero-filled outside the image ('same' padding for free), and the loads of
//    tile t+1 are issued before tile t is computed (register prefetch across the tile loop).
//  * No A tile: the MFMA operand layout D[cout][pixel] = W[cout][k] * Act[k][pixel] wants, per
//    lane, 8 consecutive k of ONE pixel -- exactly one depthwise result (8 channels of a
//    pixel) or, for the 3x3 stem conv, 16 bytes of a shifted halo pixel.  Each wave owns two
//    tile rows (32 pixels = one 32x32 fragment) and feeds its depthwise results straight into
//    its MFMAs: no LDS round trip, no barrier between the vector-ALU and matrix stages.
//  * Epilogue: folded BN + ReLU in registers, tile parked in LDS (aliasing the dead halo),
//    streamed out as 16 x Cout x 2 B contiguous row segments.
#include "gemm_common.h"

#include <stdlib.h>

namespace {
using namespace bqk;

constexpr int TH = 8, TW = 16, RH = TH + 2, RW = TW + 2, RPIX = RH * RW;   // 180 halo pixels
enum { MODE_CONV3 = 0, MODE_SEP = 1 };

typedef float f32x2t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned relu2(unsigned x) { return relu_pk16(x); }   // ReLU on two packed bf16 / f16

template <typename T>
struct TileParams {
    const T* in;           // NHWC [n][Hi][Wi][CIN]
    const uint4* wp;       // fragment-packed weights [NF][KB][64] x 16 B
    const float* dw;       // [9][CIN] fp32 (MODE_SEP)
    const float* scale;    // [NF*32]
    const float* bias;
    T* out;                // NHWC [n][H][W][NF*32]; tile_sep2p_kernel: [n][H][W][ldo]
    int n, H, W, Hi, Wi;   // output / input maps
    int tyn, txn;          // tiles per image
    int relu;
    int ldo;               // tile_sep2p_kernel: channels per output pixel in memory (a launch may write a slice of them).
                           // LAST on purpose: the 64 -> 128 instance of tile_sep2_kernel lost 16 % (0.48 -> 0.56 ms) to the
                           // schedule hipcc found when this field sat in the middle of its kernel arguments.
};

// WPE = waves per SIMD the register budget is set for (= persistent workgroups per CU): measured 0.68 -> 0.57 ms
// for the 64 -> 128 layer at 3 (its LDS footprint allows 3 workgroups); the 128 -> 128 layer's LDS allows 2.
template <typename T, int MODE, int CIN, int NF, bool RELU_IN, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
tile_conv_kernel(const TileParams<T> p) {
    if constexpr (H16<T>::F16) bq_f16_saturate();
    constexpr int NT = 256;
    constexpr int CC = CIN < 64 ? CIN : 64;            // channels staged per pass (<= 64)
    constexpr int NPASS = CIN / CC;
    constexpr int PPP = CC / 8;                        // 16-byte pieces per halo pixel and pass
    constexpr int PS = CC * 2 + 16;                    // halo pixel stride in LDS (odd # of 16-B slots)
    // Halo row pitch padded to a multiple of 16 slots: a ds_read_b128 lane group mixes pixels
    // 0-3/12-15 of one tile row with 4-11 of the next; with pitch = 0 (mod 16 slots) the second
    // row lands exactly on the slots the first leaves free (measured 24 % conflict cycles before)
    constexpr int RP = (RW * PS + 255) / 256 * 256;
    constexpr int KB = (MODE == MODE_CONV3 ? 9 * CIN : CIN) / 16;
    constexpr int KBP = KB / NPASS;                    // k-blocks per pass
    constexpr int NLOAD = (RPIX * PPP + NT - 1) / NT;  // raw 16-byte loads per thread and pass
    constexpr int N = NF * 32;
    constexpr int SST = N * 2 + 16;                    // staging row stride
    constexpr int W_BYTES = NF * KB * 1024;
    constexpr int TAP_BYTES = MODE == MODE_SEP ? 9 * CIN * 4 : 0;
    // folded-BN scale | bias as fp32 [2][N] in LDS -- except for the 64->128 instance, whose three workgroups
    // per CU leave no room for it (LDS is handed out in 1280-byte granules: +1 KB costs the third workgroup)
    constexpr bool SB_LDS = WPE < 3;
    constexpr int SB_OFF = W_BYTES + TAP_BYTES;
    constexpr int BUF_OFF = SB_OFF + (SB_LDS ? 2 * N * 4 : 0);   // raw halo / output staging share this region
    static_assert(MODE == MODE_SEP || NPASS == 1, "the 3x3 conv stages all its channels at once");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;

    // ---- one-time: weights (fragment order) and depthwise taps -> LDS
    for (int i = tid; i < W_BYTES / 16; i += NT)
        *reinterpret_cast<uint4*>(smem + i * 16) = p.wp[i];
    if (MODE == MODE_SEP) {
        for (int i = tid * 4; i < 9 * CIN; i += NT * 4) {
            const float4 wv = *reinterpret_cast<const float4*>(p.dw + i);
            // (w0,w2,w1,w3): the order the packed-FMA depthwise consumes them in
            *reinterpret_cast<float4*>(smem + W_BYTES + i * 4) = make_float4(wv.x, wv.z, wv.y, wv.w);
        }
    }

    if (SB_LDS) {
        for (int i = tid; i < N; i += NT) {
            reinterpret_cast<float*>(smem + SB_OFF)[i] = p.scale[i];
            reinterpret_cast<float*>(smem + SB_OFF)[N + i] = p.bias[i];
        }
    }

    const int tiles_per_img = p.tyn * p.txn;
    const int ntiles = p.n * tiles_per_img;
    const int org = MODE == MODE_SEP ? -1 : 0;         // halo origin relative to the tile origin

    // tile-independent part of this thread's halo pieces: (ry, rx) and the element offset
    int rel[NLOAD], ryx[NLOAD];
#pragma unroll
    for (int q = 0; q < NLOAD; ++q) {
        const int idx = tid + q * NT;
        const int pix = idx / PPP, j = idx - pix * PPP;
        const int ry = pix / RW, rx = pix - ry * RW;
        rel[q] = (ry * p.Wi + rx) * CIN + j * 8;
        ryx[q] = pix < RPIX ? ((ry << 8) | rx) : -1;
    }
    uint4 rreg[NLOAD];
    // issue the halo loads of (tile, pass) into registers; zeros outside the image
    auto load_pass = [&](int tile, int pass) {
        const int img = tile / tiles_per_img;
        const int trem = tile - img * tiles_per_img;
        const int ty = trem / p.txn, tx = trem - ty * p.txn;
        const int gy0 = ty * TH + org, gx0 = tx * TW + org;
        const long long base = ((long long)(img * p.Hi + gy0) * p.Wi + gx0) * CIN + pass * CC;
#pragma unroll
        for (int q = 0; q < NLOAD; ++q) {
            const int gy = gy0 + (ryx[q] >> 8), gx = gx0 + (ryx[q] & 255);
            const bool ok = ryx[q] >= 0 && (unsigned)gy < (unsigned)p.Hi && (unsigned)gx < (unsigned)p.Wi;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) v = *reinterpret_cast<const uint4*>(p.in + base + rel[q]);
            if (RELU_IN) { v.x = relu2(v.x); v.y = relu2(v.y); v.z = relu2(v.z); v.w = relu2(v.w); }
            rreg[q] = v;
        }
    };
    auto store_pass = [&]() {
#pragma unroll
        for (int q = 0; q < NLOAD; ++q) {
            const int idx = tid + q * NT;
            const int pix = idx / PPP, j = idx - pix * PPP;
            if (ryx[q] >= 0)
                *reinterpret_cast<uint4*>(smem + BUF_OFF + (ryx[q] >> 8) * RP + (ryx[q] & 255) * PS + j * 16) = rreg[q];
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) load_pass(tile, 0);
    // this wave's pixels: tile rows 2*wave and 2*wave+1, lane&31 -> (row, column)
    const int py = 2 * wave + (r32 >> 4), px = r32 & 15;
    const int raw_lane = BUF_OFF + py * RP + px * PS;          // halo pixel of tap (0,0)

    for (; tile < ntiles; tile += gridDim.x) {
        f32x16 acc[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

#pragma unroll 1
        for (int pass = 0; pass < NPASS; ++pass) {
            __syncthreads();             // previous readers of the halo / staging region are done
            store_pass();
            __syncthreads();             // halo (and, first time, weights/taps) visible
            // next halo in flight while this one is computed
            if (pass + 1 < NPASS) load_pass(tile, pass + 1);
            else if (tile + (int)gridDim.x < ntiles) load_pass(tile + gridDim.x, 0);

#pragma unroll 2
            for (int kl = 0; kl < KBP; ++kl) {
                const int kb = pass * KBP + kl;
                uint4 opnd;
                if constexpr (MODE == MODE_CONV3) {
                    // k = tap*CIN + channel: this k-block is 16 channels of one tap
                    const int tap = (kb * 16) / CIN, c0 = (kb * 16) % CIN;
                    const int dy = tap / 3, dx = tap - dy * 3;
                    opnd = *reinterpret_cast<const uint4*>(smem + raw_lane + dy * RP + dx * PS + (c0 + h * 8) * 2);
                } else {
                    // depthwise 3x3 of 8 channels (piece 2*kl + h of this pass) of this lane's pixel
                    const int wbase = W_BYTES + (2 * kb + h) * 32;
                    const int rbase = raw_lane + (2 * kl + h) * 16;
                    f32x2t aA = {0.f, 0.f}, aB = {0.f, 0.f}, aC = {0.f, 0.f}, aD = {0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const uint4 v = *reinterpret_cast<const uint4*>(smem + rbase + (t / 3) * RP + (t % 3) * PS);
                        const float4 w0 = *reinterpret_cast<const float4*>(smem + wbase + t * CIN * 4);
                        const float4 w1 = *reinterpret_cast<const float4*>(smem + wbase + t * CIN * 4 + 16);
                        typedef H16<T> F;
                        const f32x2t lo01 = {F::lo(v.x), F::lo(v.y)};
                        const f32x2t hi01 = {F::hi(v.x), F::hi(v.y)};
                        const f32x2t lo23 = {F::lo(v.z), F::lo(v.w)};
                        const f32x2t hi23 = {F::hi(v.z), F::hi(v.w)};
                        aA = __builtin_elementwise_fma((f32x2t){w0.x, w0.y}, lo01, aA);
                        aB = __builtin_elementwise_fma((f32x2t){w0.z, w0.w}, hi01, aB);
                        aC = __builtin_elementwise_fma((f32x2t){w1.x, w1.y}, lo23, aC);
                        aD = __builtin_elementwise_fma((f32x2t){w1.z, w1.w}, hi23, aD);
                    }
                    const float a8[8] = {aA.x, aB.x, aA.y, aB.y, aC.x, aD.x, aC.y, aD.y};
                    opnd = pack<T>(a8);
                }
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const uint4 wf = *reinterpret_cast<const uint4*>(smem + ((j * KB + kb) * 64 + lane) * 16);
                    mma<T>(acc[j], wf, opnd);
                }
            }
        }

        // ---- epilogue: BN + ReLU in registers -> LDS staging (aliases the halo) -> row segments.
        // Scale and bias come from LDS (per tile they would be 32 KB of L1 traffic per wave, more than the
        // tile's own pixels); ReLU is a packed int16 max on the converted pair (0x8000 = no-op).
        __syncthreads();                 // every wave is done reading the halo
        {
            const unsigned lo2 = p.relu ? 0u : 0x80008000u;
            const float* sbl = (SB_LDS ? reinterpret_cast<const float*>(smem + SB_OFF) : p.scale) + h * 4;
            const float* bbl = (SB_LDS ? reinterpret_cast<const float*>(smem + SB_OFF) + N : p.bias) + h * 4;
            unsigned char* row = smem + BUF_OFF + (wave * 32 + r32) * SST + h * 8;
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ng = j * 32 + g * 8;
                    const float4 sc = *reinterpret_cast<const float4*>(sbl + ng);
                    const float4 bi = *reinterpret_cast<const float4*>(bbl + ng);
                    const float v0 = fmaf(acc[j][g * 4 + 0], sc.x, bi.x), v1 = fmaf(acc[j][g * 4 + 1], sc.y, bi.y);
                    const float v2 = fmaf(acc[j][g * 4 + 2], sc.z, bi.z), v3 = fmaf(acc[j][g * 4 + 3], sc.w, bi.w);
                    uint2 o;
                    o.x = H16<T>::pack2(v0, v1);
                    o.y = H16<T>::pack2(v2, v3);
                    asm("v_pk_max_i16 %0, %1, %2" : "=v"(o.x) : "v"(o.x), "v"(lo2));
                    asm("v_pk_max_i16 %0, %1, %2" : "=v"(o.y) : "v"(o.y), "v"(lo2));
                    *reinterpret_cast<uint2*>(row + ng * 2) = o;
                }
        }
        __syncthreads();
        {
            const int img = tile / tiles_per_img;
            const int trem = tile - img * tiles_per_img;
            const int ty = trem / p.txn, tx = trem - ty * p.txn;
            constexpr int PPR = N * 2 / 16;            // 16-byte pieces per output pixel
            for (int idx = tid; idx < TH * TW * PPR; idx += NT) {
                const int pix = idx / PPR, pc = idx - pix * PPR;
                const int oy = ty * TH + (pix >> 4), ox = tx * TW + (pix & 15);
                if (oy < p.H && ox < p.W)
                    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.out) +
                                              (((size_t)(img * p.H + oy) * p.W + ox) * N) * 2 + pc * 16) =
                        *reinterpret_cast<const uint4*>(smem + BUF_OFF + pix * SST + pc * 16);
            }
        }
    }
}

template <typename T, int MODE, int CIN, int NF, bool RELU_IN, int WPE>
int launch_tile(const TileParams<T>& p, int num_cus, hipStream_t s) {
    constexpr int KB = (MODE == MODE_CONV3 ? 9 * CIN : CIN) / 16;
    constexpr size_t W_BYTES = (size_t)NF * KB * 1024;
    constexpr size_t TAP_BYTES = MODE == MODE_SEP ? 9 * CIN * 4 : 0;
    constexpr size_t RAW_BYTES = (size_t)RH * ((RW * ((CIN < 64 ? CIN : 64) * 2 + 16) + 255) / 256 * 256);
    constexpr size_t STAGE_BYTES = (size_t)TH * TW * (NF * 64 + 16);
    constexpr size_t lds = W_BYTES + TAP_BYTES + (WPE < 3 ? (size_t)NF * 32 * 8 : 0) + (RAW_BYTES > STAGE_BYTES ? RAW_BYTES : STAGE_BYTES);
    static_assert(lds <= 160 * 1024, "tile kernel LDS budget");
    auto kern = tile_conv_kernel<T, MODE, CIN, NF, RELU_IN, WPE>;
    static BqLdsAttr attr;
    if (const int e = attr.ensure(reinterpret_cast<const void*>(kern), lds)) return e;
    const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
    static const int env_wgs = bq_exp_env("BQ_TILE_WGS") ? atoi(bq_exp_env("BQ_TILE_WGS")) : 0;
    int wgs = per_cu > WPE ? WPE : per_cu;     // persistent workgroups per CU = waves per SIMD
    if (env_wgs > 0 && env_wgs < wgs) wgs = env_wgs;
    const int ntiles = p.n * p.tyn * p.txn;
    int grid = num_cus * wgs;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, p);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Second form of the separable tile kernel: depthwise with lane = CHANNEL PAIR and a sliding 3x3 window.
//
// In the kernel above a lane owns one pixel and eight channels, so for every tap it reads 16 B of halo and 32 B
// of fp32 taps from LDS: 432 B per pixel and 8 channels, 6.9 KB per pixel at 128 channels -- the kernel is bound
// by LDS bandwidth, two thirds of it tap vectors that are the same for every pixel.  Here a half-wave owns one
// row of the tile and its 32 lanes the 64 channels of a pass, two each: the nine tap pairs of a lane live in
// registers for the whole pass, and walking along the row only the three halo values of the new column are read
// (4 B each) -- 0.77 KB of LDS per pixel at 128 channels, and 15 instead of 27 vector instructions per pixel and
// channel pair.  The depthwise results go to a wave-private A tile in LDS (32 pixels x 64 channels, read back as
// MFMA fragments by the same wave: no workgroup barrier between the stages), the matrix stage and the staged
// epilogue are the ones above.  8 waves, 16 x 16-pixel tiles (halo 18 x 18), one workgroup per CU.
constexpr int T2 = 16, R2 = T2 + 2;                    // tile edge, halo edge

template <typename T, int CIN, int NF, bool RELU_IN>
__global__ void __launch_bounds__(512) tile_sep2_kernel(const TileParams<T> p) {
    if constexpr (H16<T>::F16) bq_f16_saturate();
    constexpr int NT = 512;
    constexpr int CC = 64;                             // channels per pass
    constexpr int NPASS = CIN / CC;
    constexpr int PPP = CC / 8;                        // 16-byte pieces per halo pixel and pass
    constexpr int PS = CC * 2;                         // halo pixel stride: 128 B, a half-wave's 32 dwords
    constexpr int RP = R2 * PS + 128;                  // row pitch = 128 (mod 256): the two half-waves of a read
                                                       // (rows y, y+1) land in disjoint banks
    constexpr int KB = CIN / 16, KBP = CC / 16;
    constexpr int N = NF * 32;
    constexpr int SST = N * 2 + 16;                    // staging row stride
    constexpr int AST = CC * 2 + 16;                   // A row stride (odd number of 16-byte slots)
    constexpr int W_BYTES = NF * KB * 1024;
    constexpr int TAP_OFF = W_BYTES;                   // fp32 [9][CIN]
    constexpr int SB_OFF = TAP_OFF + 9 * CIN * 4;      // scale | bias fp32 [2][N]
    constexpr int RAW_OFF = SB_OFF + 2 * N * 4;
    constexpr int PRIV_OFF = RAW_OFF + R2 * RP;        // per wave: A tile (32 x AST), later its 32 staging rows
    constexpr int PRIV = 32 * SST;
    constexpr int RPIX2 = R2 * R2;
    constexpr int NLOAD = (RPIX2 * PPP + NT - 1) / NT;
    static_assert(32 * AST <= PRIV, "A tile must fit the wave's staging rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;

    for (int i = tid; i < W_BYTES / 16; i += NT) *reinterpret_cast<uint4*>(smem + i * 16) = p.wp[i];
    for (int i = tid; i < 9 * CIN; i += NT) reinterpret_cast<float*>(smem + TAP_OFF)[i] = p.dw[i];
    for (int i = tid; i < N; i += NT) {
        reinterpret_cast<float*>(smem + SB_OFF)[i] = p.scale[i];
        reinterpret_cast<float*>(smem + SB_OFF)[N + i] = p.bias[i];
    }

    const int tiles_per_img = p.tyn * p.txn;
    const int ntiles = p.n * tiles_per_img;

    int rel[NLOAD], ryx[NLOAD];
#pragma unroll
    for (int q = 0; q < NLOAD; ++q) {
        const int idx = tid + q * NT;
        const int pix = idx / PPP, j = idx - pix * PPP;
        const int ry = pix / R2, rx = pix - ry * R2;
        rel[q] = (ry * p.Wi + rx) * CIN + j * 8;
        ryx[q] = pix < RPIX2 ? ((ry << 8) | rx) : -1;
    }
    uint4 rreg[NLOAD];
    auto load_pass = [&](int tile, int pass) {
        const int img = tile / tiles_per_img;
        const int trem = tile - img * tiles_per_img;
        const int ty = trem / p.txn, tx = trem - ty * p.txn;
        const int gy0 = ty * T2 - 1, gx0 = tx * T2 - 1;
        const long long base = ((long long)(img * p.Hi + gy0) * p.Wi + gx0) * CIN + pass * CC;
#pragma unroll
        for (int q = 0; q < NLOAD; ++q) {
            const int gy = gy0 + (ryx[q] >> 8), gx = gx0 + (ryx[q] & 255);
            const bool ok = ryx[q] >= 0 && (unsigned)gy < (unsigned)p.Hi && (unsigned)gx < (unsigned)p.Wi;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) v = *reinterpret_cast<const uint4*>(p.in + base + rel[q]);
            if (RELU_IN) { v.x = relu2(v.x); v.y = relu2(v.y); v.z = relu2(v.z); v.w = relu2(v.w); }
            rreg[q] = v;
        }
    };
    auto store_pass = [&]() {
#pragma unroll
        for (int q = 0; q < NLOAD; ++q) {
            const int idx = tid + q * NT;
            const int pix = idx / PPP, j = idx - pix * PPP;
            if (ryx[q] >= 0)
                *reinterpret_cast<uint4*>(smem + RAW_OFF + (ryx[q] >> 8) * RP + (ryx[q] & 255) * PS + j * 16) = rreg[q];
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) load_pass(tile, 0);
    // depthwise role of this lane: tile row 2*wave + h, channel pair r32 of the pass; halo (row, col) of output
    // pixel (y, x) and tap (dy, dx) is (y + dy, x + dx)
    const int d_row = 2 * wave + h;
    const int raw_lane = RAW_OFF + d_row * RP + r32 * 4;
    unsigned char* priv = smem + PRIV_OFF + wave * PRIV;
    const int a_write = (16 * h) * AST + r32 * 4;       // A row of output pixel (d_row, x): fragment pixel 16*h + x
    const int a_read = r32 * AST + h * 16;

    for (; tile < ntiles; tile += gridDim.x) {
        f32x16 acc[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

#pragma unroll 1
        for (int pass = 0; pass < NPASS; ++pass) {
            __syncthreads();             // halo readers and the previous tile's store pass are done
            store_pass();
            __syncthreads();
            if (pass + 1 < NPASS) load_pass(tile, pass + 1);
            else if (tile + (int)gridDim.x < ntiles) load_pass(tile + gridDim.x, 0);

            // ---- depthwise: nine tap pairs in registers, sliding window along the row
            f32x2t tap[9];
#pragma unroll
            for (int t = 0; t < 9; ++t)
                tap[t] = *reinterpret_cast<const f32x2t*>(smem + TAP_OFF + (t * CIN + pass * CC + 2 * r32) * 4);
            f32x2t win[3][3];            // [dy][column slot]: fp32 pairs of halo columns x, x+1, x+2
            auto fetch = [&](int col, f32x2t (&dst)[3], int slot_unused) {
                (void)slot_unused;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const unsigned u = *reinterpret_cast<const unsigned*>(smem + raw_lane + dy * RP + col * PS);
                    dst[dy] = (f32x2t){H16<T>::lo(u), H16<T>::hi(u)};
                }
            };
            f32x2t c0[3], c1[3], c2[3];
            fetch(0, c0, 0);
            fetch(1, c1, 0);
#pragma unroll
            for (int x = 0; x < T2; ++x) {
                fetch(x + 2, c2, 0);
                f32x2t a = {0.f, 0.f};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    a = __builtin_elementwise_fma(tap[dy * 3 + 0], c0[dy], a);
                    a = __builtin_elementwise_fma(tap[dy * 3 + 1], c1[dy], a);
                    a = __builtin_elementwise_fma(tap[dy * 3 + 2], c2[dy], a);
                }
                const unsigned o = H16<T>::pack2(a.x, a.y);
                *reinterpret_cast<unsigned*>(priv + a_write + x * AST) = o;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) { c0[dy] = c1[dy]; c1[dy] = c2[dy]; }
            }
            (void)win;
            // ---- matrix stage on the wave's own A tile (LDS operations of a wave complete in order)
#pragma unroll
            for (int kl = 0; kl < KBP; ++kl) {
                const int kb = pass * KBP + kl;
                const uint4 opnd = *reinterpret_cast<const uint4*>(priv + a_read + kl * 32);
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const uint4 wf = *reinterpret_cast<const uint4*>(smem + ((j * KB + kb) * 64 + lane) * 16);
                    mma<T>(acc[j], wf, opnd);
                }
            }
        }

        // ---- epilogue: BN + ReLU -> this wave's 32 staging rows (its A tile is dead) -> row segments
        {
            const unsigned lo2 = p.relu ? 0u : 0x80008000u;
            const float* sbl = reinterpret_cast<const float*>(smem + SB_OFF) + h * 4;
            unsigned char* row = priv + r32 * SST + h * 8;
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ng = j * 32 + g * 8;
                    const float4 sc = *reinterpret_cast<const float4*>(sbl + ng);
                    const float4 bi = *reinterpret_cast<const float4*>(sbl + N + ng);
                    const float v0 = fmaf(acc[j][g * 4 + 0], sc.x, bi.x), v1 = fmaf(acc[j][g * 4 + 1], sc.y, bi.y);
                    const float v2 = fmaf(acc[j][g * 4 + 2], sc.z, bi.z), v3 = fmaf(acc[j][g * 4 + 3], sc.w, bi.w);
                    uint2 o;
                    o.x = H16<T>::pack2(v0, v1);
                    o.y = H16<T>::pack2(v2, v3);
                    asm("v_pk_max_i16 %0, %1, %2" : "=v"(o.x) : "v"(o.x), "v"(lo2));
                    asm("v_pk_max_i16 %0, %1, %2" : "=v"(o.y) : "v"(o.y), "v"(lo2));
                    *reinterpret_cast<uint2*>(row + ng * 2) = o;
                }
        }
        {
            // each wave streams out its own 32 pixels (two tile rows x 16 columns x N channels: 4 pixels = 1 KB
            // contiguous per instruction); no workgroup barrier, the rows are wave-private
            const int img = tile / tiles_per_img;
            const int trem = tile - img * tiles_per_img;
            const int ty = trem / p.txn, tx = trem - ty * p.txn;
            constexpr int PPR = N * 2 / 16;
#pragma unroll
            for (int it = 0; it < 32 * PPR / 64; ++it) {
                const int idx = it * 64 + lane;
                const int pix = idx / PPR, pc = idx - pix * PPR;
                const int oy = ty * T2 + 2 * wave + (pix >> 4), ox = tx * T2 + (pix & 15);
                if (oy < p.H && ox < p.W)
                    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.out) +
                                              (((size_t)(img * p.H + oy) * p.W + ox) * N) * 2 + pc * 16) =
                        *reinterpret_cast<const uint4*>(priv + pix * SST + pc * 16);
            }
        }
    }
}

// Two instances of the same design: `tile_sep2_kernel` (one halo register set, the next pass requested after the
// current one is stored) serves the single-pass 64-channel layer; `tile_sep2p_kernel` keeps one register set per
// pass and requests (tile + grid, pass) right after (tile, pass) went to LDS -- a whole tile's work ahead of its
// use -- with branch-free loads: 128->128 0.75 -> 0.70 ms.  On the 64-channel layer the second form measured
// 0.43 -> 0.55 ms (same instruction counts, a worse schedule), so it keeps the first.
template <typename T, int CIN, int NF, bool RELU_IN>
__global__ void __launch_bounds__(512) tile_sep2p_kernel(const TileParams<T> p) {
    if constexpr (H16<T>::F16) bq_f16_saturate();
    constexpr int NT = 512;
    constexpr int CC = 64;                             // channels per pass
    constexpr int NPASS = CIN / CC;
    constexpr int PPP = CC / 8;                        // 16-byte pieces per halo pixel and pass
    constexpr int PS = CC * 2;                         // halo pixel stride: 128 B, a half-wave's 32 dwords
    constexpr int RP = R2 * PS + 128;                  // row pitch = 128 (mod 256): the two half-waves of a read
                                                       // (rows y, y+1) land in disjoint banks
    constexpr int KB = CIN / 16, KBP = CC / 16;
    constexpr int N = NF * 32;
    constexpr int SST = N * 2 + 16;                    // staging row stride
    constexpr int AST = CC * 2 + 16;                   // A row stride (odd number of 16-byte slots)
    constexpr int W_BYTES = NF * KB * 1024;
    constexpr int TAP_OFF = W_BYTES;                   // fp32 [9][CIN]
    constexpr int SB_OFF = TAP_OFF + 9 * CIN * 4;      // scale | bias fp32 [2][N]
    constexpr int RAW_OFF = SB_OFF + 2 * N * 4;
    constexpr int PRIV_OFF = RAW_OFF + R2 * RP;        // per wave: A tile (32 x AST), later its 32 staging rows
    constexpr int PRIV = 32 * SST;
    constexpr int RPIX2 = R2 * R2;
    constexpr int NLOAD = (RPIX2 * PPP + NT - 1) / NT;
    static_assert(32 * AST <= PRIV, "A tile must fit the wave's staging rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;

    for (int i = tid; i < W_BYTES / 16; i += NT) *reinterpret_cast<uint4*>(smem + i * 16) = p.wp[i];
    for (int i = tid; i < 9 * CIN; i += NT) reinterpret_cast<float*>(smem + TAP_OFF)[i] = p.dw[i];
    for (int i = tid; i < N; i += NT) {
        reinterpret_cast<float*>(smem + SB_OFF)[i] = p.scale[i];
        reinterpret_cast<float*>(smem + SB_OFF)[N + i] = p.bias[i];
    }

    const int tiles_per_img = p.tyn * p.txn;
    const int ntiles = p.n * tiles_per_img;

    int rel[NLOAD], ryx[NLOAD];
#pragma unroll
    for (int q = 0; q < NLOAD; ++q) {
        const int idx = tid + q * NT;
        const int pix = idx / PPP, j = idx - pix * PPP;
        const int ry = pix / R2, rx = pix - ry * R2;
        rel[q] = (ry * p.Wi + rx) * CIN + j * 8;
        ryx[q] = pix < RPIX2 ? ((ry << 8) | rx) : -1;
    }
    // one register set per pass: the halo of (tile + grid, pass) is requested right after (tile, pass) went to
    // LDS, a whole tile's work ahead of its use
    uint4 rreg[NPASS][NLOAD];
    unsigned okbits[NPASS];                            // bit q: piece q of the set lies inside the image
    // Branch-free: a piece outside the image loads the tensor's first bytes (valid, unused) and is zeroed when it
    // is stored -- with the load under `if (ok)` the compiler put a vmcnt(0) in the middle of the sequence.
    auto load_pass = [&](int tile, int pass, uint4 (&dst)[NLOAD], unsigned& bits) {
        const int img = tile / tiles_per_img;
        const int trem = tile - img * tiles_per_img;
        const int ty = trem / p.txn, tx = trem - ty * p.txn;
        const int gy0 = ty * T2 - 1, gx0 = tx * T2 - 1;
        const long long base = ((long long)(img * p.Hi + gy0) * p.Wi + gx0) * CIN + pass * CC;
        bits = 0;
#pragma unroll
        for (int q = 0; q < NLOAD; ++q) {
            const int gy = gy0 + (ryx[q] >> 8), gx = gx0 + (ryx[q] & 255);
            const bool ok = ryx[q] >= 0 && (unsigned)gy < (unsigned)p.Hi && (unsigned)gx < (unsigned)p.Wi;
            dst[q] = *reinterpret_cast<const uint4*>(p.in + (ok ? base + rel[q] : 0));
            bits |= ok ? (1u << q) : 0u;
        }
    };
    auto store_pass = [&](const uint4 (&src)[NLOAD], unsigned bits) {
#pragma unroll
        for (int q = 0; q < NLOAD; ++q) {
            const int idx = tid + q * NT;
            const int pix = idx / PPP, j = idx - pix * PPP;
            uint4 v = src[q];
            if (!((bits >> q) & 1u)) v = make_uint4(0, 0, 0, 0);
            if (RELU_IN) { v.x = relu2(v.x); v.y = relu2(v.y); v.z = relu2(v.z); v.w = relu2(v.w); }
            if (ryx[q] >= 0)
                *reinterpret_cast<uint4*>(smem + RAW_OFF + (ryx[q] >> 8) * RP + (ryx[q] & 255) * PS + j * 16) = v;
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) {
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) load_pass(tile, pass, rreg[pass], okbits[pass]);
    }
    // depthwise role of this lane: tile row 2*wave + h, channel pair r32 of the pass; halo (row, col) of output
    // pixel (y, x) and tap (dy, dx) is (y + dy, x + dx)
    const int d_row = 2 * wave + h;
    const int raw_lane = RAW_OFF + d_row * RP + r32 * 4;
    unsigned char* priv = smem + PRIV_OFF + wave * PRIV;
    const int a_write = (16 * h) * AST + r32 * 4;       // A row of output pixel (d_row, x): fragment pixel 16*h + x
    const int a_read = r32 * AST + h * 16;

    for (; tile < ntiles; tile += gridDim.x) {
        f32x16 acc[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            __syncthreads();             // halo readers and the previous tile's store pass are done
            store_pass(rreg[pass], okbits[pass]);
            __syncthreads();
            if (tile + (int)gridDim.x < ntiles) load_pass(tile + gridDim.x, pass, rreg[pass], okbits[pass]);

            // ---- depthwise: nine tap pairs in registers, sliding window along the row
            f32x2t tap[9];
#pragma unroll
            for (int t = 0; t < 9; ++t)
                tap[t] = *reinterpret_cast<const f32x2t*>(smem + TAP_OFF + (t * CIN + pass * CC + 2 * r32) * 4);
            f32x2t win[3][3];            // [dy][column slot]: fp32 pairs of halo columns x, x+1, x+2
            auto fetch = [&](int col, f32x2t (&dst)[3], int slot_unused) {
                (void)slot_unused;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const unsigned u = *reinterpret_cast<const unsigned*>(smem + raw_lane + dy * RP + col * PS);
                    dst[dy] = (f32x2t){H16<T>::lo(u), H16<T>::hi(u)};
                }
            };
            f32x2t c0[3], c1[3], c2[3];
            fetch(0, c0, 0);
            fetch(1, c1, 0);
#pragma unroll
            for (int x = 0; x < T2; ++x) {
                fetch(x + 2, c2, 0);
                f32x2t a = {0.f, 0.f};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    a = __builtin_elementwise_fma(tap[dy * 3 + 0], c0[dy], a);
                    a = __builtin_elementwise_fma(tap[dy * 3 + 1], c1[dy], a);
                    a = __builtin_elementwise_fma(tap[dy * 3 + 2], c2[dy], a);
                }
                const unsigned o = H16<T>::pack2(a.x, a.y);
                *reinterpret_cast<unsigned*>(priv + a_write + x * AST) = o;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) { c0[dy] = c1[dy]; c1[dy] = c2[dy]; }
            }
            (void)win;
            // ---- matrix stage on the wave's own A tile (LDS operations of a wave complete in order)
#pragma unroll
            for (int kl = 0; kl < KBP; ++kl) {
                const int kb = pass * KBP + kl;
                const uint4 opnd = *reinterpret_cast<const uint4*>(priv + a_read + kl * 32);
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const uint4 wf = *reinterpret_cast<const uint4*>(smem + ((j * KB + kb) * 64 + lane) * 16);
                    mma<T>(acc[j], wf, opnd);
                }
            }
        }

        // ---- epilogue: BN + ReLU -> this wave's 32 staging rows (its A tile is dead) -> row segments
        {
            const unsigned lo2 = p.relu ? 0u : 0x80008000u;
            const float* sbl = reinterpret_cast<const float*>(smem + SB_OFF) + h * 4;
            unsigned char* row = priv + r32 * SST + h * 8;
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ng = j * 32 + g * 8;
                    const float4 sc = *reinterpret_cast<const float4*>(sbl + ng);
                    const float4 bi = *reinterpret_cast<const float4*>(sbl + N + ng);
                    const float v0 = fmaf(acc[j][g * 4 + 0], sc.x, bi.x), v1 = fmaf(acc[j][g * 4 + 1], sc.y, bi.y);
                    const float v2 = fmaf(acc[j][g * 4 + 2], sc.z, bi.z), v3 = fmaf(acc[j][g * 4 + 3], sc.w, bi.w);
                    uint2 o;
                    o.x = H16<T>::pack2(v0, v1);
                    o.y = H16<T>::pack2(v2, v3);
                    asm("v_pk_max_i16 %0, %1, %2" : "=v"(o.x) : "v"(o.x), "v"(lo2));
                    asm("v_pk_max_i16 %0, %1, %2" : "=v"(o.y) : "v"(o.y), "v"(lo2));
                    *reinterpret_cast<uint2*>(row + ng * 2) = o;
                }
        }
        {
            // each wave streams out its own 32 pixels (two tile rows x 16 columns x N channels: 4 pixels = 1 KB
            // contiguous per instruction); no workgroup barrier, the rows are wave-private
            const int img = tile / tiles_per_img;
            const int trem = tile - img * tiles_per_img;
            const int ty = trem / p.txn, tx = trem - ty * p.txn;
            constexpr int PPR = N * 2 / 16;
#pragma unroll
            for (int it = 0; it < 32 * PPR / 64; ++it) {
                const int idx = it * 64 + lane;
                const int pix = idx / PPR, pc = idx - pix * PPR;
                const int oy = ty * T2 + 2 * wave + (pix >> 4), ox = tx * T2 + (pix & 15);
                if (oy < p.H && ox < p.W)
                    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.out) +
                                              (((size_t)(img * p.H + oy) * p.W + ox) * p.ldo) * 2 + pc * 16) =
                        *reinterpret_cast<const uint4*>(priv + pix * SST + pc * 16);
            }
        }
    }
}

template <typename T, int CIN, int NF, bool RELU_IN>
int launch_tile_sep2(TileParams<T> p, int num_cus, hipStream_t s) {
    constexpr int KB = CIN / 16, N = NF * 32;
    constexpr size_t lds = (size_t)NF * KB * 1024 + 9 * CIN * 4 + 2 * N * 4 + (size_t)R2 * (R2 * 128 + 128) +
                           (size_t)8 * 32 * (N * 2 + 16);
    static_assert(lds <= 160 * 1024, "tile kernel LDS budget");
    auto kern = CIN > 64 ? tile_sep2p_kernel<T, CIN, NF, RELU_IN> : tile_sep2_kernel<T, CIN, NF, RELU_IN>;
    static BqLdsAttr attr;
    if (const int e = attr.ensure(reinterpret_cast<const void*>(kern), lds)) return e;
    p.tyn = (p.H + T2 - 1) / T2;
    p.txn = (p.W + T2 - 1) / T2;
    const int ntiles = p.n * p.tyn * p.txn;
    int grid = num_cus;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, p);
    return (int)hipGetLastError();
}

// kind: 0 = 3x3 valid conv 32->64 (block1_conv2); 1 = sepconv 64->128; 2 = sepconv 128->128;
// 3 = sepconv 128->256 with ReLU on the input.  Returns <0 if the combination is not built.
template <typename T>
int launch_tile_conv_t(int kind, const void* in, const void* wp, const float* dw, const float* scale,
                       const float* bias, void* out, int n, int H, int W, int Hi, int Wi, int relu, int num_cus,
                       hipStream_t s) {
    TileParams<T> p;
    p.in = reinterpret_cast<const T*>(in);
    p.wp = reinterpret_cast<const uint4*>(wp);
    p.dw = dw; p.scale = scale; p.bias = bias;
    p.out = reinterpret_cast<T*>(out);
    p.ldo = 0;                                         // (only the split launches of kind 3 set it)
    p.n = n; p.H = H; p.W = W; p.Hi = Hi; p.Wi = Wi;
    p.tyn = (H + TH - 1) / TH; p.txn = (W + TW - 1) / TW;
    p.relu = relu;
    static const bool sep2 = bq_exp_env("BQ_TILE_SEP1") == nullptr;   // the lane = channel-pair form (default)
    if (sep2 && kind == 1) return launch_tile_sep2<T, 64, 4, false>(p, num_cus, s);
    if (sep2 && kind == 2) { p.ldo = 128; return launch_tile_sep2<T, 128, 4, false>(p, num_cus, s); }
    if (sep2 && kind == 3) {
        // 128 -> 256 (block3_sepconv1, 74x74): the 128 -> 128 kernel twice, each launch its half of the output channels
        // (weights, scale and bias of a half are contiguous; the pixel rows of the output are 256 channels apart).
        // The depthwise stage and the input read are done twice -- 0.58 ms on the pipelined kernel against 2 x 0.2 ms.
        p.ldo = 256;
        for (int half = 0; half < 2; ++half) {
            TileParams<T> q = p;
            q.wp = p.wp + (size_t)half * 4 * (128 / 16) * 64;
            q.scale = p.scale + half * 128; q.bias = p.bias + half * 128;
            q.out = p.out + half * 128;
            if (const int e = launch_tile_sep2<T, 128, 4, true>(q, num_cus, s)) return e;
        }
        return 0;
    }
    switch (kind) {
        case 0: return launch_tile<T, MODE_CONV3, 32, 2, false, 2>(p, num_cus, s);
        case 1: return launch_tile<T, MODE_SEP, 64, 4, false, 3>(p, num_cus, s);
        case 2: return launch_tile<T, MODE_SEP, 128, 4, false, 2>(p, num_cus, s);
        case 3: return launch_tile<T, MODE_SEP, 128, 8, true, 1>(p, num_cus, s);
    }
    return -1;
}

}  // namespace

int launch_tile_conv(int dtype, int kind, const void* in, const void* wp, const float* dw, const float* scale,
                     const float* bias, void* out, int n, int H, int W, int Hi, int Wi, int relu, int num_cus,
                     hipStream_t s) {
    return dtype == 2 ? launch_tile_conv_t<f16_t>(kind, in, wp, dw, scale, bias, out, n, H, W, Hi, Wi, relu, num_cus, s)
                      : launch_tile_conv_t<bf16_t>(kind, in, wp, dw, scale, bias, out, n, H, W, Hi, Wi, relu, num_cus, s);
}
