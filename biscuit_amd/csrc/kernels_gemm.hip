// Fused "producer -> LDS A tile -> MFMA" kernel: the workhorse of the Xception backbone
// (SeparableConv2D = depthwise 3x3 + pointwise 1x1 + folded BN [+ residual] [+ ReLU],
// residual 1x1/s2 convs, block1_conv2 as im2col).  (The MC-dropout Dense layers have their own kernel, kernels_head.hip.)
//
// Design (gfx950 / CDNA4, wave64):
//  * One workgroup owns MT = 32*MF output pixels and ALL output channels.  Phase 1
//    builds the [MT][K] A operand in LDS once (depthwise 3x3 on the vector ALU, a
//    stride-2 gather or an im2col gather), so the depthwise result never
//    touches HBM and is not recomputed per output-channel tile.
//  * Phase 2: every wave owns RN 32-wide output-channel fragments per pass and all RM
//    row fragments; weights were pre-swizzled on the host into MFMA fragment order, so a
//    B fragment is ONE coalesced 1 KiB global load straight into VGPRs (L2-resident
//    weights, no LDS hop), prefetched PF k-blocks ahead through a register ring.
//  * MFMA orientation D[cout][pixel] = W[cout][k] * A[k][pixel]: each lane then holds
//    4 consecutive output channels of one pixel per accumulator quad, i.e. 8-byte (bf16)
//    or 16-byte (fp32) NHWC stores instead of 2-byte ones.
//  * LDS row stride is an odd number of 16-byte slots: ds_read_b128 of 16 distinct rows
//    is bank-conflict free (MI355X guide, LDS section).
//  * bf16 / f16 use v_mfma_f32_32x32x16_{bf16,f16}; fp32 (the exact mode) uses four
//    exact-fp32 v_mfma_f32_32x32x2_f32 per 16-byte fragment with the k order permuted
//    identically on both operands.
#include "gemm_common.h"

#include <type_traits>

namespace {
using namespace bqk;

// ------------------------------------------------------------------ producers
template <typename T, int PROD, int MT, int NT>
__device__ __forceinline__ void produce(const GemmParams& p, unsigned char* smem, int stride,
                                        int m0, int tid) {
    constexpr int VEC = TT<T>::VEC;
    const int CH = p.K / VEC;
    const int cpp = CH < NT ? CH : NT;  // chunk columns handled per sweep
    const int RF = NT / cpp;            // rows in flight
    const int tc = tid % cpp, tr = tid / cpp;
    if (tr >= RF) return;
    const T* __restrict__ in = reinterpret_cast<const T*>(p.in);
    const int ldi = p.ldi;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    (void)zero4;

    for (int c = tc; c < CH; c += cpp) {
        const int ch0 = p.k_off + c * VEC;  // first input channel / unit of this chunk
        if constexpr (PROD == PROD_DW || PROD == PROD_DW_RELU) {
            const int H = p.H, W = p.W;
            float w[9][VEC];
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < VEC; j += 4) {
                    const float4 wv = *reinterpret_cast<const float4*>(p.dw + (size_t)t * ldi + ch0 + j);
                    w[t][j] = wv.x; w[t][j + 1] = wv.y; w[t][j + 2] = wv.z; w[t][j + 3] = wv.w;
                }
            // Two items per trip: all 18 tap loads of both pixels are issued before either is
            // consumed, doubling the bytes each wave keeps in flight (these layers are bound by
            // L2/HBM latency, not by the ALUs).
            constexpr int U = 2;
            PixIt it;
            it.init(m0 + tr, H, W);
            for (int r = tr; r < MT; r += U * RF) {
                uint4 v[U][9];
                bool ok[U][9];
                bool live[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ru = r + u * RF;
                    live[u] = ru < MT && m0 + ru < p.M;
                    // rows past the tensor clamp to a valid pixel (loaded, never stored)
                    const int img = live[u] ? it.img : 0, y = live[u] ? it.y : 0, x = live[u] ? it.x : 0;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int yy = y + dy - 1;
                        const bool vy = (unsigned)yy < (unsigned)H;
                        const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const int xx = x + dx - 1;
                            ok[u][dy * 3 + dx] = vy && ((unsigned)xx < (unsigned)W);
                            const int xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                            const size_t off = ((size_t)(img * H + yc) * W + xc) * ldi + ch0;
                            v[u][dy * 3 + dx] = *reinterpret_cast<const uint4*>(in + off);
                        }
                    }
                    it.advance(RF, H, W);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float acc[VEC];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const uint4 vv = ok[u][t] ? v[u][t] : zero4;
                        float f[VEC];
                        unpack<T>(vv, f);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) {
                            if (PROD == PROD_DW_RELU) f[j] = fmaxf(f[j], 0.f);
                            acc[j] = fmaf(w[t][j], f[j], acc[j]);
                        }
                    }
                    if (live[u])
                        *reinterpret_cast<uint4*>(smem + (size_t)(r + u * RF) * stride + c * 16) = pack<T>(acc);
                }
            }
        } else if constexpr (PROD == PROD_S2 || PROD == PROD_IM2COL) {
            // Pure gathers (1x1/s2 sampling, 3x3 valid im2col): issue UC independent 16-byte
            // loads per trip before the first LDS store so the wave keeps them all in flight.
            constexpr int UC = 6;
            const int H = p.H, W = p.W;  // output map
            int dy = 0, dx = 0, cc = ch0;
            if constexpr (PROD == PROD_IM2COL) {
                const int tap = ch0 / ldi;
                cc = ch0 - tap * ldi;
                dy = tap / 3;
                dx = tap - dy * 3;
            }
            PixIt it;
            it.init(m0 + tr, H, W);
            for (int r = tr; r < MT; r += UC * RF) {
                uint4 v[UC];
#pragma unroll
                for (int u = 0; u < UC; ++u) {
                    const int ru = r + u * RF;
                    const bool live = ru < MT && m0 + ru < p.M;
                    const int img = live ? it.img : 0, y = live ? it.y : 0, x = live ? it.x : 0;
                    size_t off;
                    if constexpr (PROD == PROD_S2)
                        off = ((size_t)(img * p.Hi + 2 * y) * p.Wi + 2 * x) * ldi + cc;
                    else
                        off = ((size_t)(img * p.Hi + y + dy) * p.Wi + x + dx) * ldi + cc;
                    v[u] = *reinterpret_cast<const uint4*>(in + off);
                    it.advance(RF, H, W);
                }
#pragma unroll
                for (int u = 0; u < UC; ++u) {
                    const int ru = r + u * RF;
                    if (ru < MT && m0 + ru < p.M)
                        *reinterpret_cast<uint4*>(smem + (size_t)ru * stride + c * 16) = v[u];
                }
            }
        }
    }
}

// ------------------------------------------------------------------ the kernel
template <typename T, int PROD, int MF, int WM, int WN, int RN, int PF>
__global__ void __launch_bounds__(64 * WM * WN) gemm_fused_kernel(const GemmParams p) {
    constexpr int VEC = TT<T>::VEC;
    constexpr int NT = 64 * WM * WN;
    constexpr int MT = 32 * MF;
    constexpr int RM = MF / WM;
    static_assert(MF % WM == 0, "row fragments must split evenly over WM");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    if constexpr (std::is_same<T, f16_t>::value) bq_f16_saturate();
    const int tid = threadIdx.x;
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int m0 = tile * MT;
    const int CH = p.K / VEC;
    const int stride = (CH | 1) * 16;

    produce<T, PROD, MT, NT>(p, smem, stride, m0, tid);
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r32 = lane & 31, h = lane >> 5;
    const int KB = p.K / (2 * VEC);
    const unsigned char* a_base = smem + (size_t)(wm * RM * 32 + r32) * stride + h * 16;
    const uint4* __restrict__ wp = reinterpret_cast<const uint4*>(p.wp);
    for (int nfb = wn * RN; nfb < p.NFp; nfb += WN * RN) {
        f32x16 acc[RM][RN];
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        const uint4* bp[RN];
#pragma unroll
        for (int j = 0; j < RN; ++j)
            bp[j] = wp + ((size_t)(nfb + j) * p.KBtot + p.kb0) * 64 + lane;

        uint4 bq[PF][RN];
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int idx = d < KB ? d : KB - 1;
#pragma unroll
            for (int j = 0; j < RN; ++j) bq[d][j] = bp[j][(size_t)idx * 64];
        }

        int kb = 0;
        for (; kb + PF <= KB; kb += PF) {
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                uint4 a[RM];
#pragma unroll
                for (int i = 0; i < RM; ++i)
                    a[i] = *reinterpret_cast<const uint4*>(a_base + (size_t)i * 32 * stride + (kb + d) * 32);
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j) mma<T>(acc[i][j], bq[d][j], a[i]);
                const int nx = kb + d + PF;
                const int idx = nx < KB ? nx : KB - 1;
#pragma unroll
                for (int j = 0; j < RN; ++j) bq[d][j] = bp[j][(size_t)idx * 64];
            }
        }
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            if (kb + d < KB) {
                uint4 a[RM];
#pragma unroll
                for (int i = 0; i < RM; ++i)
                    a[i] = *reinterpret_cast<const uint4*>(a_base + (size_t)i * 32 * stride + (kb + d) * 32);
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j) mma<T>(acc[i][j], bq[d][j], a[i]);
            }
        }

        if (p.NFp == WN * RN) {
            // single pass over N: the A tile is dead, reuse LDS to coalesce the stores
            __syncthreads();
            epilogue_to_lds<T, RM, RN>(p, acc, nfb, wm * RM * 32, m0, r32, h, smem);
            __syncthreads();
            lds_rows_to_global<T, NT, MT>(p, m0, tid, smem);
        } else {
            epilogue<T, RM, RN>(p, acc, nfb, m0 + wm * RM * 32, r32, h);
        }
    }
}

struct ShapeDesc { int MF, WM, WN, RN; };
constexpr ShapeDesc kShapes[11] = {{4, 2, 2, 1}, {4, 1, 4, 1}, {4, 1, 4, 2}, {3, 1, 8, 3}, {3, 1, 8, 2}, {2, 1, 8, 3},
                                   {1, 1, 8, 4}, {1, 1, 4, 2}, {1, 1, 8, 2}, {2, 1, 4, 1}, {2, 1, 4, 2}};

template <typename T, int PROD, int SH>
int launch_inst(const GemmParams& p, hipStream_t s) {
    constexpr ShapeDesc d = kShapes[SH];
    constexpr int PF = 4;
    auto kern = gemm_fused_kernel<T, PROD, d.MF, d.WM, d.WN, d.RN, PF>;
    const int vec = TT<T>::VEC;
    size_t lds = (size_t)(((p.K / vec) | 1) * 16) * (32 * d.MF);
    if (p.NFp == d.WN * d.RN) {   // staged epilogue needs MT rows of the output tile
        const size_t stage = (size_t)(32 * d.MF) * (p.Nstore * sizeof(T) + 16);
        if (stage > lds) lds = stage;
    }
    static BqLdsAttr attr;
    if (const int e = attr.ensure(reinterpret_cast<const void*>(kern), lds)) return e;
    if (p.NFp % (d.WN * d.RN) != 0 || p.K % (2 * vec) != 0 || lds > 160 * 1024)
        return (int)hipErrorInvalidValue;
    const int mt = 32 * d.MF;
    const int grid = (p.M + mt - 1) / mt;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * d.WM * d.WN), lds, s, p);
    return (int)hipGetLastError();
}

}  // namespace

int gemm_tile_rows(int shape) { return 32 * kShapes[shape].MF; }

size_t gemm_lds_bytes(int dtype, int shape, int K) {
    const int vec = dtype != 0 ? 8 : 4;
    return (size_t)(((K / vec) | 1) * 16) * (32 * kShapes[shape].MF);
}

#define BQ_CASE(T, PROD, SH) \
    if (prod == PROD && shape == SH) return launch_inst<T, PROD, SH>(p, s);

int launch_gemm(int dtype, int prod, int shape, const GemmParams& p, hipStream_t s) {
    if (dtype == 1) {
        BQ_CASE(bf16_t, PROD_IM2COL, SHAPE_A)
        BQ_CASE(bf16_t, PROD_DW, SHAPE_B)
        BQ_CASE(bf16_t, PROD_S2, SHAPE_B)
        BQ_CASE(bf16_t, PROD_DW, SHAPE_C)
        BQ_CASE(bf16_t, PROD_DW_RELU, SHAPE_C)
        BQ_CASE(bf16_t, PROD_S2, SHAPE_C)
        BQ_CASE(bf16_t, PROD_DW, SHAPE_D)
        BQ_CASE(bf16_t, PROD_DW_RELU, SHAPE_D)
        BQ_CASE(bf16_t, PROD_S2, SHAPE_D)
        BQ_CASE(bf16_t, PROD_DW, SHAPE_E)
        BQ_CASE(bf16_t, PROD_S2, SHAPE_E)
        BQ_CASE(bf16_t, PROD_S2, SHAPE_K)
        BQ_CASE(bf16_t, PROD_DW, SHAPE_F)
        BQ_CASE(bf16_t, PROD_DW, SHAPE_G)
    } else if (dtype == 2) {
        BQ_CASE(f16_t, PROD_IM2COL, SHAPE_A)
        BQ_CASE(f16_t, PROD_DW, SHAPE_B)
        BQ_CASE(f16_t, PROD_S2, SHAPE_B)
        BQ_CASE(f16_t, PROD_DW, SHAPE_C)
        BQ_CASE(f16_t, PROD_DW_RELU, SHAPE_C)
        BQ_CASE(f16_t, PROD_S2, SHAPE_C)
        BQ_CASE(f16_t, PROD_DW, SHAPE_D)
        BQ_CASE(f16_t, PROD_DW_RELU, SHAPE_D)
        BQ_CASE(f16_t, PROD_S2, SHAPE_D)
        BQ_CASE(f16_t, PROD_DW, SHAPE_E)
        BQ_CASE(f16_t, PROD_S2, SHAPE_E)
        BQ_CASE(f16_t, PROD_S2, SHAPE_K)
        BQ_CASE(f16_t, PROD_DW, SHAPE_F)
        BQ_CASE(f16_t, PROD_DW, SHAPE_G)
    } else {
        BQ_CASE(float, PROD_IM2COL, SHAPE_A)
        BQ_CASE(float, PROD_DW, SHAPE_B)
        BQ_CASE(float, PROD_S2, SHAPE_B)
        BQ_CASE(float, PROD_DW, SHAPE_C)
        BQ_CASE(float, PROD_DW_RELU, SHAPE_C)
        BQ_CASE(float, PROD_S2, SHAPE_C)
        BQ_CASE(float, PROD_DW, SHAPE_H)
        BQ_CASE(float, PROD_DW_RELU, SHAPE_H)
        BQ_CASE(float, PROD_S2, SHAPE_H)
    }
    return (int)hipErrorInvalidValue;
}
