// Software-pipelined SeparableConv2D kernel for the 728-wide layers (27 of the 34 separable
// convolutions, ~80 % of the network's FLOPs), bf16.
//
// One workgroup of 8 waves owns MT = 96 consecutive output pixels and ALL 768 (padded)
// output channels: every wave keeps a 96 x 96 fp32 accumulator block in registers for the
// whole kernel, so the contraction can be walked in 64-channel chunks and the A operand
// (the depthwise result) only ever exists as two 13.5 KB LDS chunk buffers.  Per chunk c a
// three-stage pipeline runs inside the workgroup, one barrier per stage:
//     L(c+2)  coalesced 16-byte global loads of the raw input halo rows -> registers -> LDS
//             (each input byte is fetched from L2/HBM once, not nine times)
//     D(c+1)  depthwise 3x3 on the vector ALU from the LDS halo rows -> A chunk (bf16)
//     G(c)    36 v_mfma_f32_32x32x16_bf16 per wave on A chunk c, B fragments streamed from
//             L2 through a 4-deep register ring (weights are pre-swizzled to fragment order)
// The vector-ALU depthwise work and the LDS traffic of stage D sit in the shadow of the
// matrix-core work of stage G (separate pipes), and no stage waits on a global load issued
// in the same iteration.  Depthwise taps live in LDS as fp32 (26 KB for 736 channels).
#include "gemm_common.h"

namespace {
using namespace bqk;

constexpr int KC = 64;             // channels per pipeline chunk
constexpr int CPR = KC / 8;        // 16-byte pieces per raw pixel row (8)
constexpr int RAW_ROW = KC * 2;    // 128 B of one pixel's chunk
constexpr int A_STR = KC * 2 + 16; // 144 B: odd number of 16-byte slots -> conflict-free ds_read_b128

template <bool RELU, int MF, int WN, int RN>
__global__ void __launch_bounds__(64 * WN) sepconv_pipe_kernel(const GemmParams p) {
    constexpr int NT = 64 * WN;
    constexpr int MT = 32 * MF;
    constexpr int NRAW = 3;                        // raw 16-byte loads per thread per chunk (max)
    constexpr int NITEM = (MT * CPR + NT - 1) / NT;
    constexpr int KBC = KC / 16;                   // k-blocks per chunk (4)
    constexpr int PF = 2;                          // B register ring depth (k-blocks ahead)
    static_assert(NT % CPR == 0, "a thread must keep the same channel piece for all its items");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int W = p.W, H = p.H;
    const int HP = MT + 2 * (W + 1);               // halo rows of the flattened pixel range
    unsigned char* raw[2] = {smem, smem + (size_t)HP * RAW_ROW};
    unsigned char* abuf[2] = {raw[1] + (size_t)HP * RAW_ROW, raw[1] + (size_t)HP * RAW_ROW + MT * A_STR};
    float* wl = reinterpret_cast<float*>(abuf[1] + MT * A_STR);     // [9][K] depthwise taps

    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int m0 = tile * MT;
    const int p_lo = m0 - (W + 1);
    const int K = p.K;                             // padded input channels (multiple of 16)
    const int KB = K / 16;
    const int NC = (K + KC - 1) / KC;
    const bf16_t* __restrict__ in = reinterpret_cast<const bf16_t*>(p.in);
    const int ldi = p.ldi;

    // depthwise taps -> LDS
    for (int i = tid * 4; i < 9 * K; i += NT * 4) {
        const int t = i / K, k = i - t * K;
        *reinterpret_cast<float4*>(wl + i) = *reinterpret_cast<const float4*>(p.dw + (size_t)t * ldi + k);
    }

    // ---- per-thread constants of the depthwise stage (independent of the chunk)
    const int jch = tid & (CPR - 1);               // this thread's 16-byte piece (8 channels)
    int item_row[NITEM];
    unsigned item_mask[NITEM];                     // 9 validity bits ('same' zero padding)
#pragma unroll
    for (int q = 0; q < NITEM; ++q) {
        const int r = (tid + q * NT) >> 3;
        item_row[q] = r;
        unsigned bits = 0;
        if (r < MT && m0 + r < p.M) {
            PixIt it;
            it.init(m0 + r, H, W);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int yy = it.y + dy - 1, xx = it.x + dx - 1;
                    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) bits |= 1u << (dy * 3 + dx);
                }
        }
        item_mask[q] = bits;
    }
    // ---- raw staging map: piece idx = tid + q*NT -> halo row idx>>3, 16-byte piece idx&7 == jch
    uint4 rreg[NRAW];
    auto raw_load = [&](int c) {
        // pieces past K (last chunk of a K that is not a multiple of 64) are neither loaded
        // nor used
        if (c * KC + jch * 8 < K) {
#pragma unroll
            for (int q = 0; q < NRAW; ++q) {
                const int row = (tid + q * NT) >> 3;
                const int prow = p_lo + row;
                if (row < HP && prow >= 0 && prow < p.M)
                    rreg[q] = *reinterpret_cast<const uint4*>(in + (size_t)prow * ldi + c * KC + jch * 8);
            }
        }
    };
    auto raw_store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NRAW; ++q) {
            const int row = (tid + q * NT) >> 3;
            const int prow = p_lo + row;
            if (row < HP && prow >= 0 && prow < p.M)
                *reinterpret_cast<uint4*>(raw[buf] + row * RAW_ROW + jch * 16) = rreg[q];
        }
    };
    // D stage: depthwise of chunk c from raw[buf] into abuf[buf2]
    auto depthwise = [&](int c, const unsigned char* rb, unsigned char* ab) {
        const int kvalid = K - c * KC;             // channels of this chunk that exist
        if (jch * 8 >= kvalid) return;             // padded tail of the last chunk: A never read there
        const float* wc = wl + c * KC + jch * 8;
#pragma unroll
        for (int q = 0; q < NITEM; ++q) {
            const int r = item_row[q];
            if (r < MT) {                          // wave-uniform (NT and MT*8 are multiples of 64)
                float acc[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = 0.f;
                const unsigned char* base = rb + (r + W + 1) * RAW_ROW + jch * 16;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int off = ((t / 3 - 1) * W + (t % 3 - 1)) * RAW_ROW;
                    uint4 v = *reinterpret_cast<const uint4*>(base + off);
                    if (!((item_mask[q] >> t) & 1u)) v = make_uint4(0, 0, 0, 0);
                    const float4 w0 = *reinterpret_cast<const float4*>(wc + t * K);
                    const float4 w1 = *reinterpret_cast<const float4*>(wc + t * K + 4);
                    float f[8];
                    unpack<bf16_t>(v, f);
                    if (RELU) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j], 0.f);
                    }
                    acc[0] = fmaf(w0.x, f[0], acc[0]); acc[1] = fmaf(w0.y, f[1], acc[1]);
                    acc[2] = fmaf(w0.z, f[2], acc[2]); acc[3] = fmaf(w0.w, f[3], acc[3]);
                    acc[4] = fmaf(w1.x, f[4], acc[4]); acc[5] = fmaf(w1.y, f[5], acc[5]);
                    acc[6] = fmaf(w1.z, f[6], acc[6]); acc[7] = fmaf(w1.w, f[7], acc[7]);
                }
                *reinterpret_cast<uint4*>(ab + r * A_STR + jch * 16) = pack<bf16_t>(acc);
            }
        }
    };

    // ---- prologue: raw(0) -> LDS, D(0), raw(1) -> LDS, raw(2) in flight
    raw_load(0);
    raw_store(0);
    raw_load(1);
    __syncthreads();                               // raw[0] and the taps are visible
    depthwise(0, raw[0], abuf[0]);
    raw_store(1);
    raw_load(2);

    const int lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    const int nfb = wave * RN;                     // single pass over N: NFp == WN*RN
    const uint4* __restrict__ wp = reinterpret_cast<const uint4*>(p.wp);
    const uint4* bp0 = wp + ((size_t)nfb * p.KBtot + p.kb0) * 64 + lane;
    uint4 bq[PF][RN];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int idx = d < KB ? d : KB - 1;
#pragma unroll
        for (int j = 0; j < RN; ++j) bq[d][j] = bp0[((size_t)j * p.KBtot + idx) * 64];
    }
    f32x16 acc[MF][RN];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    __syncthreads();                               // A(0) and raw[1] visible

    for (int c = 0; c < NC; ++c) {
        // L: raw chunk c+2 (loaded during the previous iteration) -> raw[c&1], whose last
        // reader D(c) finished before the previous barrier; then start loading chunk c+3
        if (c + 2 < NC) raw_store(c & 1);
        raw_load(c + 3);
        // D: depthwise of chunk c+1
        if (c + 1 < NC) depthwise(c + 1, raw[(c + 1) & 1], abuf[(c + 1) & 1]);
        // G: matrix cores on chunk c
        const unsigned char* a_base = abuf[c & 1] + (size_t)r32 * A_STR + h * 16;
#pragma unroll
        for (int d = 0; d < KBC; ++d) {
            const int kb = c * KBC + d;
            if (kb < KB) {
                uint4 a[MF];
#pragma unroll
                for (int i = 0; i < MF; ++i)
                    a[i] = *reinterpret_cast<const uint4*>(a_base + (size_t)i * 32 * A_STR + d * 32);
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j) mma<bf16_t>(acc[i][j], bq[d & (PF - 1)][j], a[i]);
                const int nx = kb + PF;
                const int idx = nx < KB ? nx : KB - 1;
#pragma unroll
                for (int j = 0; j < RN; ++j) bq[d & (PF - 1)][j] = bp0[((size_t)j * p.KBtot + idx) * 64];
            }
        }
        __syncthreads();
    }
    epilogue<bf16_t, MF, RN>(p, acc, nfb, m0, r32, h);
}

template <bool RELU>
int launch_pipe(const GemmParams& p, hipStream_t s) {
    constexpr int MF = 3, WN = 8, RN = 3;
    auto kern = sepconv_pipe_kernel<RELU, MF, WN, RN>;
    const int MT = 32 * MF;
    const int HP = MT + 2 * (p.W + 1);
    const size_t lds = (size_t)2 * HP * RAW_ROW + 2 * MT * A_STR + (size_t)9 * p.K * 4;
    if (p.NFp != WN * RN || p.K % 16 != 0 || HP * CPR > 3 * 64 * WN || lds > 160 * 1024 || p.k_off != 0)
        return (int)hipErrorInvalidValue;
    static size_t lds_set = 0;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        lds_set = lds;
    }
    const int grid = (p.M + MT - 1) / MT;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WN), lds, s, p);
    return (int)hipGetLastError();
}

}  // namespace

// bf16 SeparableConv2D with NFp == 24 (728-wide outputs) on maps up to 37x37.
bool pipe_supported(int dtype, int prod, int nfp, int W, int K) {
    if (dtype != 1 || (prod != PROD_DW && prod != PROD_DW_RELU) || nfp != 24 || K % 16 != 0) return false;
    const int HP = 96 + 2 * (W + 1);
    const size_t lds = (size_t)2 * HP * RAW_ROW + 2 * 96 * A_STR + (size_t)9 * K * 4;
    return HP * CPR <= 3 * 512 && lds <= 160 * 1024;
}

int launch_sepconv_pipe(int prod, const GemmParams& p, hipStream_t s) {
    return prod == PROD_DW_RELU ? launch_pipe<true>(p, s) : launch_pipe<false>(p, s);
}
