// MC-dropout head, the two Dense(1024, relu) layers (K6; biscuit/hp.py:11,13,21; the UQ loop of results.py:257-258):
//     out[row][n] = relu( bias[n] + sum_k dropout(x[row][k]) * W[k][n] ),   rows = tiles x passes
// Round 3.  Rounds 1-2 ran these as exact-fp32 MFMAs (v_mfma_f32_32x32x2_f32, 1/16 of the 16-bit matrix rate): 0.48 ms
// per batch, 68 % of that pipe's peak.  Here every fp32 operand is split on the fly into two IEEE halves,
//     v = hi + lo / 2^11,   hi = f16(v),   lo = f16((v - hi) * 2^11)      (22 significand bits together; the scale keeps
//                                                                          lo out of the subnormals)
// and the product is three f16 MFMAs with fp32 accumulation,
//     x * w  ~  hi_x hi_w  +  2^-11 (hi_x lo_w + lo_x hi_w)               (the dropped lo lo term is 2^-22 of the product)
// into two accumulator sets that are combined once at the end.  Every f16 x f16 product is exact in fp32, so the result
// carries the rounding of the fp32 accumulation only: measured against float64 on this network's own tensors the
// three-term form is as close as a plain fp32 GEMM (rms 4.6e-7 both, oracle tolerance of the tests 1e-6).  It is the
// head of ALL three storage types -- the backbone's type never reaches it.
//
// Workgroup = 8 waves = 64 rows x 512 columns (grid: row tiles x 2 column halves: 240 workgroups at 7 680 rows, and half
// the weight traffic of 32-row tiles); wave w owns the 2 x 2 fragments of 32x32 at columns [64 w, 64 w + 64).  K is
// walked in chunks of 512: the producer draws the Philox4x32-10 masks (counter = (unit / 4, layer, pass, global tile),
// key = seed: the contract of oracle/philox.py), applies the inverted dropout, splits and writes the two A planes to
// LDS; the matrix stage reads A fragments from LDS and streams the pre-split weight fragments (host-packed hi / lo in
// fragment order) from L2 through a register ring.
#include "gemm_common.h"

namespace {
using namespace bqk;

constexpr int HM = 64, HN = 512;            // rows, columns per workgroup
constexpr float HSCALE = 2048.f, HINV = 1.f / 2048.f;

struct HeadParams {
    const float* in;        // [tiles][K] (in_row_is_tile) or [rows][K]
    const uint4* wh;        // f16(W) in fragment order [1024 / 32][K / 16][64] x 16 B
    const uint4* wl;        // f16((W - hi) * 2^11), same order
    const float* bias;      // [1024]
    float* out;             // [rows][1024]
    int rows, K;
    int mc_n, pass0, in_row_is_tile, layer;
    unsigned seed_lo, seed_hi, thresh;
    float dscale;
    long long tile0;
    const long long* tile0_dev;
    const long long* tile_idx;  // per-tile Philox indices [tiles] (bq_set_tile_index_array) or null: tile i counts as tile0 + i
};

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    hi = H16<f16_t>::pack2(a, b);
    const float ra = (a - H16<f16_t>::lo(hi)) * HSCALE, rb = (b - H16<f16_t>::hi(hi)) * HSCALE;
    lo = H16<f16_t>::pack2(ra, rb);
}

// Producer and matrix stage overlapped (round 5; rounds 3-4 ran them in lock step -- barrier, every wave draws Philox masks and
// splits (vector ALU, ~9 us per 512-deep chunk), barrier, every wave multiplies (matrix pipe, ~12 us) -- so each pipe idled while the
// other worked: 0.155 ms for layer 0 where this takes 0.123).  The A planes are double-buffered (chunks of 256: 2 x 2 x 33 KB of
// LDS) and a chunk step is "produce chunk c + 1, multiply chunk c" for waves 0-3 and "multiply chunk c, produce chunk c + 1" for
// waves 4-7: waves w and w + 4 of a workgroup share a SIMD, so the two waves of a SIMD are in opposite stages; one barrier per chunk.
constexpr int PKC = 256;
constexpr int PSTR = PKC * 2 + 16;          // 33 slots of 16 B
constexpr int PPLANE = HM * PSTR;

__global__ void __launch_bounds__(512) head_dense_pipe_kernel(const HeadParams p) {
    bq_f16_saturate();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [buffer][hi | lo] planes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * HM;
    const int nf0 = blockIdx.y * (HN / 32) + wave * 2;
    const int KB = p.K / 16;
    const long long tile0 = p.tile0 + (p.tile0_dev ? *p.tile0_dev : 0);
    const int prow = tid >> 3, pseg = tid & 7;
    const int pm = m0 + prow;
    const bool plive = pm < p.rows;
    const int ptile = plive ? pm / p.mc_n : 0;
    const int ppass = p.pass0 + (plive ? pm - ptile * p.mc_n : 0);
    const unsigned pctr = (unsigned)(tile0 + (p.tile_idx ? p.tile_idx[ptile] : (long long)ptile));   // the Philox tile counter of this row
    const float* prow_ptr = p.in + (size_t)(p.in_row_is_tile ? ptile : (plive ? pm : 0)) * p.K;

    f32x16 acc1[2][2], acc2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc1[i][j][e] = 0.f; acc2[i][j][e] = 0.f; }
    const uint4* bh0 = p.wh + (size_t)nf0 * KB * 64 + lane;
    const uint4* bl0 = p.wl + (size_t)nf0 * KB * 64 + lane;
    constexpr int PF = 2;
    uint4 bh[PF][2], bl[PF][2];
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bh[d][j] = bh0[((size_t)j * KB + d) * 64];
            bl[d][j] = bl0[((size_t)j * KB + d) * 64];
        }

    auto produce = [&](int k0, int buf) {
        unsigned char* const base = smem + buf * 2 * PPLANE;
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(prow_ptr + k0 + (i * 8 + pseg) * 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int gi = i * 8 + pseg;
            const int g = (k0 >> 2) + gi;
            unsigned rnd[4];
            philox4x32_10((unsigned)g, (unsigned)p.layer, (unsigned)ppass, pctr, p.seed_lo, p.seed_hi, rnd);
            const float f0 = (plive && rnd[0] >= p.thresh) ? v[i].x * p.dscale : 0.f;
            const float f1 = (plive && rnd[1] >= p.thresh) ? v[i].y * p.dscale : 0.f;
            const float f2 = (plive && rnd[2] >= p.thresh) ? v[i].z * p.dscale : 0.f;
            const float f3 = (plive && rnd[3] >= p.thresh) ? v[i].w * p.dscale : 0.f;
            uint2 hi, lo;
            split2(f0, f1, hi.x, lo.x);
            split2(f2, f3, hi.y, lo.y);
            const int off = prow * PSTR + gi * 8;
            *reinterpret_cast<uint2*>(base + off) = hi;
            *reinterpret_cast<uint2*>(base + PPLANE + off) = lo;
        }
    };
    auto multiply = [&](int k0, int buf) {
        const unsigned char* a_hi = smem + buf * 2 * PPLANE + r32 * PSTR + h * 16;
        const unsigned char* a_lo = a_hi + PPLANE;
        const int kb0 = k0 / 16;
#pragma unroll 2
        for (int kl = 0; kl < PKC / 16; ++kl) {
            const int d = kl & (PF - 1);
            uint4 ah[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const uint4*>(a_hi + i * 32 * PSTR + kl * 32);
                al[i] = *reinterpret_cast<const uint4*>(a_lo + i * 32 * PSTR + kl * 32);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    mma<f16_t>(acc1[i][j], bh[d][j], ah[i]);
                    mma<f16_t>(acc2[i][j], bh[d][j], al[i]);
                    mma<f16_t>(acc2[i][j], bl[d][j], ah[i]);
                }
            const int nx = kb0 + kl + PF;
            const int idx = nx < KB ? nx : KB - 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[d][j] = bh0[((size_t)j * KB + idx) * 64];
                bl[d][j] = bl0[((size_t)j * KB + idx) * 64];
            }
        }
    };

    const int NC = p.K / PKC;
    produce(0, 0);
    __syncthreads();
    const bool first_produce = (wave >> 2) == 0;      // (wave-uniform)
    for (int c = 0; c < NC; ++c) {
        const bool more = c + 1 < NC;
        if (first_produce) {
            if (more) produce((c + 1) * PKC, (c + 1) & 1);
            multiply(c * PKC, c & 1);
        } else {
            multiply(c * PKC, c & 1);
            if (more) produce((c + 1) * PKC, (c + 1) & 1);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n0 = (nf0 + j) * 32 + g * 8 + h * 4;
            const float4 b = *reinterpret_cast<const float4*>(p.bias + n0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = m0 + i * 32 + r32;
                if (m < p.rows) {
                    float4 o;
                    o.x = fmaxf(fmaf(acc2[i][j][g * 4 + 0], HINV, acc1[i][j][g * 4 + 0]) + b.x, 0.f);
                    o.y = fmaxf(fmaf(acc2[i][j][g * 4 + 1], HINV, acc1[i][j][g * 4 + 1]) + b.y, 0.f);
                    o.z = fmaxf(fmaf(acc2[i][j][g * 4 + 2], HINV, acc1[i][j][g * 4 + 2]) + b.z, 0.f);
                    o.w = fmaxf(fmaf(acc2[i][j][g * 4 + 3], HINV, acc1[i][j][g * 4 + 3]) + b.w, 0.f);
                    *reinterpret_cast<float4*>(p.out + (size_t)m * 1024 + n0) = o;
                }
            }
        }
}

}  // namespace

// One Dense(1024, relu) layer of the MC head over `rows` = tiles x passes rows.  K = 2048 (layer 0) or 1024 (layer 1).
int launch_head_dense(const float* in, const void* wh, const void* wl, const float* bias, float* out, int rows, int K,
                      int mc_n, int pass0, int in_row_is_tile, int layer, unsigned seed_lo, unsigned seed_hi, unsigned thresh,
                      float dscale, long long tile0, const long long* tile0_dev, const long long* tile_idx, hipStream_t s) {
    if (rows <= 0) return 0;
    if (K % PKC != 0 || !wh || !wl || !bias) return (int)hipErrorInvalidValue;
    HeadParams p;
    p.in = in; p.wh = reinterpret_cast<const uint4*>(wh); p.wl = reinterpret_cast<const uint4*>(wl);
    p.bias = bias; p.out = out; p.rows = rows; p.K = K;
    p.mc_n = mc_n; p.pass0 = pass0; p.in_row_is_tile = in_row_is_tile; p.layer = layer;
    p.seed_lo = seed_lo; p.seed_hi = seed_hi; p.thresh = thresh; p.dscale = dscale;
    p.tile0 = tile0; p.tile0_dev = tile0_dev; p.tile_idx = tile_idx;
    constexpr size_t ldsp = 4 * (size_t)PPLANE;
    static BqLdsAttr attr;
    if (const int e = attr.ensure(reinterpret_cast<const void*>(head_dense_pipe_kernel), ldsp)) return e;
    hipLaunchKernelGGL(head_dense_pipe_kernel, dim3((rows + HM - 1) / HM, 1024 / HN), dim3(512), ldsp, s, p);
    return (int)hipGetLastError();
}
