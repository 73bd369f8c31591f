// Pointwise convolutions of the exit flow (block 14: 1024 -> 1536 and 1536 -> 2048 on 10x10 maps), 16-bit storage, round 4:
// the GEMM behind the depthwise kernel of kernels_split.hip, rebuilt around what bounded it there.
//
// The 128 x 128-tile kernel gives a wave 64 x 64 outputs: per 16-deep k-step two A fragments from LDS and two weight
// fragments from L2 for four MFMAs -- 2 KB through the vector L1 per 128 MFMA cycles and wave, 64 B/clk per CU: all the L1
// delivers (the two waves of a workgroup that share a column half fetch the same fragments twice).  0.27 ms for the last
// layer (600 TFLOP/s) where the vendor GEMM takes 0.15 on the same shape, and with the global average pool as its epilogue 28
// of the 128 rows of a tile are another image's pixels, computed and dropped.
//
// Here a workgroup owns ONE IMAGE's 100 pixels x 256 output channels: seven 16-row fragments (112 rows, 89 % used) x four
// waves of 64 channels.  Per 32-deep k-step a wave reads 7 A fragments from LDS and 4 weight fragments from L2 for 28
// v_mfma_f32_16x16x32 (448 cycles): 9 B/clk of L1 and 16 B/clk of LDS per wave -- a third of either at 8 waves per CU -- and
// no weight fragment is fetched twice by a workgroup.  The weights are the A operand (D[cout][pixel]) in the fragment order of
// the other 16x16x32 kernels ("<layer>/wp16", interleaved fragment pairs), so a lane holds 8 consecutive channels of its pixel
// per pair: the plain epilogue is 16-byte stores straight from the accumulators, the pooling epilogue sums a lane's seven
// pixels, then the sixteen lanes of a row by DPP -- no staging tile, no second pass.  256 images x 8 (6) column tiles = 4 (3)
// workgroups for each of the 512 slots: no partial round.
#include "mfma16_common.h"

namespace {

constexpr int XMF = 7;                      // 16-row fragments per image (H W <= 112)
constexpr int XN = 256, XK = 64;            // output channels per workgroup, channels per chunk
constexpr int XA_STR = XK * 2 + 32;         // 160 B: the 16x16x32 operand read is conflict-free at 10 slots per row
constexpr int XA_BUF = 128 * XA_STR;        // (staging writes 128 rows: 4 pieces per thread)

template <typename T>
struct ExitParams {
    const T* in;          // [n * HW][K]: the depthwise result
    const uint4* wp16;    // [K / 32][N / 16][64] x 16 B
    const float* scale;   // [N] folded BN
    const float* bias;
    T* out;               // [n * HW][N]       (GAP: unused)
    float* gap;           // [n][N] fp32 means (GAP only)
    int n, HW, K, N, relu;
    float gap_mul;        // GAP: factor on the means (2^k: undoes the activation exponent of the pooled tensor; 1.0f otherwise)
};

template <typename T, bool GAP>
__global__ void __launch_bounds__(256, 2) exit_gemm_kernel(const ExitParams<T> p) {
    if constexpr (H16<T>::F16) bq_f16_saturate();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, g = lane >> 4;
    const int ntn = p.N / XN;
    const int tile = xcd_tile(blockIdx.x, gridDim.x);             // column tiles of an image are neighbours on an XCD: they share its A rows in L2
    const int img = tile / ntn, nt = tile - img * ntn;
    const int NF = p.N / 16, NC = p.K / XK;
    const int n0 = nt * XN + wave * 64;                           // this wave's 64 channels: fragments n0 / 16 .. + 3

    // A staging: piece = 16 B of a pixel's 64-channel chunk, four pieces per thread (rows >= HW: a copy of the last pixel, dropped)
    const int jp = tid & 7;
    const T* arow[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int row = (tid + q * 256) >> 3;
        row = row < p.HW ? row : p.HW - 1;
        arow[q] = p.in + ((size_t)img * p.HW + row) * p.K + jp * 8;
    }
    // (a native vector type: as an array of HIP's uint4 struct the four vectors stayed in scratch memory, 80 bytes per lane)
#define XLOAD_A(c)  _Pragma("unroll") for (int q = 0; q < 4; ++q) areg[q] = *reinterpret_cast<const u32x4s*>(arow[q] + (c) * XK)
#define XSTORE_A(buf) _Pragma("unroll") for (int q = 0; q < 4; ++q) \
        *reinterpret_cast<u32x4s*>(smem + (buf) * XA_BUF + ((tid + q * 256) >> 3) * XA_STR + jp * 16) = areg[q]
    u32x4s areg[4];
    XLOAD_A(0);
    XSTORE_A(0);
    // The vector-memory queue enters the K loop in the shape an iteration leaves it in -- weight fragments of k-step 0, of
    // k-step 1, then the next A chunk -- because the compiler's wait in front of the loop's first MFMA is the minimum over
    // both ways in (with the A chunk requested first it was vmcnt(0) at the top of every chunk).
    const uint4* const bp = p.wp16 + ((size_t)(n0 >> 4)) * 64 + lane;         // + (ks * NF + j) * 64
    uint4 bq[2][4];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bq[d][j] = bp[((size_t)d * NF + j) * 64];
        __builtin_amdgcn_sched_barrier(0);
    }
    XLOAD_A(1);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[XMF][4];
#pragma unroll
    for (int i = 0; i < XMF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const int KS = p.K / 32;
    for (int c = 0; c < NC; ++c) {
        const unsigned char* const ab = smem + (c & 1) * XA_BUF + px * XA_STR + g * 16;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            uint4 a[XMF];
#pragma unroll
            for (int i = 0; i < XMF; ++i) a[i] = *reinterpret_cast<const uint4*>(ab + i * 16 * XA_STR + d * 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < XMF; ++i) acc[i][j] = mma16<T>(bq[d][j], a[i], acc[i][j]);
                // the fragment is dead: fetch the one two k-steps on (past the end: any valid address)
                const int nx = 2 * c + d + 2;
                bq[d][j] = bp[((size_t)(nx < KS ? nx : 0) * NF + j) * 64];
                __builtin_amdgcn_sched_barrier(0);                 // (the reloads stay in fragment order, behind their last use)
            }
        }
        // chunk c+1 (in registers since the last iteration) -> the other buffer, whose readers finished before the last barrier
        // (unconditional; past the end a clamped repeat nobody reads)
        XSTORE_A((c + 1) & 1);
        XLOAD_A(c + 2 < NC ? c + 2 : NC - 1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }

#undef XLOAD_A
#undef XSTORE_A

    // ---- epilogue: folded BN (+ ReLU), rounded to the storage type, straight from the accumulators
    const unsigned lo2 = p.relu ? 0u : 0x80008000u;               // ReLU = packed signed 16-bit max with 0 (0x8000: no-op)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int ch = n0 + 32 * q + 8 * g;                       // the lane's eight channels of this pair
        const float4 s0 = *reinterpret_cast<const float4*>(p.scale + ch), s1 = *reinterpret_cast<const float4*>(p.scale + ch + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + ch), b1 = *reinterpret_cast<const float4*>(p.bias + ch + 4);
        float sum[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[e] = 0.f;
#pragma unroll
        for (int i = 0; i < XMF; ++i) {
            unsigned o[4];
            bn_pair4<T>(acc[i][2 * q], acc[i][2 * q + 1], s0, s1, b0, b1, o);
#pragma unroll
            for (int e = 0; e < 4; ++e) asm("v_pk_max_i16 %0, %1, %2" : "=v"(o[e]) : "v"(o[e]), "v"(lo2));
            const int row = 16 * i + px;
            if constexpr (GAP) {
                // the mean is taken over the ROUNDED tensor (the rounding point of the activation GlobalAveragePooling2D read)
                if (row < p.HW) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x2s v = unpack2<T>(o[e]);
                        sum[2 * e] += v.x; sum[2 * e + 1] += v.y;
                    }
                }
            } else {
                if (row < p.HW)
                    *reinterpret_cast<uint4*>(p.out + ((size_t)img * p.HW + row) * p.N + ch) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
        if constexpr (GAP) {
            // sixteen pixel lanes of a row -> lane 15: row_shr 1, 2, 4, 8 with zero fill
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = sum[e];
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
                sum[e] = v / (float)p.HW * p.gap_mul;
            }
            if (px == 15) {
                float* dst = p.gap + (size_t)img * p.N + ch;
                *reinterpret_cast<float4*>(dst) = make_float4(sum[0], sum[1], sum[2], sum[3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(sum[4], sum[5], sum[6], sum[7]);
            }
        }
    }
}

template <typename T>
int launch_exit_t(const void* in, const void* wp16, const float* scale, const float* bias, void* out, float* gap, int n, int HW,
                  int K, int N, int relu, float gap_mul, hipStream_t s) {
    ExitParams<T> p{};
    p.in = reinterpret_cast<const T*>(in); p.wp16 = reinterpret_cast<const uint4*>(wp16); p.scale = scale; p.bias = bias;
    p.out = reinterpret_cast<T*>(out); p.gap = gap; p.n = n; p.HW = HW; p.K = K; p.N = N; p.relu = relu; p.gap_mul = gap_mul;
    const int grid = n * (N / XN);
    const size_t lds = 2 * XA_BUF;
    if (gap) hipLaunchKernelGGL((exit_gemm_kernel<T, true>), dim3(grid), dim3(256), lds, s, p);
    else hipLaunchKernelGGL((exit_gemm_kernel<T, false>), dim3(grid), dim3(256), lds, s, p);
    return (int)hipGetLastError();
}

}  // namespace

// one image's pixels fit seven 16-row fragments; whole 64-channel chunks (at least two); 256-channel column tiles
bool exit_supported(int dtype, int K, int N, int HW, long long n) {
    return (dtype == 1 || dtype == 2) && HW > 0 && HW <= 16 * XMF && K % XK == 0 && K >= 2 * XK && N % XN == 0 &&
           n > 0 && n * (long long)(N / XN) < (1ll << 30) && n * (long long)HW * (K > N ? K : N) * 2 < (1ll << 40);
}

// out = relu?(BN(in[n * HW][K] x W)) as [n * HW][N] -- or, with gap != nullptr, its mean over each image's HW pixels as fp32
// [n][N] (out is not written).  dtype: 1 = bf16, 2 = f16.
int launch_exit_gemm(int dtype, const void* in, const void* wp16, const float* scale, const float* bias, void* out, float* gap,
                     int n, int HW, int K, int N, int relu, float gap_mul, hipStream_t s) {
    if (!exit_supported(dtype, K, N, HW, n)) return (int)hipErrorInvalidValue;
    return dtype == 2 ? launch_exit_t<f16_t>(in, wp16, scale, bias, out, gap, n, HW, K, N, relu, gap_mul, s)
                      : launch_exit_t<bf16_t>(in, wp16, scale, bias, out, gap, n, HW, K, N, relu, gap_mul, s);
}
