// Two-kernel form of the 728-wide SeparableConv2D layers (bf16):
//   dw3x3_kernel      depthwise 3x3 'same' (+ optional ReLU on the input), NHWC -> NHWC, HBM-bound
//   gemm_tile_kernel  pointwise 1x1 as a tiled MFMA GEMM with folded BN / residual / ReLU epilogue
//
// Why not fused: keeping the whole 96 x 768 output tile of a workgroup in registers (the fused
// kernels) leaves one fat workgroup per CU whose prologue, depthwise stage, weight stream and
// epilogue all serialise against its MFMAs (measured 0.236 ms per layer at n = 256, 17 % of
// the bf16 peak).  Splitting costs one extra round trip of the depthwise result through
// HBM / Infinity Cache (272 MB per layer) but lets the GEMM run 128 x 128 tiles at 3
// workgroups per CU, so one workgroup's loads and stores hide under another's MFMAs.
//
// GEMM: out[M][N] = D[M][K] * W, D row-major with K innermost, W in the host-packed MFMA fragment
// order.  Workgroup = 128 rows x 128 columns, 4 waves as 2 x 2 (64 x 64 each = 2 x 2
// v_mfma_f32_32x32x16_bf16 tiles, 64 accumulator registers).  A streams global -> registers ->
// LDS in 64-deep chunks (double-buffered, one barrier per chunk, rows padded to an odd number
// of 16-byte slots); B fragments come straight from L2 into a register ring (fragment order =
// one coalesced 1 KiB load, no LDS).  Consecutive workgroups on an XCD share the A rows (N-tile
// index fastest).
#include "gemm_common.h"

namespace {
using namespace bqk;

__device__ __forceinline__ unsigned relu2s(unsigned x) { return relu_pk16(x); }

// ------------------------------------------------------------------ depthwise 3x3
// One thread = one 16-byte piece (8 channels) of one image column: it walks down the rows
// with a 3x3 window of vectors in registers, so every output costs 3 new 16-byte loads
// instead of 9.  Lanes run over pieces of a pixel first: a wave touches contiguous memory.
template <typename T, bool RELU>
__global__ void __launch_bounds__(256) dw3x3_kernel(const T* __restrict__ in, const float* __restrict__ dw,
                                                    T* __restrict__ out, int n, int H, int W, int C) {
    if constexpr (H16<T>::F16) bq_f16_saturate();
    const int ppp = C / 8;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)n * W * ppp) return;
    const int piece = (int)(gid % ppp);
    const int x = (int)((gid / ppp) % W);
    const int img = (int)(gid / ((long long)ppp * W));
    float w[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 a = *reinterpret_cast<const float4*>(dw + (size_t)t * C + piece * 8);
        const float4 b = *reinterpret_cast<const float4*>(dw + (size_t)t * C + piece * 8 + 4);
        w[t][0] = a.x; w[t][1] = a.y; w[t][2] = a.z; w[t][3] = a.w;
        w[t][4] = b.x; w[t][5] = b.y; w[t][6] = b.z; w[t][7] = b.w;
    }
    const T* base = in + ((size_t)img * H * W) * C + piece * 8;
    T* obase = out + ((size_t)img * H * W) * C + piece * 8;
    const bool xl = x > 0, xr = x + 1 < W;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    auto load_row = [&](int y, uint4 (&r)[3]) {
        // branch-free: clamp the row, load, then zero what lies outside the image
        const bool vy = y >= 0 && y < H;
        const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
        const T* p = base + ((size_t)yc * W + x) * C;
        r[1] = *reinterpret_cast<const uint4*>(p);
        r[0] = *reinterpret_cast<const uint4*>(xl ? p - C : p);
        r[2] = *reinterpret_cast<const uint4*>(xr ? p + C : p);
        if (!(vy && xl)) r[0] = zero;
        if (!vy) r[1] = zero;
        if (!(vy && xr)) r[2] = zero;
        if (RELU) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                r[k].x = relu2s(r[k].x); r[k].y = relu2s(r[k].y); r[k].z = relu2s(r[k].z); r[k].w = relu2s(r[k].w);
            }
        }
    };
    uint4 r0[3], r1[3], r2[3];
    load_row(-1, r0);
    load_row(0, r1);
    for (int y = 0; y < H; ++y) {
        load_row(y + 1, r2);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            float f[8];
            unpack<T>(r0[dx], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(w[dx][j], f[j], acc[j]);
            unpack<T>(r1[dx], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(w[3 + dx][j], f[j], acc[j]);
            unpack<T>(r2[dx], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(w[6 + dx][j], f[j], acc[j]);
        }
        *reinterpret_cast<uint4*>(obase + ((size_t)y * W + x) * C) = pack<T>(acc);
#pragma unroll
        for (int k = 0; k < 3; ++k) { r0[k] = r1[k]; r1[k] = r2[k]; }
    }
}

// ------------------------------------------------------------------ tiled GEMM
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int A_ROW = BK * 2 + 16;        // 144 B: 9 slots, conflict-free ds_read_b128
constexpr int A_BUF = BM * A_ROW;         // 18 KB
constexpr int ST_ROW = BN * 2 + 16;       // staging row of the output tile

// S2: the rows of A are the even pixels of a larger map (1x1 / stride 2 / 'same' shortcut convolutions): row m = output
// pixel (img, y, x) of an H x W map reads input pixel (img, 2y, 2x) of the Hi x Wi map.
// EPI (round 4: the fusions BASELINE.json's north_star names, so that neither tensor goes through HBM on its own):
//   EPI_GAP   the tile's rows are ONE image's H x W pixels (<= 128; the rest of the 128 MFMA rows is computed and dropped):
//             folded BN + ReLU, rounded to the storage type -- the rounding point of the tensor GlobalAveragePooling2D
//             used to read --, then the column means in the pixel order of the old pooling kernel (bit-identical to it),
//             fp32 [n][ldo] to p.out.  No atomics: one workgroup owns an image's pixels.
//   EPI_POOL  (with S2) out = MaxPool3x3/s2 'same' (y) + BN(conv1x1/s2(x)): the shortcut tile, rounded, waits in LDS and
//             the store pass becomes the pooling pass of kernels_misc.hip (thread = 8 channels of a pooled pixel, nine
//             coalesced 16-byte loads of y = p.residual [n][Hi][Wi][ldo], fp32 max, + shortcut, rounded).  Three workgroups
//             per CU: one's pooling pass runs under the others' MFMAs.
enum { EPI_PLAIN = 0, EPI_GAP = 1, EPI_POOL = 2 };

template <typename T, int PF, bool S2, int EPI = EPI_PLAIN>
__global__ void __launch_bounds__(256) gemm_tile_kernel(const GemmParams p) {
    if constexpr (H16<T>::F16) bq_f16_saturate();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;
    const int ntn = p.NFp / 4;                       // 128-column tiles
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int mt = tile / ntn, nt = tile - mt * ntn;
    const int rows_per_tile = EPI == EPI_GAP ? p.H * p.W : BM;
    const int m0 = mt * rows_per_tile;
    const int K = p.K, KB = K / 16, NC = (K + BK - 1) / BK;
    const T* __restrict__ A = reinterpret_cast<const T*>(p.in);

    // A staging: 4 pieces per thread and chunk; row = idx >> 3, piece = idx & 7
    const int jp = tid & 7;
    size_t arow[4];                                  // element offset of this thread's four A rows (the same for every chunk)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int row = m0 + ((tid + q * 256) >> 3);
        row = row < p.M ? row : p.M - 1;
        if constexpr (S2) {
            const int hw = p.H * p.W;
            const int img = row / hw, rem = row - img * hw;
            const int y = rem / p.W, x = rem - y * p.W;
            arow[q] = ((size_t)(img * p.Hi + 2 * y) * p.Wi + 2 * x) * p.ldi;
        } else {
            arow[q] = (size_t)row * p.ldi;
        }
    }
    auto load_a = [&](int c, uint4 (&r)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k0 = c * BK + jp * 8;
            const int kc = k0 < K ? k0 : 0;           // past K (tail chunk): any valid address, zeroed below
            uint4 v = *reinterpret_cast<const uint4*>(A + arow[q] + kc);
            if (k0 >= K) v = make_uint4(0, 0, 0, 0);
            r[q] = v;
        }
    };
    auto store_a = [&](int buf, const uint4 (&r)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<uint4*>(smem + buf * A_BUF + ((tid + q * 256) >> 3) * A_ROW + jp * 16) = r[q];
    };

    uint4 areg[4];
    load_a(0, areg);
    store_a(0, areg);
    load_a(1, areg);

    const int nfb = nt * 4 + wn * 2;                 // this wave's two 32-column fragments
    const uint4* __restrict__ wp = reinterpret_cast<const uint4*>(p.wp);
    const uint4* bp0 = wp + ((size_t)nfb * p.KBtot + p.kb0) * 64 + lane;
    uint4 bq[PF][2];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int idx = d < KB ? d : KB - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) bq[d][j] = bp0[((size_t)j * p.KBtot + idx) * 64];
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    __syncthreads();

    for (int c = 0; c < NC; ++c) {
        const int a_base = (c & 1) * A_BUF + (wm * 64 + r32) * A_ROW + h * 16;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int kb = c * 4 + d;
            uint4 a[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[i] = *reinterpret_cast<const uint4*>(smem + a_base + i * 32 * A_ROW + d * 32);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma<T>(acc[i][j], bq[d % PF][j], a[i]);
            const int nx = kb + PF;
            const int idx = nx < KB ? nx : KB - 1;   // clamped: blocks past K meet zero A columns
#pragma unroll
            for (int j = 0; j < 2; ++j) bq[d % PF][j] = bp0[((size_t)j * p.KBtot + idx) * 64];
        }
        // chunk c+1 (in registers since the previous iteration) -> the other buffer, whose last
        // readers finished before the previous barrier; then start loading chunk c+2
        if (c + 1 < NC) store_a((c + 1) & 1, areg);
        if (c + 2 < NC) load_a(c + 2, areg);
        __syncthreads();
    }

    // ---- epilogue: folded BN, residual, ReLU in registers -> LDS -> whole 256-byte row segments
    const T* __restrict__ res = EPI == EPI_PLAIN ? reinterpret_cast<const T*>(p.residual) : nullptr;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = (wn * 2 + j) * 32 + g * 8 + h * 4;      // column inside the tile
            const int n0 = nt * BN + nl;
            float sc[4] = {1.f, 1.f, 1.f, 1.f}, bi[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.scale) {
                const float4 t = *reinterpret_cast<const float4*>(p.scale + n0);
                sc[0] = t.x; sc[1] = t.y; sc[2] = t.z; sc[3] = t.w;
            }
            if (p.bias) {
                const float4 t = *reinterpret_cast<const float4*>(p.bias + n0);
                bi[0] = t.x; bi[1] = t.y; bi[2] = t.z; bi[3] = t.w;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rl = wm * 64 + i * 32 + r32;
                const int m = m0 + rl;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[i][j][g * 4 + e], sc[e], bi[e]);
                if (res && m < p.M && n0 < p.Nstore) {
                    float rv[4];
                    load4<T>(res + (size_t)m * p.ldo + n0, rv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rv[e];
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                store4<T>(reinterpret_cast<T*>(smem + (size_t)rl * ST_ROW) + nl, v);
            }
        }
    __syncthreads();
    if constexpr (EPI == EPI_GAP) {
        // column means over the image's pixels, in pixel order (the summation order of gap_kernel)
        if (tid < BN) {
            const int ncol = nt * BN + tid;
            if (ncol < p.Nstore) {
                float sum = 0.f;
                for (int r = 0; r < rows_per_tile; ++r)
                    sum += (float)*reinterpret_cast<const T*>(smem + (size_t)r * ST_ROW + tid * 2);
                reinterpret_cast<float*>(p.out)[(size_t)mt * p.ldo + ncol] = sum / (float)rows_per_tile * (p.gap_mul != 0.f ? p.gap_mul : 1.f);
            }
        }
    } else if constexpr (EPI == EPI_POOL) {
        const T* __restrict__ y = reinterpret_cast<const T*>(p.residual);
        T* __restrict__ out = reinterpret_cast<T*>(p.out);
        const int pc = tid & 15;
        const int ncol = nt * BN + pc * 8;
        const int pt = (p.Hi & 1) ? 1 : 0, pl = (p.Wi & 1) ? 1 : 0;   // TensorFlow 'same' padding of the pool: (1,1) odd, (0,1) even
        if (ncol < p.Nstore) {
            // Branch-free taps: a tap outside the map is clamped onto the map's edge, which lies inside the same window (a
            // duplicate changes no maximum), so all nine 16-byte loads of a pixel -- and of the next pixel: two rows per
            // step -- are in flight together.  (First form, taps under `if`, one row at a time: 0.36 ms for block 4 against
            // 0.26 ms for the two kernels it replaces.)
            const int hw = p.H * p.W;
            constexpr int RU = 2;
            for (int r0 = tid >> 4; r0 < BM; r0 += 16 * RU) {
                uint4 u[RU][9];
                int mrow[RU];
#pragma unroll
                for (int k = 0; k < RU; ++k) {
                    int m = m0 + r0 + 16 * k;
                    mrow[k] = m;
                    m = m < p.M ? m : p.M - 1;
                    const int img = m / hw, rem = m - img * hw;
                    const int yo = rem / p.W, xo = rem - yo * p.W;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        int yy = 2 * yo + dy - pt;
                        yy = yy < 0 ? 0 : (yy > p.Hi - 1 ? p.Hi - 1 : yy);
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            int xx = 2 * xo + dx - pl;
                            xx = xx < 0 ? 0 : (xx > p.Wi - 1 ? p.Wi - 1 : xx);
                            u[k][dy * 3 + dx] = *reinterpret_cast<const uint4*>(y + ((size_t)(img * p.Hi + yy) * p.Wi + xx) * p.ldo + ncol);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < RU; ++k) {
                    const int r = r0 + 16 * k;
                    if (r >= BM || mrow[k] >= p.M) continue;
                    float mx[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) mx[j] = -INFINITY;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const T* e = reinterpret_cast<const T*>(&u[k][t]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) mx[j] = fmaxf(mx[j], (float)e[j]);
                    }
                    const uint4 ru = *reinterpret_cast<const uint4*>(smem + (size_t)r * ST_ROW + pc * 16);
                    const T* re = reinterpret_cast<const T*>(&ru);
                    uint4 ou;
                    T* oe = reinterpret_cast<T*>(&ou);
#pragma unroll
                    for (int j = 0; j < 8; ++j) oe[j] = (T)(mx[j] + (float)re[j]);
                    *reinterpret_cast<uint4*>(out + (size_t)mrow[k] * p.ldo + ncol) = ou;
                }
            }
        }
    } else {
        unsigned char* __restrict__ out = reinterpret_cast<unsigned char*>(p.out);
        const int pc = tid & 15;                                   // 16-byte piece of the 256-byte row segment
        const int ncol = nt * BN + pc * 8;
        if (ncol < p.Nstore) {
            for (int r = tid >> 4; r < BM; r += 16) {
                const int m = m0 + r;
                if (m < p.M)
                    *reinterpret_cast<uint4*>(out + ((size_t)m * p.ldo + ncol) * 2) =
                        *reinterpret_cast<const uint4*>(smem + (size_t)r * ST_ROW + pc * 16);
            }
        }
    }
}

}  // namespace

namespace {
template <typename T>
int launch_dw3x3_t(const void* in, const float* dw, void* out, int n, int H, int W, int C, int relu, hipStream_t s) {
    const long long total = (long long)n * W * (C / 8);
    const int grid = (int)((total + 255) / 256);
    if (relu)
        hipLaunchKernelGGL((dw3x3_kernel<T, true>), dim3(grid), dim3(256), 0, s, (const T*)in, dw, (T*)out, n, H, W, C);
    else
        hipLaunchKernelGGL((dw3x3_kernel<T, false>), dim3(grid), dim3(256), 0, s, (const T*)in, dw, (T*)out, n, H, W, C);
    return (int)hipGetLastError();
}
}  // namespace

// dtype: 1 = bf16, 2 = f16 (the two-kernel form only exists for the 16-bit types)
int launch_dw3x3(int dtype, const void* in, const float* dw, void* out, int n, int H, int W, int C, int relu, hipStream_t s) {
    return dtype == 2 ? launch_dw3x3_t<f16_t>(in, dw, out, n, H, W, C, relu, s)
                      : launch_dw3x3_t<bf16_t>(in, dw, out, n, H, W, C, relu, s);
}

// p.in = depthwise result [M][ldi] -- or, with s2, the map [n][Hi][Wi][ldi] whose even pixels are the rows;
// p.NFp multiple of 4; 16-bit types only.  epi: 0 plain; 1 = global average pool (p.out = fp32 [M / (H W)][ldo], H W <= 128,
// M a multiple of H W, no s2); 2 = max-pool + add (s2 only: p.residual = the map to pool, [n][Hi][Wi][ldo])
int launch_gemm_tile(int dtype, const GemmParams& p, bool s2, hipStream_t s, int epi) {
    if (p.NFp % 4 != 0 || p.K % 16 != 0) return (int)hipErrorInvalidValue;
    const size_t lds = 2 * A_BUF > BM * ST_ROW ? 2 * A_BUF : BM * ST_ROW;
    if (epi == EPI_GAP) {
        const int hw = p.H * p.W;
        if (s2 || hw <= 0 || hw > BM || p.M % hw) return (int)hipErrorInvalidValue;
        const int grid = (p.M / hw) * (p.NFp / 4);
        if (dtype == 2) hipLaunchKernelGGL((gemm_tile_kernel<f16_t, 4, false, EPI_GAP>), dim3(grid), dim3(256), lds, s, p);
        else hipLaunchKernelGGL((gemm_tile_kernel<bf16_t, 4, false, EPI_GAP>), dim3(grid), dim3(256), lds, s, p);
        return (int)hipGetLastError();
    }
    if (epi == EPI_POOL) {
        if (!s2 || !p.residual) return (int)hipErrorInvalidValue;
        const int grid = ((p.M + BM - 1) / BM) * (p.NFp / 4);
        if (dtype == 2) hipLaunchKernelGGL((gemm_tile_kernel<f16_t, 4, true, EPI_POOL>), dim3(grid), dim3(256), lds, s, p);
        else hipLaunchKernelGGL((gemm_tile_kernel<bf16_t, 4, true, EPI_POOL>), dim3(grid), dim3(256), lds, s, p);
        return (int)hipGetLastError();
    }
    const int grid = ((p.M + BM - 1) / BM) * (p.NFp / 4);
    if (dtype == 2) {
        if (s2) hipLaunchKernelGGL((gemm_tile_kernel<f16_t, 4, true>), dim3(grid), dim3(256), lds, s, p);
        else hipLaunchKernelGGL((gemm_tile_kernel<f16_t, 4, false>), dim3(grid), dim3(256), lds, s, p);
    } else {
        if (s2) hipLaunchKernelGGL((gemm_tile_kernel<bf16_t, 4, true>), dim3(grid), dim3(256), lds, s, p);
        else hipLaunchKernelGGL((gemm_tile_kernel<bf16_t, 4, false>), dim3(grid), dim3(256), lds, s, p);
    }
    return (int)hipGetLastError();
}
