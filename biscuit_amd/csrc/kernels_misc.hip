// HBM-bound kernels around the MFMA path: tile staging (K0), the 3->32 stem conv (K1a),
// max-pool + residual add (K4), global average pool (K5), the MC head's last layer with
// softmax + Welford fold (K6 tail) and the slide-level segmented reduce (K7).
#include "bq_common.h"

namespace {

template <typename T> __device__ __forceinline__ T from_f32(float f);
template <> __device__ __forceinline__ float from_f32<float>(float f) { return f; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float f) { return (bf16_t)f; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float f) { return (f16_t)f; }   // FP16_OVFL: saturates
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <> __device__ __forceinline__ float to_f32<f16_t>(f16_t v) { return (float)v; }

template <typename T>
__device__ __forceinline__ unsigned pk16(float lo, float hi) {
    const unsigned short a = __builtin_bit_cast(unsigned short, from_f32<T>(lo));
    const unsigned short b = __builtin_bit_cast(unsigned short, from_f32<T>(hi));
    return (unsigned)a | ((unsigned)b << 16);
}
template <typename T> __device__ __forceinline__ void f16_mode() {
    if constexpr (sizeof(T) == 2 && !__is_same(T, bf16_t)) bq_f16_saturate();
}

// ---------------------------------------------------------------- K0 staging
// One workgroup per tile.  Pass 1: exact integer sum / sum of squares of the uint8
// bytes (order-independent, so the statistics are bit-reproducible), pass 2 re-reads the
// (now L2-resident) 268 KB tile, standardises and writes three contiguous planes.
template <typename T>
__global__ void __launch_bounds__(512) stage_u8_kernel(const uint8_t* __restrict__ tiles, int px,
                                                       T* __restrict__ out) {
    f16_mode<T>();
    const int npix = px * px;
    const int nbytes = npix * 3;
    const uint8_t* src = tiles + (size_t)blockIdx.x * nbytes;
    const int tid = threadIdx.x, nt = blockDim.x;

    unsigned long long s1 = 0, s2 = 0;
    // aligned 4-byte body, scalar head/tail
    const int head = (int)((4 - ((uintptr_t)src & 3)) & 3);
    const int body = (nbytes - head) >> 2;
    if (tid < head) { const unsigned v = src[tid]; s1 += v; s2 += v * v; }
    const unsigned* w = reinterpret_cast<const unsigned*>(src + head);
    for (int i = tid; i < body; i += nt) {
        const unsigned u = w[i];
        const unsigned a = u & 255u, b = (u >> 8) & 255u, c = (u >> 16) & 255u, d = u >> 24;
        s1 += a + b + c + d;
        s2 += a * a + b * b + c * c + d * d;
    }
    const int tail0 = head + body * 4;
    if (tid < nbytes - tail0) { const unsigned v = src[tail0 + tid]; s1 += v; s2 += v * v; }

    __shared__ unsigned long long red[2][8];
    __shared__ float stat[2];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s1; red[1][tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
        unsigned long long a = 0, b = 0;
        for (int i = 0; i < nt / 64; ++i) { a += red[0][i]; b += red[1][i]; }
        const double n = (double)nbytes;
        const double mean = (double)a / n;
        double var = (double)b / n - mean * mean;
        if (var < 0) var = 0;
        const double sd = sqrt(var);
        const double floor_sd = 1.0 / sqrt(n);  // tf.image.per_image_standardization
        stat[0] = (float)mean;
        stat[1] = (float)(1.0 / (sd > floor_sd ? sd : floor_sd));
    }
    __syncthreads();
    const float mean = stat[0], inv = stat[1];
    T* o0 = out + (size_t)blockIdx.x * nbytes;
    for (int p = tid; p < npix; p += nt) {
        const uint8_t* q = src + p * 3;
        o0[p] = from_f32<T>(((float)q[0] - mean) * inv);
        o0[npix + p] = from_f32<T>(((float)q[1] - mean) * inv);
        o0[2 * npix + p] = from_f32<T>(((float)q[2] - mean) * inv);
    }
}

// Two-kernel form of K0 with SLICES workgroups per tile (one workgroup per tile leaves the chip at one
// workgroup per CU for an HBM-bound pass: 0.157 ms per 256 tiles against a 0.04 ms roofline): integer sums
// into two 64-bit atomics per tile (exact and order-independent, like the one-kernel form), then the
// standardisation pass.
constexpr int kStageSlices = 8;

__global__ void __launch_bounds__(256) stage_stats_kernel(const uint8_t* __restrict__ tiles, int px,
                                                          unsigned long long* __restrict__ stats) {
    const int nbytes = px * px * 3;
    const int tile = blockIdx.x / kStageSlices, sl = blockIdx.x - tile * kStageSlices;
    const uint8_t* src = tiles + (size_t)tile * nbytes;
    const int tid = threadIdx.x, nt = blockDim.x;
    // A thread's share of a slice is ~130 bytes: 32-bit partial sums cannot overflow (130 x 255^2 < 2^24).  Four bytes per
    // instruction: v_sad_u8 against zero is their sum, v_dot4_u32_u8 of a word with itself the sum of their squares (round 4:
    // 16-byte loads and these two instead of shifts, masks and 64-bit adds per byte: 0.036 -> 0.02 ms per batch, same integers).
    unsigned p1 = 0, p2 = 0;
    const int head = (int)((4 - ((uintptr_t)src & 3)) & 3);
    const int body = (nbytes - head) >> 2;
    const int d0 = (int)((long long)body * sl / kStageSlices), d1 = (int)((long long)body * (sl + 1) / kStageSlices);
    if (sl == 0 && tid < head) { const unsigned v = src[tid]; p1 += v; p2 += v * v; }
    const unsigned* w = reinterpret_cast<const unsigned*>(src + head);
    const int nq = (d1 - d0) >> 2;                      // whole groups of four words (16 bytes: dword alignment is all it needs)
    for (int i = tid; i < nq; i += nt) {
        const uint4 u = *reinterpret_cast<const uint4*>(w + d0 + 4 * i);
        p1 = __builtin_amdgcn_sad_u8(u.x, 0u, p1); p2 = __builtin_amdgcn_udot4(u.x, u.x, p2, false);
        p1 = __builtin_amdgcn_sad_u8(u.y, 0u, p1); p2 = __builtin_amdgcn_udot4(u.y, u.y, p2, false);
        p1 = __builtin_amdgcn_sad_u8(u.z, 0u, p1); p2 = __builtin_amdgcn_udot4(u.z, u.z, p2, false);
        p1 = __builtin_amdgcn_sad_u8(u.w, 0u, p1); p2 = __builtin_amdgcn_udot4(u.w, u.w, p2, false);
    }
    for (int i = d0 + 4 * nq + tid; i < d1; i += nt) {
        const unsigned u = w[i];
        p1 = __builtin_amdgcn_sad_u8(u, 0u, p1); p2 = __builtin_amdgcn_udot4(u, u, p2, false);
    }
    const int tail0 = head + body * 4;
    if (sl == kStageSlices - 1 && tid < nbytes - tail0) { const unsigned v = src[tail0 + tid]; p1 += v; p2 += v * v; }
    unsigned long long s1 = p1, s2 = p2;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if ((tid & 63) == 0) {
        atomicAdd(&stats[2 * tile], s1);
        atomicAdd(&stats[2 * tile + 1], s2);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) stage_apply_kernel(const uint8_t* __restrict__ tiles, int px,
                                                          const unsigned long long* __restrict__ stats,
                                                          T* __restrict__ out) {
    f16_mode<T>();
    const int npix = px * px, nbytes = npix * 3;
    const int tile = blockIdx.x / kStageSlices, sl = blockIdx.x - tile * kStageSlices;
    const uint8_t* src = tiles + (size_t)tile * nbytes;
    // same arithmetic as the one-kernel form: float64 statistics from the exact integer sums
    const double n = (double)nbytes;
    const double mean_d = (double)stats[2 * tile] / n;
    double var = (double)stats[2 * tile + 1] / n - mean_d * mean_d;
    if (var < 0) var = 0;
    const double sd = sqrt(var), floor_sd = 1.0 / sqrt(n);   // tf.image.per_image_standardization
    const float mean = (float)mean_d, inv = (float)(1.0 / (sd > floor_sd ? sd : floor_sd));
    T* o0 = out + (size_t)tile * nbytes;
    const int p0 = (int)((long long)npix * sl / kStageSlices), p1 = (int)((long long)npix * (sl + 1) / kStageSlices);
    for (int p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
        const uint8_t* q = src + p * 3;
        o0[p] = from_f32<T>(((float)q[0] - mean) * inv);
        o0[npix + p] = from_f32<T>(((float)q[1] - mean) * inv);
        o0[2 * npix + p] = from_f32<T>(((float)q[2] - mean) * inv);
    }
}

template <typename T>
__global__ void stage_f32_kernel(const float* __restrict__ tiles, long long total_pix, int npix,
                                 T* __restrict__ out) {
    f16_mode<T>();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_pix) return;
    const long long img = i / npix;
    const int p = (int)(i - img * npix);
    const float* q = tiles + i * 3;
    T* o0 = out + img * 3 * npix;
    o0[p] = from_f32<T>(q[0]);
    o0[npix + p] = from_f32<T>(q[1]);
    o0[2 * npix + p] = from_f32<T>(q[2]);
}

// ---------------------------------------------------------------- K1a stem conv1
// Conv2D(32, 3x3, strides 2, 'valid', no bias) + folded BN + ReLU on the planar staged
// tile.  0.2 % of the FLOPs: vector ALU, one output pixel x 32 channels per lane, weights
// through the scalar cache (uniform addresses), NHWC store of 32 contiguous channels.
template <typename T>
__global__ void __launch_bounds__(256) stem1_kernel(const T* __restrict__ in, int n, int px,
                                                    const float* __restrict__ w,
                                                    const float* __restrict__ scale,
                                                    const float* __restrict__ bias,
                                                    T* __restrict__ out) {
    f16_mode<T>();
    const int po = (px - 3) / 2 + 1;  // 149
    const long long total = (long long)n * po * po;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int img = (int)(gid / (po * po));
    const int rem = (int)(gid - (long long)img * po * po);
    const int yo = rem / po, xo = rem - yo * po;
    const T* base = in + (size_t)img * 3 * px * px;
    float x[27];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                x[(dy * 3 + dx) * 3 + c] = to_f32<T>(base[(size_t)c * px * px + (2 * yo + dy) * px + 2 * xo + dx]);
    // two output channels per v_pk_fma_f32 (weights as scalar-register pairs, the pixel value splat): the
    // kernel is bound by its 864 multiply-adds per pixel, not by memory
    typedef float f32x2s __attribute__((ext_vector_type(2)));
    f32x2s acc2[16];
#pragma unroll
    for (int co = 0; co < 16; ++co) acc2[co] = (f32x2s){0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const f32x2s xs = {x[k], x[k]};
#pragma unroll
        for (int co = 0; co < 16; ++co)
            acc2[co] = __builtin_elementwise_fma(xs, (f32x2s){w[k * 32 + 2 * co], w[k * 32 + 2 * co + 1]}, acc2[co]);
    }
    float acc[32];
#pragma unroll
    for (int co = 0; co < 16; ++co) {
        acc[2 * co] = fmaxf(fmaf(acc2[co].x, scale[2 * co], bias[2 * co]), 0.f);
        acc[2 * co + 1] = fmaxf(fmaf(acc2[co].y, scale[2 * co + 1], bias[2 * co + 1]), 0.f);
    }
    T* o = out + (size_t)gid * 32;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int j = 0; j < 32; j += 8)
            *reinterpret_cast<uint4*>(o + j) =
                make_uint4(pk16<T>(acc[j], acc[j + 1]), pk16<T>(acc[j + 2], acc[j + 3]),
                           pk16<T>(acc[j + 4], acc[j + 5]), pk16<T>(acc[j + 6], acc[j + 7]));
    } else {
#pragma unroll
        for (int j = 0; j < 32; j += 4)
            *reinterpret_cast<float4*>(o + j) = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
    }
}

// ---------------------------------------------------------------- K4 maxpool + add
// MaxPooling2D(3, strides 2, 'same') with TensorFlow's asymmetric padding (odd input:
// pad (1,1); even input: pad (0,1); padded value -inf) fused with the residual add.
template <typename T, int VEC>
__global__ void __launch_bounds__(256) pool_add_kernel(const T* __restrict__ y, const T* __restrict__ res,
                                                       T* __restrict__ out, int n, int Hi, int Wi,
                                                       int Ho, int Wo, int C) {
    f16_mode<T>();
    const int chunks = C / VEC;
    const long long total = (long long)n * Ho * Wo * chunks;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c = (int)(gid % chunks);
    const long long pix = gid / chunks;
    const int xo = (int)(pix % Wo);
    const int yo = (int)((pix / Wo) % Ho);
    const int img = (int)(pix / ((long long)Wo * Ho));
    const int pt = (Hi & 1) ? 1 : 0, pl = (Wi & 1) ? 1 : 0;
    float m[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) m[j] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = 2 * yo + dy - pt;
        if ((unsigned)yy >= (unsigned)Hi) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = 2 * xo + dx - pl;
            if ((unsigned)xx >= (unsigned)Wi) continue;
            const T* q = y + ((size_t)(img * Hi + yy) * Wi + xx) * C + c * VEC;
            const uint4 u = *reinterpret_cast<const uint4*>(q);
            const T* e = reinterpret_cast<const T*>(&u);
#pragma unroll
            for (int j = 0; j < VEC; ++j) m[j] = fmaxf(m[j], to_f32<T>(e[j]));
        }
    }
    const size_t o = (size_t)pix * C + c * VEC;
    const uint4 ru = *reinterpret_cast<const uint4*>(res + o);
    const T* re = reinterpret_cast<const T*>(&ru);
    uint4 ou;
    T* oe = reinterpret_cast<T*>(&ou);
#pragma unroll
    for (int j = 0; j < VEC; ++j) oe[j] = from_f32<T>(m[j] + to_f32<T>(re[j]));
    *reinterpret_cast<uint4*>(out + o) = ou;
}

// ---------------------------------------------------------------- K5 global average pool
template <typename T>
__global__ void __launch_bounds__(256) gap_kernel(const T* __restrict__ x, int n, int HW, int C, int ld,
                                                  float* __restrict__ feat, float mul) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)n * C) return;
    const int img = (int)(gid / C), c = (int)(gid - (long long)img * C);
    const T* p = x + (size_t)img * HW * ld + c;
    float s = 0.f;
    for (int i = 0; i < HW; ++i) s += to_f32<T>(p[(size_t)i * ld]);
    feat[gid] = s / (float)HW * mul;
}

// ---------------------------------------------------------------- K6 tail
// One workgroup of eight waves per tile.  Wave w takes MC passes w, w + 8, ...: dropout(hidden_1 row) . W2 -> 2 logits ->
// softmax; the two probabilities of a block of eight passes meet in LDS and ONE lane folds them into the Welford state (count,
// mean[2], M2[2]) in pass order -- the arithmetic, and its order, of the one-wave-per-tile form this replaces (round 4: that
// form ran 256 waves on the whole chip, each through 30 passes of ~500 vector instructions: 0.036 ms; the passes are
// independent, only the fold is sequential).  W2 lives in registers (16 k per lane x 2 classes).  state layout [n][5] fp32.
constexpr int kHeadFinalWaves = 8;
__global__ void __launch_bounds__(64 * kHeadFinalWaves) head_final_kernel(const float* __restrict__ h1, int n, int mc_n,
                                                         int pass0, long long tile0_imm,
                                                         const long long* __restrict__ tile0_dev,
                                                         const long long* __restrict__ tile_idx, unsigned seed_lo,
                                                         unsigned seed_hi, unsigned thresh, float dscale,
                                                         const float* __restrict__ w2,
                                                         const float* __restrict__ b2, int init,
                                                         int finalize, float* __restrict__ state,
                                                         float* __restrict__ mean2,
                                                         float* __restrict__ std2) {
    __shared__ float probs[kHeadFinalWaves][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x;
    const unsigned tctr = (unsigned)(tile0_imm + (tile0_dev ? *tile0_dev : 0) + (tile_idx ? tile_idx[tile] : (long long)tile));
    float wa[16], wb[16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 4 * (lane + 64 * j) + e;
            wa[j * 4 + e] = w2[k * 2];
            wb[j * 4 + e] = w2[k * 2 + 1];
        }
    float cnt = 0.f, mu0 = 0.f, mu1 = 0.f, q0 = 0.f, q1 = 0.f;       // (kept by thread 0 only)
    if (!init && threadIdx.x == 0) {
        const float* st = state + (size_t)tile * 5;
        cnt = st[0]; mu0 = st[1]; mu1 = st[2]; q0 = st[3]; q1 = st[4];
    }
    for (int base = 0; base < mc_n; base += kHeadFinalWaves) {
        const int p = base + wave;
        if (p < mc_n) {
            const float* row = h1 + ((size_t)tile * mc_n + p) * 1024;
            float z0 = 0.f, z1 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g = lane + 64 * j;  // Philox group = unit / 4
                const float4 v = *reinterpret_cast<const float4*>(row + 4 * g);
                unsigned r[4];
                philox4x32_10((unsigned)g, 2u, (unsigned)(pass0 + p), tctr, seed_lo,
                              seed_hi, r);
                const float f0 = r[0] >= thresh ? v.x * dscale : 0.f;
                const float f1 = r[1] >= thresh ? v.y * dscale : 0.f;
                const float f2 = r[2] >= thresh ? v.z * dscale : 0.f;
                const float f3 = r[3] >= thresh ? v.w * dscale : 0.f;
                z0 = fmaf(f0, wa[j * 4], z0); z0 = fmaf(f1, wa[j * 4 + 1], z0);
                z0 = fmaf(f2, wa[j * 4 + 2], z0); z0 = fmaf(f3, wa[j * 4 + 3], z0);
                z1 = fmaf(f0, wb[j * 4], z1); z1 = fmaf(f1, wb[j * 4 + 1], z1);
                z1 = fmaf(f2, wb[j * 4 + 2], z1); z1 = fmaf(f3, wb[j * 4 + 3], z1);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                z0 += __shfl_xor(z0, o);
                z1 += __shfl_xor(z1, o);
            }
            z0 += b2[0];
            z1 += b2[1];
            const float zm = fmaxf(z0, z1);
            const float e0 = expf(z0 - zm), e1 = expf(z1 - zm);
            const float inv = 1.f / (e0 + e1);
            if (lane == 0) { probs[wave][0] = e0 * inv; probs[wave][1] = e1 * inv; }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int np = mc_n - base < kHeadFinalWaves ? mc_n - base : kHeadFinalWaves;
            for (int k = 0; k < np; ++k) {
                float p0 = probs[k][0], p1 = probs[k][1];
                // Pin the rounded probabilities: if the compiler contracts a product into the Welford differences below
                // (fma(e, inv, -mean)), p - mean keeps the product's rounding error instead of an exact 0 and a single pass
                // reports std ~1e-4 instead of 0.
                asm volatile("" : "+v"(p0), "+v"(p1));
                cnt += 1.f;
                const float d0 = p0 - mu0, d1 = p1 - mu1;
                mu0 += d0 / cnt;
                mu1 += d1 / cnt;
                q0 = fmaf(d0, p0 - mu0, q0);
                q1 = fmaf(d1, p1 - mu1, q1);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float* st = state + (size_t)tile * 5;
        st[0] = cnt; st[1] = mu0; st[2] = mu1; st[3] = q0; st[4] = q1;
        if (finalize) {
            mean2[tile * 2] = mu0;
            mean2[tile * 2 + 1] = mu1;
            std2[tile * 2] = sqrtf(fmaxf(q0, 0.f) / cnt);      // population std (ddof = 0)
            std2[tile * 2 + 1] = sqrtf(fmaxf(q1, 0.f) / cnt);
        }
    }
}

// ---------------------------------------------------------------- K7 slide reduce
// Fixed-point (2^-40) 64-bit integer atomics: the sum is exact per addend and associative,
// so slide means are bit-reproducible whatever the arrival order.
__global__ void __launch_bounds__(256) slide_reduce_kernel(const float* __restrict__ mean2,
                                                           const float* __restrict__ std2,
                                                           const int32_t* __restrict__ slide_idx, int n,
                                                           int n_slides, float tile_uq, int use_uq,
                                                           unsigned long long* acc_pred,
                                                           unsigned long long* acc_unc, int32_t* count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = slide_idx[i];
    if ((unsigned)s >= (unsigned)n_slides) return;
    const float yp = mean2[i * 2 + 1];  // P(class 1): utils.py:27-28 y_pred1
    const float un = std2[i * 2 + 1];   // utils.py:19-20 uncertainty1
    if (use_uq && !(un < tile_uq)) return;  // strict '<' (threshold.py:298)
    const double sc = (double)(1ull << BQ_FIXED_SHIFT);
    const long long a = __double2ll_rn((double)yp * sc);
    const long long b = __double2ll_rn((double)un * sc);
    atomicAdd(acc_pred + s, (unsigned long long)a);
    atomicAdd(acc_unc + s, (unsigned long long)b);
    atomicAdd(count + s, 1);
}

__global__ void slide_finish_kernel(const long long* acc_pred, const long long* acc_unc,
                                    const int32_t* count, int n_slides, double* mean_pred,
                                    double* mean_unc) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slides) return;
    const double inv = 1.0 / (double)(1ull << BQ_FIXED_SHIFT);
    const int c = count[s];
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    mean_pred[s] = c > 0 ? ((double)acc_pred[s] * inv) / (double)c : nan;
    mean_unc[s] = c > 0 ? ((double)acc_unc[s] * inv) / (double)c : nan;
}

// ---------------------------------------------------------------- debug copies
template <typename T>
__global__ void to_f32_nhwc_kernel(const T* __restrict__ x, long long rows, int C, int ld,
                                   float* __restrict__ out) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rows * C) return;
    const long long r = gid / C;
    const int c = (int)(gid - r * C);
    out[gid] = to_f32<T>(x[r * ld + c]);
}

template <typename T>
__global__ void nchw_to_f32_nhwc_kernel(const T* __restrict__ x, int n, int C, int HW,
                                        float* __restrict__ out) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)n * C * HW) return;
    const int c = (int)(gid % C);
    const long long pix = gid / C;
    const int p = (int)(pix % HW);
    const long long img = pix / HW;
    out[gid] = to_f32<T>(x[(img * C + c) * HW + p]);
}

inline int grid_for(long long total, int block) { return (int)((total + block - 1) / block); }

}  // namespace

// CALL is written for the type name T_; instantiated for bf16 (dtype 1), f16 (2) and fp32 (0)
#define BQ_DISPATCH_T(dtype, ...)                                              \
    do {                                                                       \
        if ((dtype) == 1) { using T_ = bf16_t; __VA_ARGS__; }                  \
        else if ((dtype) == 2) { using T_ = f16_t; __VA_ARGS__; }              \
        else { using T_ = float; __VA_ARGS__; }                                \
    } while (0)

int launch_stage_u8(const uint8_t* tiles, int n, int px, void* out, int dtype, double* stats_scratch, hipStream_t s) {
    if (n <= 0) return 0;
    if (stats_scratch) {          // 2 x 64-bit per tile, zeroed here
        unsigned long long* st = reinterpret_cast<unsigned long long*>(stats_scratch);
        hipError_t e = hipMemsetAsync(st, 0, (size_t)n * 16, s);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(stage_stats_kernel, dim3(n * kStageSlices), dim3(256), 0, s, tiles, px, st);
        BQ_DISPATCH_T(dtype,
                      hipLaunchKernelGGL(stage_apply_kernel<T_>, dim3(n * kStageSlices), dim3(256), 0, s, tiles, px,
                                         st, (T_*)out));
        return (int)hipGetLastError();
    }
    BQ_DISPATCH_T(dtype,
                  hipLaunchKernelGGL(stage_u8_kernel<T_>, dim3(n), dim3(512), 0, s, tiles, px, (T_*)out));
    return (int)hipGetLastError();
}

// the per-tile integer sums alone (2 x 64-bit per tile, zeroed here): the pre-pass of the fused front kernel (kernels_front.hip)
int launch_stage_stats(const uint8_t* tiles, int n, int px, double* stats_scratch, hipStream_t s) {
    if (n <= 0) return 0;
    unsigned long long* st = reinterpret_cast<unsigned long long*>(stats_scratch);
    hipError_t e = hipMemsetAsync(st, 0, (size_t)n * 16, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(stage_stats_kernel, dim3(n * kStageSlices), dim3(256), 0, s, tiles, px, st);
    return (int)hipGetLastError();
}

int launch_stage_f32(const float* tiles, int n, int px, void* out, int dtype, hipStream_t s) {
    if (n <= 0) return 0;
    const long long total = (long long)n * px * px;
    BQ_DISPATCH_T(dtype,
                  hipLaunchKernelGGL(stage_f32_kernel<T_>, dim3(grid_for(total, 256)), dim3(256), 0, s,
                                     tiles, total, px * px, (T_*)out));
    return (int)hipGetLastError();
}

int launch_stem1(const void* in, int n, const float* w, const float* scale, const float* bias, void* out,
                 int dtype, hipStream_t s) {
    const int px = 299, po = 149;
    const long long total = (long long)n * po * po;
    BQ_DISPATCH_T(dtype,
                  hipLaunchKernelGGL(stem1_kernel<T_>, dim3(grid_for(total, 256)), dim3(256), 0, s,
                                     (const T_*)in, n, px, w, scale, bias, (T_*)out));
    return (int)hipGetLastError();
}

int launch_pool_add(const void* y, const void* res, void* out, int n, int Hi, int Wi, int C, int dtype,
                    hipStream_t s) {
    const int Ho = (Hi + 1) / 2, Wo = (Wi + 1) / 2;
    if (dtype == 1) {
        const long long total = (long long)n * Ho * Wo * (C / 8);
        hipLaunchKernelGGL((pool_add_kernel<bf16_t, 8>), dim3(grid_for(total, 256)), dim3(256), 0, s,
                           (const bf16_t*)y, (const bf16_t*)res, (bf16_t*)out, n, Hi, Wi, Ho, Wo, C);
    } else if (dtype == 2) {
        const long long total = (long long)n * Ho * Wo * (C / 8);
        hipLaunchKernelGGL((pool_add_kernel<f16_t, 8>), dim3(grid_for(total, 256)), dim3(256), 0, s,
                           (const f16_t*)y, (const f16_t*)res, (f16_t*)out, n, Hi, Wi, Ho, Wo, C);
    } else {
        const long long total = (long long)n * Ho * Wo * (C / 4);
        hipLaunchKernelGGL((pool_add_kernel<float, 4>), dim3(grid_for(total, 256)), dim3(256), 0, s,
                           (const float*)y, (const float*)res, (float*)out, n, Hi, Wi, Ho, Wo, C);
    }
    return (int)hipGetLastError();
}

int launch_gap(const void* x, int n, int HW, int C, int ld, float* feat, float mul, int dtype, hipStream_t s) {
    const long long total = (long long)n * C;
    BQ_DISPATCH_T(dtype,
                  hipLaunchKernelGGL(gap_kernel<T_>, dim3(grid_for(total, 256)), dim3(256), 0, s,
                                     (const T_*)x, n, HW, C, ld, feat, mul));
    return (int)hipGetLastError();
}

int launch_head_final(const float* h1, int n, int mc_n, int pass0, long long tile0, const long long* tile0_dev, const long long* tile_idx,
                      unsigned seed_lo,
                      unsigned seed_hi, unsigned thresh, float dscale, const float* w2, const float* b2,
                      int init, int finalize, float* state, float* mean2, float* std2, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(head_final_kernel, dim3(n), dim3(64 * kHeadFinalWaves), 0, s, h1, n, mc_n, pass0, tile0, tile0_dev, tile_idx,
                       seed_lo, seed_hi, thresh, dscale, w2, b2, init, finalize, state, mean2, std2);
    return (int)hipGetLastError();
}

int launch_slide_reduce(const float* mean2, const float* std2, const int32_t* slide_idx, int n,
                        int n_slides, float tile_uq, long long* acc_pred, long long* acc_unc,
                        int32_t* count, hipStream_t s) {
    if (n <= 0) return 0;
    const int use_uq = (tile_uq == tile_uq) && tile_uq != 0.f;  // NaN / 0 disable the filter
    hipLaunchKernelGGL(slide_reduce_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, mean2, std2, slide_idx,
                       n, n_slides, tile_uq, use_uq, (unsigned long long*)acc_pred,
                       (unsigned long long*)acc_unc, count);
    return (int)hipGetLastError();
}

int launch_slide_finish(const long long* acc_pred, const long long* acc_unc, const int32_t* count,
                        int n_slides, double* mean_pred, double* mean_unc, hipStream_t s) {
    if (n_slides <= 0) return 0;
    hipLaunchKernelGGL(slide_finish_kernel, dim3(grid_for(n_slides, 256)), dim3(256), 0, s, acc_pred,
                       acc_unc, count, n_slides, mean_pred, mean_unc);
    return (int)hipGetLastError();
}

int launch_to_f32_nhwc(const void* x, long long rows, int C, int ld, float* out, int dtype, hipStream_t s) {
    const long long total = rows * C;
    BQ_DISPATCH_T(dtype,
                  hipLaunchKernelGGL(to_f32_nhwc_kernel<T_>, dim3(grid_for(total, 256)), dim3(256), 0, s,
                                     (const T_*)x, rows, C, ld, out));
    return (int)hipGetLastError();
}

int launch_nchw_to_f32_nhwc(const void* x, int n, int C, int HW, float* out, int dtype, hipStream_t s) {
    const long long total = (long long)n * C * HW;
    BQ_DISPATCH_T(dtype,
                  hipLaunchKernelGGL(nchw_to_f32_nhwc_kernel<T_>, dim3(grid_for(total, 256)), dim3(256), 0,
                                     s, (const T_*)x, n, C, HW, out));
    return (int)hipGetLastError();
}

// ---- K0 (optional front half): Reinhard-fast stain normalisation -------------------------------
// hp.py:19 normalizer='reinhard_fast', applied to the uint8 tile before the standardisation
// (results.py:251-256).  One workgroup per tile: pass 1 converts every pixel to CIE-LAB and reduces
// the six channel statistics in float64 (fixed thread map and tree: bit-reproducible); pass 2
// re-reads the (L2-resident) tile, converts again, applies (lab - mu) * (target_std / sd) + target_mean,
// converts back and stores uint8.  The precision contract (float64 for cbrt / pow / the statistics,
// one float32 rounding per other operation, no FMA contraction) is written out in oracle/stain.py;
// it makes the uint8 result comparable bit for bit.
namespace {

struct ReinhardConst {
    float m[9];        // XYZ from linear RGB
    float minv[9];     // linear RGB from XYZ
    float white[3];
    float rwhite[3];   // RN(1 / white): see divc
    float tgt_mean[3];
    float tgt_std[3];
};

struct Lab { float L, a, b; };

// x / c for a constant c, correctly rounded, in three operations instead of the ~10 of the IEEE division sequence (nine divisions by
// constants per pixel and pass): q = x * rc with rc = RN(1 / c), the exact residual by fma, one correction -- Markstein's theorem:
// RN(x / c) whenever rc is the correctly rounded reciprocal and c's significand is not all ones (0.95047, 1.08883, 116, 500, 200,
// 7.787: checked against the division itself on 56 M values, experiments/r06.md).  The contract of oracle/stain.py -- one float32
// rounding per operation -- is kept to the bit.
__device__ __forceinline__ float divc(float x, float c, float rc) {
    const float q = x * rc;
    const float r = __builtin_fmaf(-q, c, x);
    return __builtin_fmaf(r, rc, q);
}

// cbrt of a float32 t in (0.008856, ~1.1], "evaluated in float64 and rounded to float32" (the contract of oracle/stain.py), without
// the library's cbrt(double) (~80 double-precision operations): a float32 seed exp2(log2(t) / 3) (relative error ~1e-6), then two
// Newton steps y -= (y^3 - t) * r in float64 with ONE approximate reciprocal r ~ 1 / (3 y0^2) taken in float32 -- the error contracts
// by ~1e-6 per step, to the last bits of a double.  The float32 rounding of that differs from the rounding of the exact cube root only
// where the root lies within ~2e-16 (relative) of a float32 rounding boundary: one evaluation in ~3e8.
__device__ __forceinline__ float cbrt_f64_rounded(float t) {
    const float y0 = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(t) * (1.0f / 3.0f));
    const double r = (double)__builtin_amdgcn_rcpf(3.0f * y0 * y0);
    const double td = (double)t;
    double y = (double)y0;
    y = __builtin_fma(-(__builtin_fma(y * y, y, -td)), r, y);
    y = __builtin_fma(-(__builtin_fma(y * y, y, -td)), r, y);
    return (float)y;
}

#pragma clang fp contract(off)
__device__ __forceinline__ Lab rgb_to_lab(const float* __restrict__ lut, const ReinhardConst& k, unsigned r8,
                                          unsigned g8, unsigned b8) {
    const float r = lut[r8], g = lut[g8], b = lut[b8];
    float f[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float xyz = (k.m[3 * i] * r + k.m[3 * i + 1] * g) + k.m[3 * i + 2] * b;
        const float t = divc(xyz, k.white[i], k.rwhite[i]);
        f[i] = t > 0.008856f ? cbrt_f64_rounded(t) : 7.787f * t + (float)(16.0 / 116.0);
    }
    Lab o;
    o.L = 116.0f * f[1] - 16.0f;
    o.a = 500.0f * (f[0] - f[1]);
    o.b = 200.0f * (f[1] - f[2]);
    return o;
}

#pragma clang fp contract(off)
__device__ __forceinline__ void lab_to_rgb8(const ReinhardConst& k, const float* __restrict__ thr, float L, float a,
                                            float b, uint8_t* out) {
    const float fy = divc(L + 16.0f, 116.0f, 1.0f / 116.0f);
    const float fx = divc(a, 500.0f, 1.0f / 500.0f) + fy;
    const float fz = fy - divc(b, 200.0f, 1.0f / 200.0f);
    const float fv[3] = {fx, fy, fz};
    float xyz[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float v = fv[i];
        const float t = v > 0.2068966f ? (v * v) * v : divc(v - (float)(16.0 / 116.0), 7.787f, 1.0f / 7.787f);
        xyz[i] = t * k.white[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float c = (k.minv[3 * i] * xyz[0] + k.minv[3 * i + 1] * xyz[1]) + k.minv[3 * i + 2] * xyz[2];
        // out = clip(trunc(255 * clip(gamma(c), 0, 1)), 0, 255) is a monotone step function of c: the host evaluates the reference
        // formula (float64 power rounded to float32, then float32 steps) once per output level and hands over the 255 switching
        // points: thr[v-1] = smallest float32 c whose output is >= v.  Round 6: a fast float32 gamma gives the level to within one,
        // and two corrections against the switching points (two pairs of independent LDS reads) make it the formula's own result --
        // rounds 2-5 ran a bisection, eight DEPENDENT LDS reads per channel, which was most of this kernel's time.
        const float gam = c > 0.0031308f ? 1.055f * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(c) * (float)(1.0 / 2.4)) - 0.055f : 12.92f * c;
        int lo = gam > 0.f ? (int)(255.0f * fminf(gam, 1.0f)) : 0;          // (NaN -> 0, like the clip; and it stays 0 below)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const float t_hi = thr[lo < 255 ? lo : 254], t_lo = thr[lo > 0 ? lo - 1 : 0];
            const int up = (lo < 255 && c >= t_hi) ? 1 : 0;
            const int dn = (lo > 0 && !(c >= t_lo)) ? 1 : 0;
            lo += up - dn;
        }
        out[i] = (uint8_t)lo;
    }
}

// stats_out (optional): [n][6] = mean L, a, b, std L, a, b.  dst may be null (statistics only) or == src.
// One workgroup of 1 024 threads per tile (round 6: 512 left every SIMD with two waves and the kernel waiting on its own LDS reads).
constexpr int RH_NT = 1024;
__global__ void __launch_bounds__(RH_NT) reinhard_kernel(const uint8_t* __restrict__ tiles, int px,
                                                         const float* __restrict__ lut, const ReinhardConst k,
                                                         uint8_t* dst, float* __restrict__ stats_out) {
    const int npix = px * px;
    const uint8_t* src = tiles + (size_t)blockIdx.x * npix * 3;
    const int tid = threadIdx.x, nt = blockDim.x;
    __shared__ float slut[256];
    __shared__ float sthr[256];
    __shared__ double red[6][RH_NT / 64];
    __shared__ float stat[6];
    for (int i = tid; i < 256; i += nt) { slut[i] = lut[i]; sthr[i] = lut[256 + i]; }
    __syncthreads();

    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int i = tid; i < npix; i += nt) {
        const Lab v = rgb_to_lab(slut, k, src[3 * i], src[3 * i + 1], src[3 * i + 2]);
        s[0] += (double)v.L; s[1] += (double)v.a; s[2] += (double)v.b;
        s[3] += (double)v.L * (double)v.L; s[4] += (double)v.a * (double)v.a; s[5] += (double)v.b * (double)v.b;
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[q] += __shfl_xor(s[q], o);
        if ((tid & 63) == 0) red[q][tid >> 6] = s[q];
    }
    __syncthreads();
    if (tid < 3) {
        double a = 0, b = 0;
        for (int i = 0; i < nt / 64; ++i) { a += red[tid][i]; b += red[tid + 3][i]; }
        const double mu = a / (double)npix;
        double var = b / (double)npix - mu * mu;
        if (var < 0) var = 0;
        stat[tid] = (float)mu;
        stat[tid + 3] = (float)sqrt(var);
        if (stats_out) {
            stats_out[(size_t)blockIdx.x * 6 + tid] = (float)mu;
            stats_out[(size_t)blockIdx.x * 6 + tid + 3] = (float)sqrt(var);
        }
    }
    __syncthreads();
    if (!dst) return;
    uint8_t* o = dst + (size_t)blockIdx.x * npix * 3;
    float sc[3];
    {
#pragma clang fp contract(off)
        sc[0] = k.tgt_std[0] / stat[3]; sc[1] = k.tgt_std[1] / stat[4]; sc[2] = k.tgt_std[2] / stat[5];
    }
    for (int i = tid; i < npix; i += nt) {
#pragma clang fp contract(off)
        const Lab v = rgb_to_lab(slut, k, src[3 * i], src[3 * i + 1], src[3 * i + 2]);
        const float L = (v.L - stat[0]) * sc[0] + k.tgt_mean[0];
        const float a = (v.a - stat[1]) * sc[1] + k.tgt_mean[1];
        const float b = (v.b - stat[2]) * sc[2] + k.tgt_mean[2];
        uint8_t rgb[3];
        lab_to_rgb8(k, sthr, L, a, b, rgb);
        o[3 * i] = rgb[0]; o[3 * i + 1] = rgb[1]; o[3 * i + 2] = rgb[2];
    }
}

}  // namespace

int launch_reinhard(const uint8_t* tiles, int n, int px, const float* d_lut, const float* consts27,
                    const float* tgt_mean, const float* tgt_std, uint8_t* dst, float* d_stats, hipStream_t s) {
    if (n <= 0) return 0;
    ReinhardConst k;
    for (int i = 0; i < 9; ++i) { k.m[i] = consts27[i]; k.minv[i] = consts27[9 + i]; }
    for (int i = 0; i < 3; ++i) {
        k.white[i] = consts27[18 + i];
        k.rwhite[i] = 1.0f / consts27[18 + i];
        k.tgt_mean[i] = tgt_mean ? tgt_mean[i] : 0.f;
        k.tgt_std[i] = tgt_std ? tgt_std[i] : 1.f;
    }
    hipLaunchKernelGGL(reinhard_kernel, dim3(n), dim3(RH_NT), 0, s, tiles, px, d_lut, k, dst, d_stats);
    return (int)hipGetLastError();
}
