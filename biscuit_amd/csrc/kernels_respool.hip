// WHO STILL RUNS THIS FILE (round 5).  block_end (biscuit_hip.hip) comes here for block 2's end (shortcut K = 64) when the fused
// block tail of kernels_stream.hip does not run: a debug tap of block2_sepconv2 / block2_res / the float entry's taps, a blob without
// the "wp16" copies, or a tensor beyond the tail's 32-bit offsets.  Blocks 3, 4 and 13 (K >= 128) take gemm_tile_kernel<EPI_POOL>
// (kernels_split.hip) in that case; on the headline path blocks 2 and 3 end inside the fused tails.
//
// End of an Xception block with a strided shortcut (blocks 2, 3, 4, 13), bf16, one kernel:
//     out = MaxPool3x3/s2 'same' (y)  +  BN(Conv1x1/s2 'same' (x))
// y = the block's second separable convolution (full resolution), x = the block's input (the 1x1 / stride 2 / 'same'
// convolution samples it at even pixels).  Before: a GEMM kernel that wrote the shortcut tensor, and a pooling kernel
// that read it back (0.54 + 0.88 ms per batch of 256; the pooling kernel alone already ran at the ~5 TB/s this chip
// sustains).  Here the shortcut never goes to HBM:
//   phase A  a workgroup (4 waves) computes the shortcut of 64 pooled pixels x 128 channels with v_mfma_f32_32x32x16_bf16
//            (wave = 32 channels; the pixels' input rows staged in LDS in chunks of up to 256 channels, weights
//            host-packed in fragment order), applies the folded BN, rounds to bf16 -- the same rounding point as the
//            tensor it replaces -- and leaves the tile in LDS, over the input chunk;
//   phase B  the pooling pass of kernels_misc.hip (thread = 8 channels of one pooled pixel, nine coalesced 16-byte loads,
//            fp32 max) adds the shortcut from LDS and stores.
// Many small workgroups per CU (17 KiB of LDS, < 128 registers): phase A of one overlaps phase B of the others, which is
// what the earlier attempt -- pooling in the GEMM kernel's own store pass, lanes in accumulator layout -- could not do.
#include "gemm_common.h"

namespace {
using namespace bqk;

constexpr int RP_PIX = 64;              // pooled pixels per workgroup
constexpr int RP_NC = 128;              // channels per workgroup (one 32-wide fragment per wave)
constexpr int RP_STR = RP_NC * 2 + 16;  // LDS row stride of the shortcut tile: 272 B (odd number of 16-byte slots)

template <typename T>
struct RespoolParams {
    const T* x;             // [n][Hi][Wi][ldx]
    const uint4* wp;        // shortcut weights in 32x32x16 fragment order [K/16][nf32][64] x 16 B
    const float* scale;     // [nf32 * 32] folded BN
    const float* bias;
    const T* y;             // [n][Hi][Wi][ld]
    T* out;                 // [n][Ho][Wo][ld]
    int n, Hi, Wi, Ho, Wo, K, ldx, ld, nf32;
};

constexpr int RP_KCH = 256;             // input channels staged in LDS at a time

template <typename T>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) respool_kernel(const RespoolParams<T> p) {
    if constexpr (H16<T>::F16) bq_f16_saturate();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // x chunk [64][kc * 2 + 16], then the shortcut tile
    unsigned char* res = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const long long total = (long long)p.n * p.Ho * p.Wo;
    const long long gp0 = (long long)blockIdx.x * RP_PIX;
    const int nf = blockIdx.y * (RP_NC / 32) + wave;            // this wave's 32-channel fragment
    const int pt = (p.Hi & 1) ? 1 : 0, pl = (p.Wi & 1) ? 1 : 0; // TensorFlow 'same' padding of the pool: (1,1) odd, (0,1) even

    // ---- phase A: shortcut of pixels [gp0, gp0 + 64) x channels [32 nf, 32 nf + 32) ---------------------------------
    // x goes through LDS in chunks of up to 256 channels: thread = (pixel tid >> 2, every 4th 16-byte piece of its row),
    // so a pixel's row is read in 64-byte runs, once per workgroup
    const T* xrow;
    {
        long long gp = gp0 + (tid >> 2);
        gp = gp < total ? gp : total - 1;                       // past the end: a valid pixel, result never used
        const int xo = (int)(gp % p.Wo);
        const int yo = (int)((gp / p.Wo) % p.Ho);
        const int img = (int)(gp / ((long long)p.Wo * p.Ho));
        xrow = p.x + ((size_t)(img * p.Hi + 2 * yo) * p.Wi + 2 * xo) * p.ldx;
    }
    const uint4* wq = p.wp + (size_t)nf * 64 + lane;
    const size_t wstep = (size_t)p.nf32 * 64;
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    for (int kc0 = 0; kc0 < p.K; kc0 += RP_KCH) {
        const int kc = p.K - kc0 < RP_KCH ? p.K - kc0 : RP_KCH;
        const int xstr = kc * 2 + 16;                           // an odd number of 16-byte slots: conflict-free fragment reads
        if (kc0) __syncthreads();                               // the previous chunk's fragments are all read
        for (int pc = tid & 3; pc < kc / 8; pc += 4)
            *reinterpret_cast<uint4*>(smem + (tid >> 2) * xstr + pc * 16) =
                *reinterpret_cast<const uint4*>(xrow + kc0 + pc * 8);
        __syncthreads();
        const unsigned char* a0p = smem + l31 * xstr + h * 16;
        const unsigned char* a1p = a0p + 32 * xstr;
        const uint4* wk = wq + (size_t)(kc0 / 16) * wstep;
        uint4 b = wk[0];
        for (int ks = 0; ks < kc / 16; ++ks) {
            const uint4 bn = wk[(ks + 1 < kc / 16 ? ks + 1 : ks) * wstep];
            const uint4 a0 = *reinterpret_cast<const uint4*>(a0p + ks * 32);
            const uint4 a1 = *reinterpret_cast<const uint4*>(a1p + ks * 32);
            mma<T>(acc[0], b, a0);
            mma<T>(acc[1], b, a1);
            b = bn;
        }
    }
    __syncthreads();                                            // the shortcut tile goes over the x chunk
    // D layout: lane = pixel l31 of fragment m, register quad g = channels 32 nf + 8 g + 4 h + (0..3)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int cw = nf * 32 + g * 8 + h * 4;
        const float4 sc = *reinterpret_cast<const float4*>(p.scale + cw);
        const float4 bi = *reinterpret_cast<const float4*>(p.bias + cw);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float v0 = fmaf(acc[m][4 * g + 0], sc.x, bi.x), v1 = fmaf(acc[m][4 * g + 1], sc.y, bi.y);
            const float v2 = fmaf(acc[m][4 * g + 2], sc.z, bi.z), v3 = fmaf(acc[m][4 * g + 3], sc.w, bi.w);
            *reinterpret_cast<uint2*>(res + (m * 32 + l31) * RP_STR + (wave * 32 + g * 8 + h * 4) * 2) =
                make_uint2(H16<T>::pack2(v0, v1), H16<T>::pack2(v2, v3));
        }
    }
    __syncthreads();

    // ---- phase B: pooled y + shortcut, thread = (pixel, 8 channels) --------------------------------------------------
    const int ck = tid & 15;                                    // 8-channel chunk within the workgroup's 128 channels
    const int c0 = blockIdx.y * RP_NC + ck * 8;
    if (c0 >= p.ld) return;
#pragma unroll
    for (int i = 0; i < RP_PIX / 16; ++i) {
        const int px = (tid >> 4) + 16 * i;
        const long long gp = gp0 + px;
        if (gp >= total) break;
        const int xo = (int)(gp % p.Wo);
        const int yo = (int)((gp / p.Wo) % p.Ho);
        const int img = (int)(gp / ((long long)p.Wo * p.Ho));
        float mx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) mx[j] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = 2 * yo + dy - pt;
            if ((unsigned)yy >= (unsigned)p.Hi) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = 2 * xo + dx - pl;
                if ((unsigned)xx >= (unsigned)p.Wi) continue;
                const uint4 u = *reinterpret_cast<const uint4*>(p.y + ((size_t)(img * p.Hi + yy) * p.Wi + xx) * p.ld + c0);
                const T* e = reinterpret_cast<const T*>(&u);
#pragma unroll
                for (int j = 0; j < 8; ++j) mx[j] = fmaxf(mx[j], (float)e[j]);
            }
        }
        const uint4 ru = *reinterpret_cast<const uint4*>(res + px * RP_STR + ck * 16);
        const T* re = reinterpret_cast<const T*>(&ru);
        uint4 ou;
        T* oe = reinterpret_cast<T*>(&ou);
#pragma unroll
        for (int j = 0; j < 8; ++j) oe[j] = (T)(mx[j] + (float)re[j]);
        *reinterpret_cast<uint4*>(p.out + (size_t)gp * p.ld + c0) = ou;
    }
}

}  // namespace

// x: the block's input, sampled at even pixels; wp32: shortcut weights in 32x32x16 fragment order ("<layer>/wp32");
// y: the tensor to pool.  K and ldx multiples of 16, ld a multiple of 8, nf32 * 32 >= ld rounded up to 128.
namespace {
template <typename T>
int launch_respool_t(const void* x, const void* wp32, const float* scale, const float* bias, const void* y, void* out,
                     int n, int Hi, int Wi, int K, int ldx, int ld, int nf32, hipStream_t s) {
    const int ncb = (ld + RP_NC - 1) / RP_NC;                   // 128-channel blocks (grid.y)
    if (K % 16 || ldx % 8 || ld % 8 || nf32 * 32 < ncb * RP_NC || !wp32 || !scale || !bias) return (int)hipErrorInvalidValue;
    RespoolParams<T> p;
    p.x = reinterpret_cast<const T*>(x);
    p.wp = reinterpret_cast<const uint4*>(wp32);
    p.scale = scale; p.bias = bias;
    p.y = reinterpret_cast<const T*>(y);
    p.out = reinterpret_cast<T*>(out);
    p.n = n; p.Hi = Hi; p.Wi = Wi; p.Ho = (Hi + 1) / 2; p.Wo = (Wi + 1) / 2;
    p.K = K; p.ldx = ldx; p.ld = ld; p.nf32 = nf32;
    const long long total = (long long)n * p.Ho * p.Wo;
    if (total <= 0) return 0;
    const int kc = K < RP_KCH ? K : RP_KCH;
    const int xs = RP_PIX * (kc * 2 + 16), rs = RP_PIX * RP_STR;
    hipLaunchKernelGGL(respool_kernel<T>, dim3((unsigned)((total + RP_PIX - 1) / RP_PIX), (unsigned)ncb),
                       dim3(256), xs > rs ? xs : rs, s, p);
    return (int)hipGetLastError();
}
}  // namespace

int launch_respool(int dtype, const void* x, const void* wp32, const float* scale, const float* bias, const void* y,
                   void* out, int n, int Hi, int Wi, int K, int ldx, int ld, int nf32, hipStream_t s) {
    return dtype == 2 ? launch_respool_t<f16_t>(x, wp32, scale, bias, y, out, n, Hi, Wi, K, ldx, ld, nf32, s)
                      : launch_respool_t<bf16_t>(x, wp32, scale, bias, y, out, n, Hi, Wi, K, ldx, ld, nf32, s);
}
