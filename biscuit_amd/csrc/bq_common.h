// Shared declarations of libbiscuit_hip.so (gfx950 only; no CUDA/dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include <stdlib.h>

// Experiment switches (ablations, alternative tilings, in-kernel stamps) exist only in -DBQ_EXPERIMENTS builds
// (`make EXPERIMENTS=1`); the product library reads no environment variable and carries no debug branch.
#ifdef BQ_EXPERIMENTS
inline const char* bq_exp_env(const char* name) { return getenv(name); }
#else
inline const char* bq_exp_env(const char*) { return nullptr; }
#endif

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the current device only: remember, per device, the
// largest size already set for one kernel (one `BqLdsAttr` object per kernel instantiation).
struct BqLdsAttr {
    static constexpr int kMaxDev = 64;
    size_t set[kMaxDev] = {};
    int ensure(const void* kern, size_t lds) {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = -1;
        if (dev >= 0 && lds <= set[dev]) return 0;
        const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0) set[dev] = lds;
        return 0;
    }
};

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// IEEE half: the second 16-bit storage / matrix-core type (BQ_DTYPE_F16).  Same MFMA rate as bf16, 8x finer rounding
// (11 against 8 significand bits), range +-65504: kernels that write it run with MODE.FP16_OVFL set, so an overflow
// saturates instead of becoming inf (bq_f16_saturate below).
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- fused "producer -> LDS A tile -> MFMA" kernel -------------------------------
enum BqProducer : int {
    PROD_DW = 0,      // depthwise 3x3 'same' of an NHWC map (SeparableConv2D first half)
    PROD_DW_RELU = 1, // same, ReLU applied to the input on load (block*_sepconv1_act)
    PROD_S2 = 2,      // 1x1 / stride-2 gather (residual Conv2D(1, strides=2, 'same'))
    PROD_IM2COL = 3   // 3x3 valid im2col, K = 9*C (block1_conv2)
};

enum BqShape : int {       // MF, WM, WN, RN   (tile rows = 32*MF, waves = WM*WN)
    SHAPE_A = 0,           // 4, 2, 2, 1   N = 64           (block1_conv2)
    SHAPE_B = 1,           // 4, 1, 4, 1   N = 128
    SHAPE_C = 2,           // 4, 1, 4, 2   N = 256
    SHAPE_D = 3,           // 3, 1, 8, 3   N = 768 per pass (728-wide layers)
    SHAPE_E = 4,           // 3, 1, 8, 2   N = 512 per pass
    SHAPE_F = 5,           // 2, 1, 8, 3
    SHAPE_G = 6,           // 1, 1, 8, 4
    SHAPE_H = 7,           // 1, 1, 4, 2   fp32 fallback
    SHAPE_I = 8,           // 1, 1, 8, 2   (unused since the MC head got its own kernel)
    SHAPE_J = 9,           // 2, 1, 4, 1   N = 128, 64-row tiles (1x1/s2 residual convs: more workgroups per CU)
    SHAPE_K = 10           // 2, 1, 4, 2   N = 256, 64-row tiles
};

struct GemmParams {
    const void* in;        // producer input
    const void* wp;        // fragment-packed weights [NFp][KBtot][64][VEC]
    const float* scale;    // per-cout, may be null (=1)
    const float* bias;     // per-cout, may be null (=0)
    const float* dw;       // [9][ldi] depthwise taps (PROD_DW*)
    const void* residual;  // optional [M][ldo], added before ReLU
    void* out;             // [M][ldo]
    int M;                 // rows (pixels, or tile*pass rows for the head)
    int K;                 // contraction length handled by this launch (multiple of 2*VEC)
    int KBtot;             // k-blocks per n-frag in wp (stride), kb0 = first block used
    int kb0;
    int k_off;             // first input channel / unit handled (split-K launches)
    int NFp;               // n-frags to compute (multiple of WN*RN)
    int Nstore;            // output channels actually stored (multiple of 4, <= ldo)
    int ldo;               // output row stride (elements)
    int ldi;               // input row stride (elements)
    int H, W;              // output spatial size
    int Hi, Wi;            // input spatial size
    int relu;              // ReLU in the epilogue
    int lds_total;         // dynamic LDS bytes of the launch (set by the pipe launcher)
    float gap_mul;         // EPI_GAP of kernels_split.hip: factor on the means (0 = 1: undoes an activation exponent)
};

size_t gemm_lds_bytes(int dtype, int shape, int K);
// Returns hipError_t as int.
int launch_gemm(int dtype, int prod, int shape, const GemmParams& p, hipStream_t s);
int gemm_tile_rows(int shape);
bool pipe_supported(int dtype, int prod, int nfp, int W, int K);
int launch_sepconv_pipe(int dtype, int prod, const GemmParams& p, hipStream_t s);
bool wide_supported(int dtype, int prod, int nfp, int H, int W, int K, int Nstore, int ldi, int ldo, long long M, bool residual);
int launch_sepconv_wide(int dtype, int prod, const GemmParams& p, const void* wp16, int num_cus, hipStream_t s);
bool stream_supported(int dtype, int cin, int cout, bool relu_in, long long n, int H, int W);
int launch_sepconv_stream(int dtype, int cin, int cout, bool relu_in, const void* in, const void* wp16, const float* dw,
                          const float* scale, const float* bias, void* out, int n, int H, int W, int relu, int num_cus,
                          hipStream_t s);
bool tail_supported(int dtype, int cin, int cout, int cx, long long n, int H, int W);
int launch_block_tail(int dtype, int cin, int cout, int cx, const void* y1, const void* wp16, const float* dw,
                      const float* scale, const float* bias, const void* x, const void* wr16, const float* rscale,
                      const float* rbias, void* out, int n, int H, int W, int num_cus, hipStream_t s);
int launch_dw3x3(int dtype, const void* in, const float* dw, void* out, int n, int H, int W, int C, int relu, hipStream_t s);
int launch_gemm_tile(int dtype, const GemmParams& p, bool s2, hipStream_t s, int epi = 0);   // epi: 1 = + global average pool, 2 = + max-pool + add
// kernels_exit.hip (round 4): block 14's pointwise GEMMs, one image's pixels x 256 channels per workgroup; gap != nullptr: the
// per-image means instead of the tensor
bool exit_supported(int dtype, int K, int N, int HW, long long n);
int launch_exit_gemm(int dtype, const void* in, const void* wp16, const float* scale, const float* bias, void* out, float* gap,
                     int n, int HW, int K, int N, int relu, float gap_mul, hipStream_t s);
int launch_tile_conv(int dtype, int kind, const void* in, const void* wp, const float* dw, const float* scale,
                     const float* bias, void* out, int n, int H, int W, int Hi, int Wi, int relu, int num_cus,
                     hipStream_t s);

// ---- small kernels (kernels_misc.hip) ----------------------------------------------
int launch_stage_u8(const uint8_t* tiles, int n, int px, void* out, int dtype, double* stats_scratch,
                    hipStream_t s);
int launch_reinhard(const uint8_t* tiles, int n, int px, const float* d_lut, const float* consts27,
                    const float* tgt_mean, const float* tgt_std, uint8_t* dst, float* d_stats, hipStream_t s);
int launch_stage_stats(const uint8_t* tiles, int n, int px, double* stats_scratch, hipStream_t s);
int launch_front(int dtype, const uint8_t* tiles, const unsigned long long* stats, const void* ws16, const float* s_scale,
                 const float* s_bias, const void* wc16, const float* c_scale, const float* c_bias, void* out, int n, int num_cus,
                 hipStream_t s);
int launch_stage_f32(const float* tiles, int n, int px, void* out, int dtype, hipStream_t s);
// kernels_png.hip: PNG scanline un-filtering (rows: [n][px][1 + 3 px] filter byte + filtered RGB bytes -> out uint8 NHWC)
int launch_png_unfilter(const unsigned char* rows, int n, int px, unsigned char* out, hipStream_t s, size_t in_stride = 0);
// kernels_inflate.hip: n zlib streams (packed, 16-byte aligned starts) -> n x out_len bytes, Adler-32 verified; status 0 = ok
size_t inflate_scratch_bytes(int n);
int launch_inflate(const unsigned char* d_z, const unsigned* d_off, const unsigned* d_len, int n, unsigned char* d_out, unsigned out_len,
                   unsigned out_stride, void* d_scratch, int* d_status, hipStream_t s, int variant = 0, unsigned row_len = 0);
// (row_len != 0: the outputs are PNG scanlines of that length; a filter-type byte above 4 is flagged too)
int launch_stem1(const void* in_nchw, int n, const float* w27x32, const float* scale,
                 const float* bias, void* out_nhwc, int dtype, hipStream_t s);
int launch_pool_add(const void* y, const void* res, void* out, int n, int Hi, int Wi, int C,
                    int dtype, hipStream_t s);
int launch_respool(int dtype, const void* x, const void* wp32, const float* scale, const float* bias, const void* y, void* out,
                   int n, int Hi, int Wi, int K, int ldx, int ld, int nf32, hipStream_t s);
int launch_gap(const void* x, int n, int HW, int C, int ld, float* feat, float mul, int dtype, hipStream_t s);
int launch_head_dense(const float* in, const void* wh, const void* wl, const float* bias, float* out, int rows, int K,
                      int mc_n, int pass0, int in_row_is_tile, int layer, unsigned seed_lo, unsigned seed_hi, unsigned thresh,
                      float dscale, long long tile0, const long long* tile0_dev, const long long* tile_idx, hipStream_t s);
int launch_head_final(const float* h1, int n, int mc_n, int pass0, long long tile0, const long long* tile0_dev, const long long* tile_idx,
                      unsigned seed_lo, unsigned seed_hi, unsigned thresh, float dscale,
                      const float* w2, const float* b2, int init, int finalize, float* state,
                      float* mean2, float* std2, hipStream_t s);
int launch_slide_reduce(const float* mean2, const float* std2, const int32_t* slide_idx, int n,
                        int n_slides, float tile_uq, long long* acc_pred, long long* acc_unc,
                        int32_t* count, hipStream_t s);
size_t roc_workspace_bytes(long long n);
int launch_roc_youden(const double* score, const unsigned char* label, long long n, unsigned char* ws, size_t ws_bytes,
                      double* out, hipStream_t s);
int launch_slide_finish(const long long* acc_pred, const long long* acc_unc, const int32_t* count,
                        int n_slides, double* mean_pred, double* mean_unc, hipStream_t s);
int launch_to_f32_nhwc(const void* x, long long rows, int C, int ld, float* out, int dtype,
                       hipStream_t s);
int launch_nchw_to_f32_nhwc(const void* x, int n, int C, int HW, float* out, int dtype,
                            hipStream_t s);

// ---- device helpers ----------------------------------------------------------------
#define BQ_FIXED_SHIFT 40

// MODE.FP16_OVFL (bit 23 of the wave's MODE register) = 1: an FP16 result that overflows is clamped to +-MAX_FP16
// (true infinities are kept).  First statement of every kernel that converts to f16.
__device__ __forceinline__ void bq_f16_saturate() {
    __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);   // hwreg(HW_REG_MODE, 23, 1)
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned)b) << 16);
}

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                              unsigned k0, unsigned k1, unsigned (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}
