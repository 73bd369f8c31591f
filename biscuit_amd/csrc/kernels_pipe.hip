// WHO STILL RUNS THIS FILE (round 5).  Nobody on the headline path: the 728-wide layers run on kernels_wide.hip.  run_conv falls
// through to launch_sepconv_pipe when wide_supported() refuses a launch -- a blob without "<layer>/wp16", or more than 2^32 bytes of
// activations per tensor (n x 361 x 736 x 2: batches beyond ~8 000 tiles) -- and the pipelined form fits (pipe_supported).  Round 1's
// dominant kernel, kept as the correct fallback; same arithmetic order as the wide kernel.
//
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

// All LDS addresses are formed as `smem + integer offset` so the compiler keeps them in the
// LDS address space (ds_* instructions); pointer arrays / lambdas capturing pointers decay
// to flat addressing and scratch.
// Raw staging registers as three named values (an array here ends up in scratch).
struct Raw3 { uint4 a, b, c, d, e; };   // up to 5 staging slots (NRAW of them used)

typedef unsigned short h16_t;   // either 16-bit storage type, for address arithmetic

template <int NT>
__device__ __forceinline__ uint4 raw_load1(const h16_t* __restrict__ in, int ldi, int coff, int row, int p_lo, int M) {
    int prow = p_lo + row;
    prow = prow < 0 ? 0 : (prow >= M ? M - 1 : prow);
    return *reinterpret_cast<const uint4*>(in + (size_t)prow * ldi + coff);
}

template <int NT, int NRAW>
__device__ __forceinline__ Raw3 raw_load(const h16_t* __restrict__ in, int ldi, int c, int K, int jch, int tid,
                                         int p_lo, int M) {
    // Branch-free (clamped) so the loads stay in flight: rows outside the tensor and pieces
    // past K load valid-but-unused data.
    int coff = c * KC + jch * 8;
    coff = coff < K - 8 ? coff : K - 8;
    Raw3 r;
    r.a = raw_load1<NT>(in, ldi, coff, tid >> 3, p_lo, M);
    r.b = raw_load1<NT>(in, ldi, coff, (tid + NT) >> 3, p_lo, M);
    r.c = raw_load1<NT>(in, ldi, coff, (tid + 2 * NT) >> 3, p_lo, M);
    if constexpr (NRAW > 3) r.d = raw_load1<NT>(in, ldi, coff, (tid + 3 * NT) >> 3, p_lo, M);
    if constexpr (NRAW > 4) r.e = raw_load1<NT>(in, ldi, coff, (tid + 4 * NT) >> 3, p_lo, M);
    return r;
}

// ReLU on two packed bf16 / f16: as signed 16-bit integers negative floats (and -0) are negative,
// so max(x, 0) per half is exactly ReLU -- one v_pk_max_i16 per dword.
__device__ __forceinline__ unsigned relu_bf16x2(unsigned x) { return relu_pk16(x); }

template <int NT, bool RELU>
__device__ __forceinline__ void raw_store1(uint4 v, unsigned char* smem, int raw_off, int jch, int row,
                                           int p_lo, int HP, int M) {
    const int prow = p_lo + row;
    if (RELU) { v.x = relu_bf16x2(v.x); v.y = relu_bf16x2(v.y); v.z = relu_bf16x2(v.z); v.w = relu_bf16x2(v.w); }
    if (row < HP && prow >= 0 && prow < M)
        *reinterpret_cast<uint4*>(smem + raw_off + row * RAW_ROW + jch * 16) = v;
}

template <int NT, bool RELU, int NRAW>
__device__ __forceinline__ void raw_store(const Raw3& r, unsigned char* smem, int raw_off, int jch, int tid,
                                          int p_lo, int HP, int M) {
    raw_store1<NT, RELU>(r.a, smem, raw_off, jch, tid >> 3, p_lo, HP, M);
    raw_store1<NT, RELU>(r.b, smem, raw_off, jch, (tid + NT) >> 3, p_lo, HP, M);
    raw_store1<NT, RELU>(r.c, smem, raw_off, jch, (tid + 2 * NT) >> 3, p_lo, HP, M);
    if constexpr (NRAW > 3) raw_store1<NT, RELU>(r.d, smem, raw_off, jch, (tid + 3 * NT) >> 3, p_lo, HP, M);
    if constexpr (NRAW > 4) raw_store1<NT, RELU>(r.e, smem, raw_off, jch, (tid + 4 * NT) >> 3, p_lo, HP, M);
}

// D stage: depthwise 3x3 of chunk c, raw rows at smem+raw_off -> A chunk at smem+a_off.
// Per tap and 16-byte piece: 8 v_perm_b32 (bf16 -> f32 bits and the 'same'-padding mask in
// one byte-permute: an out-of-image tap selects constant-zero bytes) + 4 v_pk_fma_f32 against
// taps pre-permuted to (w0,w2,w1,w3 | w4,w6,w5,w7).  ReLU, when the layer has one in front,
// was already applied to the raw rows as they were staged.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T, int NT, int MT, int NITEM>
__device__ __forceinline__ void depthwise(unsigned char* smem, int raw_off, int a_off, int wl_off, int c, int K,
                                          int W, int jch, int tid, const unsigned (&item_mask)[NITEM]) {
    if (c * KC + jch * 8 >= K) {                   // padded channel tail of the last chunk: A = 0 (the matrix
#pragma unroll                                     // stage always runs whole chunks)
        for (int q = 0; q < NITEM; ++q) {
            const int r = (tid + q * NT) >> 3;
            if (r < MT) *reinterpret_cast<uint4*>(smem + a_off + r * A_STR + jch * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
        return;
    }
    const int wbase = wl_off + (c * KC + jch * 8) * 4;
#pragma unroll
    for (int q = 0; q < NITEM; ++q) {
        const int r = (tid + q * NT) >> 3;
        if (r < MT) {                              // wave-uniform (NT and MT*8 are multiples of 64)
            f32x2 aA = {0.f, 0.f}, aB = {0.f, 0.f}, aC = {0.f, 0.f}, aD = {0.f, 0.f};
            const int base = raw_off + (r + W + 1) * RAW_ROW + jch * 16;
            // one tap row at a time (outer loop not unrolled): keeps the live set small, the
            // accumulators of the matrix-core stage leave few spare registers
#pragma unroll 1
            for (int dy = 0; dy < 3; ++dy) {
                const int rowoff = base + (dy - 1) * W * RAW_ROW;
                const int woff = wbase + dy * 3 * K * 4;
                const unsigned mrow = item_mask[q] >> (dy * 3);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const uint4 v = *reinterpret_cast<const uint4*>(smem + rowoff + (dx - 1) * RAW_ROW);
                    const bool ok = (mrow >> dx) & 1u;
                    const float4 w0 = *reinterpret_cast<const float4*>(smem + woff + dx * K * 4);
                    const float4 w1 = *reinterpret_cast<const float4*>(smem + woff + dx * K * 4 + 16);
                    f32x2 lo01, hi01, lo23, hi23;
                    if constexpr (H16<T>::F16) {
                        // f16: the padding mask is a select on the packed dwords, then one conversion per half
                        typedef H16<T> F;
                        const unsigned vx = ok ? v.x : 0u, vy = ok ? v.y : 0u, vz = ok ? v.z : 0u, vw = ok ? v.w : 0u;
                        lo01 = (f32x2){F::lo(vx), F::lo(vy)}; hi01 = (f32x2){F::hi(vx), F::hi(vy)};
                        lo23 = (f32x2){F::lo(vz), F::lo(vw)}; hi23 = (f32x2){F::hi(vz), F::hi(vw)};
                    } else {
                        const unsigned sl = ok ? 0x01000c0cu : 0x0c0c0c0cu;   // low bf16  -> f32 bits (<< 16)
                        const unsigned sh = ok ? 0x03020c0cu : 0x0c0c0c0cu;   // high bf16 -> f32 bits (& 0xffff0000)
                        lo01 = (f32x2){__uint_as_float(__builtin_amdgcn_perm(0u, v.x, sl)),
                                       __uint_as_float(__builtin_amdgcn_perm(0u, v.y, sl))};   // ch 0, 2
                        hi01 = (f32x2){__uint_as_float(__builtin_amdgcn_perm(0u, v.x, sh)),
                                       __uint_as_float(__builtin_amdgcn_perm(0u, v.y, sh))};   // ch 1, 3
                        lo23 = (f32x2){__uint_as_float(__builtin_amdgcn_perm(0u, v.z, sl)),
                                       __uint_as_float(__builtin_amdgcn_perm(0u, v.w, sl))};   // ch 4, 6
                        hi23 = (f32x2){__uint_as_float(__builtin_amdgcn_perm(0u, v.z, sh)),
                                       __uint_as_float(__builtin_amdgcn_perm(0u, v.w, sh))};   // ch 5, 7
                    }
                    aA = __builtin_elementwise_fma((f32x2){w0.x, w0.y}, lo01, aA);
                    aB = __builtin_elementwise_fma((f32x2){w0.z, w0.w}, hi01, aB);
                    aC = __builtin_elementwise_fma((f32x2){w1.x, w1.y}, lo23, aC);
                    aD = __builtin_elementwise_fma((f32x2){w1.z, w1.w}, hi23, aD);
                }
            }
            const float acc[8] = {aA.x, aB.x, aA.y, aB.y, aC.x, aD.x, aC.y, aD.y};
            *reinterpret_cast<uint4*>(smem + a_off + r * A_STR + jch * 16) = pack<T>(acc);
        }
    }
}

// G stage: the matrix cores on one 64-channel chunk of A (KBC k-blocks), B through the ring.
// Straight-line on purpose: all k-blocks of a chunk always run (k-blocks past K multiply zero rows of A by
// a valid, unused weight block).  A load under a branch makes the number of outstanding loads
// path-dependent, and the compiler then waits vmcnt(0) - the whole ring - before every k-block.
template <typename T, int MF, int RN, int PF, int KBC>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[MF][RN], uint4 (&bq)[PF][RN], const unsigned char* smem,
                                          int a_base, const uint4* __restrict__ bp0, int c, int KB, int KBtot) {
#pragma unroll
    for (int d = 0; d < KBC; ++d) {
        const int kb = c * KBC + d;
        uint4 a[MF];
#pragma unroll
        for (int i = 0; i < MF; ++i)
            a[i] = *reinterpret_cast<const uint4*>(smem + a_base + i * 32 * A_STR + d * 32);
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) mma<T>(acc[i][j], bq[d & (PF - 1)][j], a[i]);
        const int nx = kb + PF;
        const int idx = nx < KB ? nx : KB - 1;
#pragma unroll
        for (int j = 0; j < RN; ++j) bq[d & (PF - 1)][j] = bp0[((size_t)j * KBtot + idx) * 64];
    }
}

// Epilogue, part 1: folded BN (+ residual) (+ ReLU) in registers, result parked in LDS in its final dtype.
// Same arithmetic as gemm_common.h's epilogue_to_lds, but nothing here is a global load under a branch:
// scale and bias come from LDS (`sb`, written in the prologue) and the residual tile, when there is one, was
// copied into the staging rows by LDS-DMA (`residual_dma`) -- each lane then reads back exactly the 8 bytes it
// is about to overwrite.  With the loads in the branches the compiler waited vmcnt(0) after every one of them:
// 24 (scale/bias) + 36 (residual) serialised L2/HBM round trips, 11 k and 30 k cycles of a 117-146 k tile.
template <typename T, int MF, int RN>
__device__ __forceinline__ void pipe_epilogue_to_lds(const GemmParams& p, const f32x16 (&acc)[MF][RN], int nfb,
                                                     int row_local0, int r32, int h, unsigned char* smem,
                                                     const float* sb, int nfp32, bool res_in_lds) {
    const int sstride = stage_stride<bf16_t>(p.Nstore);
    // branch-free ReLU on the packed result: a signed 16-bit max with 0 clears negative bf16 values (rounding is
    // monotone and keeps the sign, so ReLU after the conversion gives the same bits as before it); with 0x8000
    // (the smallest int16) it is a no-op
    const unsigned lo2 = p.relu ? 0u : 0x80008000u;
    const int nfbu = __builtin_amdgcn_readfirstlane(nfb);
    unsigned char* lane_base = smem + (size_t)(row_local0 + r32) * sstride + h * 8;
    const float* sbl = sb + h * 4;
#pragma unroll
    for (int j = 0; j < RN; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ng = (nfbu + j) * 32 + g * 8;        // wave-uniform; Nstore is a multiple of 8
            if (ng < p.Nstore) {
                const float4 sc = *reinterpret_cast<const float4*>(sbl + ng);
                const float4 bi = *reinterpret_cast<const float4*>(sbl + nfp32 + ng);
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    uint2* slot = reinterpret_cast<uint2*>(lane_base + (size_t)i * 32 * sstride + ng * 2);
                    float v0 = fmaf(acc[i][j][g * 4 + 0], sc.x, bi.x);
                    float v1 = fmaf(acc[i][j][g * 4 + 1], sc.y, bi.y);
                    float v2 = fmaf(acc[i][j][g * 4 + 2], sc.z, bi.z);
                    float v3 = fmaf(acc[i][j][g * 4 + 3], sc.w, bi.w);
                    if (res_in_lds) {
                        const uint2 u = *slot;
                        v0 = H16<T>::add_lo(v0, u.x); v1 = H16<T>::add_hi(v1, u.x);
                        v2 = H16<T>::add_lo(v2, u.y); v3 = H16<T>::add_hi(v3, u.y);
                    }
                    uint2 o;
                    o.x = H16<T>::pack2(v0, v1);
                    o.y = H16<T>::pack2(v2, v3);
                    asm("v_pk_max_i16 %0, %1, %2" : "=v"(o.x) : "v"(o.x), "v"(lo2));
                    asm("v_pk_max_i16 %0, %1, %2" : "=v"(o.y) : "v"(o.y), "v"(lo2));
                    *slot = o;
                }
            }
        }
    }
}

// The residual tile goes into the staging rows by LDS-DMA, no registers involved: a wave instruction writes
// 64 x 16 contiguous bytes of LDS, so a row of up to 128 pieces takes two.  Rows past M are left alone (their
// staging rows are never stored).
template <int NT, int MT>
__device__ __forceinline__ void residual_dma(const GemmParams& p, int m0, int tid, unsigned char* smem, int r_begin,
                                             int r_end) {
    const int row_bytes = p.ldo * 2;
    const int ppr = p.Nstore >> 3;                 // 16-byte pieces per row
    const int sstride = stage_stride<bf16_t>(p.Nstore);
    int rows = p.M - m0 < MT ? p.M - m0 : MT;
    if (rows > r_end) rows = r_end;
    const unsigned char* __restrict__ src = reinterpret_cast<const unsigned char*>(p.residual) + (size_t)m0 * row_bytes;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    constexpr int NW = NT / 64;
    for (int r = r_begin + wave; r < rows; r += NW) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int piece = half * 64 + lane;
            if (piece < ppr)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + (size_t)r * row_bytes + piece * 16),
                    (__attribute__((address_space(3))) void*)(smem + (size_t)r * sstride + half * 1024), 16, 0, 0);
        }
    }
}

// The same copy for use INSIDE the K loop, as inline asm: the compiler treats a visible LDS-DMA as aliasing
// every later LDS read (vmcnt(0) in front of each ds_read of the depthwise stage: +20 % per chunk).  An asm load
// is absent from its vmcnt bookkeeping, which only makes its counted waits stricter (the DMA is younger than
// the loads they guard); the epilogue waits vmcnt(0) itself before the barrier that publishes the rows.
template <int NT, int MT>
__device__ __forceinline__ void residual_dma_row_asm(const GemmParams& p, int m0, int tid, unsigned lds_base, int r) {
    const int row_bytes = p.ldo * 2;
    const int ppr = p.Nstore >> 3;
    const int sstride = stage_stride<bf16_t>(p.Nstore);
    if (m0 + r >= p.M) return;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(p.residual) + (size_t)(m0 + r) * row_bytes;
    const int lane = tid & 63;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int piece = half * 64 + lane;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(r * sstride + half * 1024));
        if (piece < ppr) {
            const unsigned char* g = src + piece * 16;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
        }
    }
}

// Halo rows of chunk c straight into a raw buffer by LDS-DMA (layers without a ReLU in front: the staged bytes
// are the input bytes).  One instruction = 8 halo pixels x 128 B, contiguous in LDS; instructions k = first,
// first + step, ... are this wave's.  Rows outside the tensor stay stale (their taps are masked to zero bytes by
// the depthwise stage's permutes), pieces past K re-read the last valid piece, like the register path.
__device__ __forceinline__ void halo_dma_asm(const h16_t* __restrict__ in, int ldi, int c, int K, int p_lo, int M,
                                             int HP, int lane, int first, int step, unsigned lds_raw) {
    int coff = c * KC + (lane & 7) * 8;
    coff = coff < K - 8 ? coff : K - 8;
    for (int k = first; k * 8 < HP; k += step) {
        const int row = k * 8 + (lane >> 3);
        const int prow = p_lo + row;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_raw + (unsigned)k * 1024u);
        if (row < HP && prow >= 0 && prow < M) {
            const h16_t* g = in + (size_t)prow * ldi + coff;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
        }
    }
}

// The same stage with the A fragments of k-block d+1 read from LDS before the MFMAs of k-block d (12 more live
// registers: only for instances that have them to spare).
template <typename T, int MF, int RN, int PF, int KBC>
__device__ __forceinline__ void mma_chunk_pre(f32x16 (&acc)[MF][RN], uint4 (&bq)[PF][RN], const unsigned char* smem,
                                              int a_base, const uint4* __restrict__ bp0, int c, int KB, int KBtot) {
    uint4 a[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) a[i] = *reinterpret_cast<const uint4*>(smem + a_base + i * 32 * A_STR);
#pragma unroll
    for (int d = 0; d < KBC; ++d) {
        const int kb = c * KBC + d;
        uint4 an[MF];
#pragma unroll
        for (int i = 0; i < MF; ++i)
            an[i] = d + 1 < KBC ? *reinterpret_cast<const uint4*>(smem + a_base + i * 32 * A_STR + (d + 1) * 32) : a[i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) mma<T>(acc[i][j], bq[d & (PF - 1)][j], a[i]);
        const int nx = kb + PF;
        const int idx = nx < KB ? nx : KB - 1;
#pragma unroll
        for (int j = 0; j < RN; ++j) bq[d & (PF - 1)][j] = bp0[((size_t)j * KBtot + idx) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MF; ++i) a[i] = an[i];
    }
}

// WM groups of WN waves: group g owns rows [32*MF*g, 32*MF*(g+1)) of the tile, every group all output columns.
// WM = 2 (16 waves, 192-row tiles) is for the 256-wide layers on large maps: a 74-wide map needs 150 halo
// pixels around ANY flattened tile, so twice the rows per tile means 30 % less halo traffic per pixel, the
// weights are streamed once per 192 rows, and the 1024-thread workgroup runs 4 waves per SIMD.
template <typename T, bool RELU, int MF, int WN, int RN, int NRAW, int WM>
__global__ void __launch_bounds__(64 * WN * WM) sepconv_pipe_kernel(const GemmParams p) {
    if constexpr (H16<T>::F16) bq_f16_saturate();
    constexpr int NT = 64 * WN * WM;
    constexpr int MT = 32 * MF * WM;
    constexpr int NITEM = (MT * CPR + NT - 1) / NT;
    constexpr int KBC = KC / 16;                   // k-blocks per chunk (4)
    constexpr int PF = 2;                          // B register ring depth (k-blocks ahead)
    static_assert(NT % CPR == 0, "a thread must keep the same channel piece for all its items");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem0[];

    const int tid = threadIdx.x;
    const int W = p.W, H = p.H;
    const int HP = MT + 2 * (W + 1);               // halo rows of the flattened pixel range
    const int raw_bytes = HP * RAW_ROW;            // LDS map: raw[0] | raw[1] | A[0] | A[1] | taps
    const int a_off0 = 2 * raw_bytes;
    const int wl_off = a_off0 + 2 * MT * A_STR;
    // The loop's buffers sit at the END of the allocation (scale/bias last): the staging rows below them are free
    // during the K loop, and the first `npre` rows of a residual tile are copied there while the loop runs.
    const int K0 = p.K;
    const int sb_off = (wl_off + 9 * K0 * 4 + 15) & ~15;
    const int lb = (p.lds_total - (sb_off + WN * RN * 32 * 8)) & ~15;
    unsigned char* smem = smem0 + lb;
    const bool res_dma = p.residual != nullptr;    // rows of up to 128 pieces: Nstore <= 1024 (launcher-checked)
    // rows [0, npre) lie below the loop's buffers; [npre, n1) also cover raw[0], free from iteration NC-2 on, and
    // [n1, n2) raw[1], free in the last iteration (NC even: raw[0] is then the buffer nobody refills)
    const int NCk = (p.K + KC - 1) / KC;
    int npre = 0, n1 = 0, n2 = 0;
    if (res_dma) {
        const int S = stage_stride<bf16_t>(p.Nstore);
        npre = min(min(lb / S, MT), max(NCk - 2, 0) * (NT / 64));
        n1 = n2 = npre;
        if (NCk % 2 == 0 && NCk >= 2) {
            n1 = max(npre, min((lb + raw_bytes) / S, MT));
            n2 = max(n1, min((lb + 2 * raw_bytes) / S, MT));
        }
    }

    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int m0 = tile * MT;
    const int p_lo = m0 - (W + 1);
    const int K = p.K;                             // padded input channels (multiple of 16)
    const int KB = K / 16;
    const int NC = (K + KC - 1) / KC;
    const h16_t* __restrict__ in = reinterpret_cast<const h16_t*>(p.in);
    const int ldi = p.ldi;

    const int jch = tid & (CPR - 1);               // this thread's 16-byte piece (8 channels)

    // ---- prologue: every global load the first stages need is issued up front (halo chunk 0, the
    // first B fragments, then the taps), so their latencies overlap instead of adding up
    // halo rows by LDS-DMA (no registers, no staging pass) where the layer has no ReLU in front.  256-wide
    // instances only: 256->256 @74x74 0.73 -> 0.67 ms; the 728-wide ones measured 1-2 % slower with it
    constexpr bool HDMA = !RELU;
    constexpr int HSTEP = WN * WM / 2;             // issued by the D-first half of the waves only
    Raw3 rreg;
    if constexpr (HDMA) {
        halo_dma_asm(in, ldi, 0, K, p_lo, p.M, HP, tid & 63, __builtin_amdgcn_readfirstlane(tid >> 6), NT / 64,
                     (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem0 + (unsigned)lb);
    } else {
        rreg = raw_load<NT, NRAW>(in, ldi, 0, K, jch, tid, p_lo, p.M);
    }
    constexpr int nfp32 = WN * RN * 32;
    constexpr int NSB = (nfp32 + NT - 1) / NT;
    float sbv[NSB][2];
    {
        const float* scp = p.scale ? p.scale : p.dw;       // any readable floats: the value is discarded if null
        const float* bip = p.bias ? p.bias : p.dw;
#pragma unroll
        for (int q = 0; q < NSB; ++q) {
            const int i = tid + q * NT;
            const int ic = i < p.Nstore ? i : 0;
            sbv[q][0] = scp[ic];
            sbv[q][1] = bip[ic];
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int nfb = wn * RN;                       // single pass over N: NFp == WN*RN
    const bool first_half = __builtin_amdgcn_readfirstlane(wave) < WN * WM / 2;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem0;
    const unsigned lds_raw0 = lds_base + (unsigned)lb;
    const uint4* __restrict__ wp = reinterpret_cast<const uint4*>(p.wp);
    const uint4* bp0 = wp + ((size_t)nfb * p.KBtot + p.kb0) * 64 + lane;
    uint4 bq[PF][RN];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int idx = d < KB ? d : KB - 1;
#pragma unroll
        for (int j = 0; j < RN; ++j) bq[d][j] = bp0[((size_t)j * p.KBtot + idx) * 64];
    }
    // depthwise taps: [9][K] fp32, all loads of a thread in flight at once
    constexpr int NTAP = (9 * 768 / 4 + NT - 1) / NT;      // float4 pieces per thread (K <= 768)
    float4 tapv[NTAP];
#pragma unroll
    for (int q = 0; q < NTAP; ++q) {
        const int i = (tid + q * NT) * 4;
        const int ic = i < 9 * K ? i : 0;
        int off = ic;                              // rows are contiguous when ldi == K (the usual case)
        if (ldi != K) { const int t = ic / K; off = t * ldi + (ic - t * K); }
        tapv[q] = *reinterpret_cast<const float4*>(p.dw + off);
    }
    // per-thread constants of the depthwise stage (independent of the chunk); the integer divisions run
    // while the loads above are in flight
    unsigned item_mask[NITEM];                     // 9 validity bits ('same' zero padding)
#pragma unroll
    for (int q = 0; q < NITEM; ++q) {
        const int r = (tid + q * NT) >> 3;
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

#pragma unroll
    for (int q = 0; q < NTAP; ++q) {
        const int i = (tid + q * NT) * 4;
        // stored as (w0,w2,w1,w3): the order the packed-FMA depthwise consumes them in
        if (i < 9 * K)
            *reinterpret_cast<float4*>(smem + wl_off + i * 4) = make_float4(tapv[q].x, tapv[q].z, tapv[q].y, tapv[q].w);
    }
    // folded-BN scale and bias -> LDS behind everything the epilogue's staging tile will overwrite (loaded at the
    // top of the prologue, unconditionally, so their latency runs under the halo loads')
    float* sb = reinterpret_cast<float*>(smem + sb_off);
#pragma unroll
    for (int q = 0; q < NSB; ++q) {
        const int i = tid + q * NT;
        if (i < nfp32) {
            sb[i] = (p.scale && i < p.Nstore) ? sbv[q][0] : 1.f;
            sb[nfp32 + i] = (p.bias && i < p.Nstore) ? sbv[q][1] : 0.f;
        }
    }

    // raw(0) -> LDS, D(0), raw(1) -> LDS, raw(2) in flight
    if constexpr (HDMA) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's share of raw[0] has landed
    } else {
        raw_store<NT, RELU, NRAW>(rreg, smem, 0, jch, tid, p_lo, HP, p.M);
        rreg = raw_load<NT, NRAW>(in, ldi, 1, K, jch, tid, p_lo, p.M);
    }
    __syncthreads();                               // raw[0] and the taps are visible
    if constexpr (HDMA) halo_dma_asm(in, ldi, 1, K, p_lo, p.M, HP, lane, wave_u, NT / 64, lds_raw0 + raw_bytes);
    depthwise<T, NT, MT, NITEM>(smem, 0, a_off0, wl_off, 0, K, W, jch, tid, item_mask);
    if constexpr (HDMA) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        raw_store<NT, RELU, NRAW>(rreg, smem, raw_bytes, jch, tid, p_lo, HP, p.M);
        rreg = raw_load<NT, NRAW>(in, ldi, 2, K, jch, tid, p_lo, p.M);
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
        const int cur = c & 1, nxt = cur ^ 1;
        // L: raw chunk c+2 (loaded during the previous iteration) -> raw[cur], whose last reader D(c)
        // finished before the previous barrier
        if constexpr (!HDMA) {
            if (c + 2 < NC) raw_store<NT, RELU, NRAW>(rreg, smem, cur * raw_bytes, jch, tid, p_lo, HP, p.M);
        }
        // D (depthwise of chunk c+1, vector ALU) and G (matrix cores on chunk c) are independent.
        // Each SIMD hosts one wave of each half of the workgroup: run them in opposite order so
        // one wave's vector-ALU stage overlaps its partner's matrix-core stage.  The halo loads of
        // chunk c+3 go right in front of the wave's own D stage (which never waits on vmcnt): in front
        // of G they would sit ahead of the B ring's loads in the in-order vmcnt queue and the first
        // k-blocks would wait out their HBM latency.
        const int a_base = a_off0 + cur * MT * A_STR + (wm * MF * 32 + r32) * A_STR + h * 16;
        const bool do_d = c + 1 < NC;
        const int dma_lo = c == NC - 2 ? npre : c == NC - 1 ? n1 : c * (NT / 64);
        const int dma_hi = c == NC - 2 ? n1 : c == NC - 1 ? n2 : min(npre, (c + 1) * (NT / 64));
        if (first_half) {
            if constexpr (HDMA) {
                // chunk c+2 -> raw[cur] (its last reader, D(c), finished before the previous barrier); only the
                // waves that run D first issue these: in front of G they would delay the B ring (in-order vmcnt)
                if (c + 2 < NC)
                    halo_dma_asm(in, ldi, c + 2, K, p_lo, p.M, HP, lane, wave_u, HSTEP, lds_raw0 + cur * raw_bytes);
            } else {
                rreg = raw_load<NT, NRAW>(in, ldi, c + 3, K, jch, tid, p_lo, p.M);
            }
            for (int r = dma_lo + wave_u; r < dma_hi; r += NT / 64) residual_dma_row_asm<NT, MT>(p, m0, tid, lds_base, r);
            if (do_d)
                depthwise<T, NT, MT, NITEM>(smem, nxt * raw_bytes, a_off0 + nxt * MT * A_STR, wl_off, c + 1, K, W, jch,
                                         tid, item_mask);
        }
        if constexpr (RN == 3) {
            mma_chunk_pre<T, MF, RN, PF, KBC>(acc, bq, smem, a_base, bp0, c, KB, p.KBtot);
        } else {
            mma_chunk<T, MF, RN, PF, KBC>(acc, bq, smem, a_base, bp0, c, KB, p.KBtot);
        }
        if (!first_half) {
            if constexpr (!HDMA) {
                rreg = raw_load<NT, NRAW>(in, ldi, c + 3, K, jch, tid, p_lo, p.M);
            }
            for (int r = dma_lo + wave_u; r < dma_hi; r += NT / 64) residual_dma_row_asm<NT, MT>(p, m0, tid, lds_base, r);
            if (do_d)
                depthwise<T, NT, MT, NITEM>(smem, nxt * raw_bytes, a_off0 + nxt * MT * A_STR, wl_off, c + 1, K, W, jch,
                                         tid, item_mask);
        }
        if constexpr (HDMA) {
            // the halo DMA is older than the PF*RN ring loads this wave's G stage left in flight.  (A register
            // spill reload inside that stage would be one more younger vector-memory operation and make this count
            // wrong: `make` fails if an instance with the DMA path spills -- csrc/check_spills.sh.  vmcnt(0) here
            // measured 3-4 % slower on the 728-wide layers.)
            if (first_half) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PF * RN) : "memory");
        }
        __syncthreads();
    }
    // every wave is past its last LDS read (the loop's closing barrier): reuse LDS as the
    // output staging tile
    {
        if (res_dma) {
            residual_dma<NT, MT>(p, m0, tid, smem0, n2, MT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the in-loop asm copies too
            __syncthreads();                       // (waits for this wave's DMA, then for everyone's)
            pipe_epilogue_to_lds<T, MF, RN>(p, acc, nfb, wm * MF * 32, r32, h, smem0, sb, nfp32, true);
        } else {
            pipe_epilogue_to_lds<T, MF, RN>(p, acc, nfb, wm * MF * 32, r32, h, smem0, sb, nfp32, false);
        }
        __syncthreads();
        lds_rows_to_global<bf16_t, NT, MT>(p, m0, tid, smem0);
    }
}

template <typename T, bool RELU, int RN, int NRAW, int WM>
int launch_pipe(const GemmParams& p, hipStream_t s) {
    constexpr int MF = 3, WN = 8;
    auto kern = sepconv_pipe_kernel<T, RELU, MF, WN, RN, NRAW, WM>;
    const int MT = 32 * MF * WM;
    const int HP = MT + 2 * (p.W + 1);
    size_t lds = (size_t)2 * HP * RAW_ROW + 2 * MT * A_STR + (size_t)9 * p.K * 4;
    const size_t stage = (size_t)MT * (p.Nstore * 2 + 16);
    if (stage > lds) lds = stage;
    lds = ((lds + 15) & ~(size_t)15) + (size_t)p.NFp * 32 * 8 + 16;      // + scale and bias
    // one workgroup per CU anyway (2 waves per SIMD): take all of LDS so a residual tile can be prefetched
    if (p.residual && WM == 1 && RN == 3 && lds <= 160 * 1024) lds = 160 * 1024;
    if (p.NFp != WN * RN || p.K % 16 != 0 || p.K > 768 || p.Nstore % 8 != 0 || p.Nstore > 1024 || HP * CPR > NRAW * 64 * WN * WM || lds > 160 * 1024 || p.k_off != 0)
        return (int)hipErrorInvalidValue;
    static BqLdsAttr attr;
    if (const int e = attr.ensure(reinterpret_cast<const void*>(kern), lds)) return e;
    const int grid = (p.M + MT - 1) / MT;
    GemmParams q = p;
    q.lds_total = (int)lds;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WN * WM), lds, s, q);
    return (int)hipGetLastError();
}

}  // namespace

// bf16 SeparableConv2D whose (padded) output width is 768 (three fragments per wave, maps up to
// 37x37) or 256 (one fragment per wave, maps up to 74x74).
static int pipe_variant(int nfp, int W) {
    const int HP = 96 + 2 * (W + 1);
    if (nfp == 24 && HP * CPR <= 3 * 512) return 0;
    if (nfp == 8 && HP * CPR <= 4 * 512) return 1;
    return -1;
}

bool pipe_supported(int dtype, int prod, int nfp, int W, int K) {
    if (dtype == 0 || (prod != PROD_DW && prod != PROD_DW_RELU) || K % 16 != 0) return false;
    if (pipe_variant(nfp, W) < 0) return false;
    const int HP = 96 + 2 * (W + 1);
    const size_t lds = (size_t)2 * HP * RAW_ROW + 2 * 96 * A_STR + (size_t)9 * K * 4;
    const size_t sbb = (size_t)nfp * 32 * 8 + 16;
    return lds + sbb <= 160 * 1024 && (size_t)96 * (nfp * 64 + 16) + sbb <= 160 * 1024;
}

namespace {
template <typename T>
int launch_sepconv_pipe_t(int prod, const GemmParams& p, hipStream_t s) {
    const bool relu = prod == PROD_DW_RELU;
    if (pipe_variant(p.NFp, p.W) == 0) return relu ? launch_pipe<T, true, 3, 3, 1>(p, s) : launch_pipe<T, false, 3, 3, 1>(p, s);
    // 256-wide: 192-row tiles on 16 waves where the halo of a 96-row tile is larger than the tile itself
    static const bool no_wide = bq_exp_env("BQ_PIPE_NO_WM2") != nullptr;
    const size_t lds2 = (size_t)2 * (192 + 2 * (p.W + 1)) * RAW_ROW + 2 * 192 * A_STR + (size_t)9 * p.K * 4;
    if (!no_wide && p.W >= 48 && lds2 + (size_t)p.NFp * 32 * 8 + 16 <= 160 * 1024 && (192 + 2 * (p.W + 1)) * CPR <= 3 * 1024)
        return relu ? launch_pipe<T, true, 1, 3, 2>(p, s) : launch_pipe<T, false, 1, 3, 2>(p, s);
    return relu ? launch_pipe<T, true, 1, 4, 1>(p, s) : launch_pipe<T, false, 1, 4, 1>(p, s);
}
}  // namespace

int launch_sepconv_pipe(int dtype, int prod, const GemmParams& p, hipStream_t s) {
    return dtype == 2 ? launch_sepconv_pipe_t<f16_t>(prod, p, s) : launch_sepconv_pipe_t<bf16_t>(prod, p, s);
}
