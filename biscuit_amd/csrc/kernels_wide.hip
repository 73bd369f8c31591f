// SeparableConv2D from block 3 to block 13: the 25 layers 728 -> 728 on 19x19 maps (~75 % of the network's FLOPs),
// block4_sepconv2 (728 -> 728, 37x37), block4_sepconv1 (256 -> 728, 37x37) and block 3's two layers (128 / 256 -> 256,
// 74x74: the narrow instance, see NPlan / BPRE); 16-bit storage (bf16 or f16).  The description is the 19x19 instance's; the
// others differ in the constants Geo, KPlan and NPlan derive (rows per tile, chunks, fragments per wave).
//
// One PERSISTENT workgroup per CU (round 3) = 8 waves (two per SIMD, 256 registers each).  It walks image-aligned tiles
// of 4 map rows (76 pixels, padded to 80 MFMA rows; the 5th tile of an image has 3 rows: 1 280 tiles per batch of 256,
// five per workgroup) and computes ALL 768 (padded) output channels of a tile; wave w keeps the 80 x 96 fp32 accumulator
// block of channels [96w, 96w+96) in the accumulator file: 5 x 6 tiles of v_mfma_f32_16x16x32 = 120 registers.
// The contraction is walked in 64-channel chunks, one workgroup barrier per chunk, and inside a chunk EVERY wave runs
// one instruction stream that carries all three stages, interleaved instruction by instruction (a wave issues in
// order: five MFMAs in a row hold it for 80 cycles with the vector ALU idle, 25 vector instructions in a row leave the
// matrix pipe idle for 100), and the SIMD's second wave fills what one stream leaves open:
//     G(c)    60 MFMAs: A fragments from the LDS chunk buffer, B fragments (weights, host-packed in 16x16x32
//             fragment order) reloaded for the next k-step as soon as its 5 MFMAs are issued
//     D(c+1)  the depthwise 3x3 of the NEXT chunk on the vector ALU: lane = (channel pair, run of 5 pixels of a tile
//             row), a 3x3 sliding window in registers, per pixel 3 ds_read_b32, 18 multiply-adds (bf16: + 6 unpack;
//             f16: v_fma_mix_f32 straight from the packed dword), 1 packed convert, 1 ds_write_b32 -- no per-tap masks
//             (the halo image in LDS has zero columns left and right of every row and zero rows outside the map), no
//             packed-f32 arithmetic (v_pk_fma_f32 halves the instruction count and measured 10 % SLOWER)
//     L(c+2)  the halo image of chunk c+2 by LDS-DMA (global_load_lds_dwordx4: no registers)
// The chunk sequence RUNS ON INTO THE NEXT TILE: iteration NCH-2 fetches the next tile's halo chunk 0, iteration NCH-1
// its chunk 1 while the depthwise stage builds its first A chunk and the weight fragments of its first k-step are
// loaded -- a tile has no prologue of its own (round 2: 8 k of a tile's 60 k cycles).  The EPILOGUE touches no LDS the
// loop uses: the host interleaves the two 16-wide n-fragments of a pair (weights.py: frag16_channel) so that a lane
// holds eight CONSECUTIVE channels of its pixel over the pair, i.e. 16-byte stores straight from the accumulators and
// 16-byte loads of the residual -- rows 0..31 of the residual tile prefetched into LDS by DMA under the K loop, rows
// 32..79 loaded at the start of the epilogue, EVERY load before the first store (vmcnt retires in order and a store
// retires ~1.8 k cycles after issue: a load behind a store cannot be consumed earlier; the first form of this epilogue,
// loads and stores alternating, took 27 k cycles per tile instead of 11 k).
// Measured (in-kernel stamps, tools/stamps_wide.py): chunks 3.0-3.3 k cycles each (the MFMAs alone: 1.92 k), the two
// chunks after an epilogue 4-5 k (their weight loads queue behind its stores), epilogue 4 k without / 11 k with a
// residual input; 51-55 k per tile against 58-66 k in round 2 -- 0.160 -> 0.148 ms per layer, bit-identical results.
// Tap order and fp32 accumulation of the depthwise stage are those of every other producer in this library.
#include "gemm_common.h"

#include <stdio.h>

#include <type_traits>
#include <vector>

#ifndef WIDE_ABLATE
#define WIDE_ABLATE 0
#endif
#ifndef WIDE_ZERO_PAD
#define WIDE_ZERO_PAD 1
#endif

namespace {
using namespace bqk;

// Output side of an instance: NOUT stored channels per pixel (736: the layers that produce 728; 256: block3_sepconv2) on 8
// waves of RN 16-wide n-fragments each
template <int NOUT>
struct NPlan {
    static constexpr int NP = NOUT;                      // row stride of the output (and of a residual input)
    static constexpr int RN = (NOUT + 127) / 128;        // n-fragments per wave: 6 (768 padded channels) or 2
    static constexpr int NFT = 8 * RN, CPW = RN * 16;    // n-fragments in all, output channels per wave
    static_assert(RN % 2 == 0, "n-fragments are stored as interleaved pairs");
};
// Input side of an instance: KIN padded input channels (736: the 728 -> 728 layers; 256: block4_sepconv1), walked in
// 64-channel chunks -- an even number of them, the last of 32 or 64 channels
template <int KIN>
struct KPlan {
    static constexpr int KP = KIN;
    static constexpr int KST = KIN / 32;                 // k-steps of 32 (one v_mfma_f32_16x16x32 deep)
    static constexpr int NCH = (KIN + 63) / 64;          // chunks
    static constexpr int LASTK = (KIN - (NCH - 1) * 64) / 32;   // k-steps of the last chunk: 1 or 2
    static_assert(KIN % 32 == 0 && NCH % 2 == 0 && NCH >= 2 && (LASTK == 1 || LASTK == 2), "chunk plan");
};
constexpr int KC = 64;                  // channels per chunk
constexpr int A_STR = KC * 2 + 32;      // 160 B = 10 slots of 16 B: the 16x16x32 fragment read (lane -> row l&15, 16-byte
                                        // k-group l>>4) is conflict-free for ds_read_b128's lane groups iff slots/row = 2 (mod 4)
constexpr int WN = 8;                   // waves
constexpr int RES_ROWS = 32;             // residual rows that go through LDS (row fragments 0 and 1)
constexpr int RES_PPR = NPlan<736>::CPW * 2 / 16 + 1; // 13 pieces of 16 B per row: odd (only 736-wide layers have a residual input)
constexpr int RES_STR = RES_PPR * 16;
constexpr int NRES = (RES_ROWS * RES_PPR + 63) / 64;   // LDS-DMA instructions per wave: 7, one per chunk

// Geometry of one instance: IW x IW maps, TR map rows per tile; KTAPS / NSB: the widest input / padded output the instance's
// tap and folded-BN tables in LDS have to hold.
template <int IW_, int TR_, int KTAPS = 736, int NSB = 768>
struct Geo {
    static constexpr int IW = IW_, IH = IW_, TR = TR_;
    static constexpr int MT = (TR * IW + 15) / 16 * 16;      // MFMA rows per tile: 80 (76 or 74 pixels) or 160 (148)
    static constexpr int MF = MT / 16;                       // 16-row m-fragments
    static constexpr int A_BYTES = (MT + 1) * A_STR;         // the row behind the last takes the results of pixels past the end of a map row
    static constexpr int TAPS_BYTES = 9 * KTAPS * 4;         // depthwise taps
    static constexpr int SB_HALF = NSB * 4, SB_BYTES = 2 * SB_HALF;   // folded BN: scales, then biases
    static constexpr int TPI = (IH + TR - 1) / TR;           // tiles per image
    static constexpr int PW = IW + 2;                        // slots per row of the padded halo image
    static constexpr int NSLOT = (TR + 2) * PW;              // slots of 128 B (64 channels of one pixel)
    static constexpr int NDMA = (NSLOT + 7) / 8;             // LDS-DMA instructions (1 KiB each) that cover the image
    static constexpr int HPW = (NDMA + WN - 1) / WN;         // ... per wave
    static constexpr int RAW_BYTES = HPW * WN * 1024;
    static constexpr int SEG = 16 / TR;                      // depthwise runs per tile row (16 lane groups in all)
    static constexpr int NSTEP = (IW + SEG - 1) / SEG;       // pixels per run: 5 (19- and 37-wide maps) or 10 (74-wide)
    // LDS map: the loop's buffers and the two tables.  Round 3: there is NO staging tile any more -- the epilogue goes
    // from the accumulators straight to HBM (see the kernel), so nothing of a tile's end touches LDS and the next
    // tile's halo, taps and first A chunk can be in place before the current tile's epilogue starts.
    // (A chunks, taps and tables first: every address the K loop forms from them is a per-lane base plus an offset that
    // fits the 16-bit immediate of ds_*; the halo images come last)
    static constexpr int OFF_A = 0;
    static constexpr int OFF_TAPS = OFF_A + 2 * A_BYTES;
    static constexpr int OFF_SB = OFF_TAPS + TAPS_BYTES;
    static constexpr int OFF_RAW = OFF_SB + SB_BYTES;        // 2 halo images (16-bit, as they arrive)
    // the first RES_ROWS rows of a tile's residual input, per wave its 96 channels (RES_STR: 12 pieces + 1 of padding),
    // copied in by LDS-DMA while the K loop runs; nothing else ever uses this region
    static constexpr int OFF_RES = OFF_RAW + 2 * RAW_BYTES;
    static constexpr int lds_bytes(bool res) { return OFF_RES + (res ? WN * RES_ROWS * RES_STR : 0); }
    static_assert(OFF_RAW < 65536, "immediate offsets of the A / tap / table accesses");
    static_assert(lds_bytes(IW_ == 19) <= 160 * 1024, "LDS budget (only the 19x19 layers have a residual input)");
    static_assert(SEG * NSTEP >= IW && SEG * TR == 16 && TR * IW <= MT, "depthwise runs do not cover the tile");
};

typedef unsigned short h16_t;   // either 16-bit storage type (the kernel only forms byte addresses from these)

struct WideParams {
    const h16_t* in;        // [n*361][736]
    const uint4* wp;        // pointwise weights in 16x16x32 fragment order [23 k-steps][48][64] x 16 B
    const float* dw;        // [9][736] depthwise taps
    const float* scale;     // [768] folded BN
    const float* bias;      // [768]
    const h16_t* residual;  // [n*361][736] or null
    h16_t* out;             // [n*361][736]
    int n;                  // images
    int relu;               // ReLU in the epilogue
    int nwg;                // persistent workgroups launched (a multiple of 8)
#ifdef BQ_EXPERIMENTS
    unsigned long long* stamps;   // s_memtime stamps [64 blocks from stamp_b0][wave][32] (diagnostic builds only)
    unsigned stamp_b0;
#endif
};

#ifdef BQ_EXPERIMENTS
constexpr int STAMP_TILES = 8;        // tiles of a workgroup that get a row of 32 stamps each
#define WSTAMP(ev) do { if (stp && stamp_it < STAMP_TILES) stp[stamp_it * 32 + (ev)] = __builtin_amdgcn_s_memtime(); } while (0)
// the 100 MHz real-time counter next to the cycle counter: events 30 (tile start) and 31 (tile end) -> the in-kernel clock
#define WSTAMP_RT(ev) do { if (stp && stamp_it < STAMP_TILES) stp[stamp_it * 32 + (ev)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WSTAMP(ev) do { } while (0)
#define WSTAMP_RT(ev) do { } while (0)
#endif

// One LDS-DMA instruction: lanes in `mask` copy 16 bytes each from sbase + voff to LDS at lds_dst + 16*lane.
// Inline asm on purpose: a DMA the compiler can see makes it wait vmcnt(0) in front of every later LDS read.
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_dst, unsigned long long mask) {
    unsigned long long keep_exec;
    unsigned keep_m0;
    asm volatile(
        "s_mov_b64 %0, exec\n\t"
        "s_mov_b32 %1, m0\n\t"
        "s_and_b64 exec, exec, %5\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %2\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(keep_exec), "=&s"(keep_m0)
        : "s"(sbase), "v"(voff), "s"(lds_dst), "s"(mask)
        : "memory", "scc");
}

__device__ __forceinline__ unsigned relu2(unsigned x) { return relu_pk16(x); }   // ReLU on two packed bf16 / f16

typedef float f32x4v __attribute__((ext_vector_type(4)));

// ---- the depthwise stage as a list of micro-operations -----------------------------------------------------------------
// A wave issues in order, so the stage is cut into micro-operations of two or three vector instructions (or three LDS
// reads) that the K loop issues BETWEEN consecutive MFMAs.  Per chunk (NDW micro-operations):
//     0..2              taps: 3 x 3 ds_read_b64
//     3, 4, 5           packed dwords (the lane's two channels) of window columns -1, 0, +1: 3 ds_read_b32 each
//     6, 7, 8           unpack them
//     9 + 14 s + k      pixel s = 0..4 of the lane's run:
//                         k = 0..6    tap t = k (row t/3, column t%3): two v_fma_f32, accumulation from zero in the
//                                     tap order of every other depthwise producer of this library
//                         k = 7       dwords of column s+2 (for pixel s+1)
//                         k = 8, 9    taps 7, 8
//                         k = 10      v_cvt_pk_bf16_f32 + ds_write_b32 into the A chunk
//                         k = 11..13  unpack column s+2, row k-11, into the registers of column s-1 (dead after tap 6)
constexpr int NPRIME = 9;
constexpr int ndw(int nstep) { return NPRIME + 14 * nstep; }

// The window holds what the taps multiply: bf16 -> the two channels unpacked to fp32 (shift / mask, two instructions per
// dword); f16 -> the packed dword itself, and each tap is ONE v_fma_mix_f32 that takes its half straight out of it (the
// product of the exact values, one rounding: bit-identical to convert-then-fma), so the six unpack instructions per
// pixel do not exist in the f16 instance (a ReLU in front is one v_pk_max_i16 per dword in both).
template <typename T> struct Win;
template <> struct Win<bf16_t> { typedef float2 type; };
template <> struct Win<f16_t> { typedef unsigned type; };

template <typename T>
struct DwState {
    float2 tw[9];
    typename Win<T>::type c[3][3];   // [window column slot][row]: column j of the lane's run lives in slot (j + 1) % 3
    unsigned d[3];          // dwords of the column being brought in
    unsigned x[2][3];       // dwords of columns -1, 0 (prime only)
    float o0, o1;
};

template <typename T, bool RELU>
__device__ __forceinline__ typename Win<T>::type unpack2(unsigned d) {
    if (RELU) d = relu2(d);
    if constexpr (H16<T>::F16) return d;
    else return make_float2(__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u));
}

// one tap of the lane's two channels: o = tw * window (+ o)
template <typename T, bool FIRST>
__device__ __forceinline__ void tap2(float& o0, float& o1, const float2& tw, const typename Win<T>::type& w) {
    if constexpr (H16<T>::F16) {
        // plain C on purpose: hipcc selects v_fma_mix_f32 for fma(fpext(half of a dword), f32, f32) by itself, and -- unlike
        // behind an inline-asm v_fma_mix -- does not pad the next instruction that reads the result with an s_nop (it takes
        // every asm result for a partial-register write: one s_nop per tap, 70 per chunk in the 10-pixel runs of the 74x74
        // instance)
        typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
        const f16x2v h = __builtin_bit_cast(f16x2v, w);
        o0 = __builtin_fmaf((float)h.x, tw.x, FIRST ? 0.f : o0);
        o1 = __builtin_fmaf((float)h.y, tw.y, FIRST ? 0.f : o1);
    } else {
        o0 = fmaf(tw.x, w.x, FIRST ? 0.f : o0);
        o1 = fmaf(tw.y, w.y, FIRST ? 0.f : o1);
    }
}

template <int PW>
__device__ __forceinline__ unsigned raw_dword(const unsigned char* smem, int raw_addr, int j, int r) {
    return *reinterpret_cast<const unsigned*>(smem + raw_addr + ((j + 1) + r * PW) * 128);
}

// where pixel s of a lane's run writes its result: base + s * A_STR (the offset folds into the ds_write), or one of the NT
// explicit addresses of the run's last steps (see OOB0 in the kernel)
template <int NT, int NSTEP>
struct AwAddr {
    int base;
    int dump;                       // the lane's slot in row 80
    unsigned long long last;        // lanes whose run is the last of its map row (steps >= NSTEP - NT lie past the row)
    template <int S> __device__ __forceinline__ int at() const {
        if constexpr (S >= NSTEP - NT) {
            int r;                  // a select on a scalar lane mask: no per-lane register per step kept across the K loop
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(base + S * A_STR), "v"(dump), "s"(last));
            return r;
        } else return base + S * A_STR;
    }
};

template <typename T, bool RELU, int PW, int KP, int NSTEP, int M, typename AW>
__device__ __forceinline__ void dw_op(DwState<T>& st, unsigned char* smem, int raw_addr, int tap_addr, const AW& aw) {
    if constexpr (M < 3) {
#pragma unroll
        for (int t = 3 * M; t < 3 * M + 3; ++t) st.tw[t] = *reinterpret_cast<const float2*>(smem + tap_addr + t * KP * 4);
    } else if constexpr (M < 5) {
#pragma unroll
        for (int r = 0; r < 3; ++r) st.x[M - 3][r] = raw_dword<PW>(smem, raw_addr, M - 4, r);
    } else if constexpr (M == 5) {
#pragma unroll
        for (int r = 0; r < 3; ++r) st.d[r] = raw_dword<PW>(smem, raw_addr, 1, r);
    } else if constexpr (M < 8) {
#pragma unroll
        for (int r = 0; r < 3; ++r) st.c[M - 6][r] = unpack2<T, RELU>(st.x[M - 6][r]);
    } else if constexpr (M == 8) {
#pragma unroll
        for (int r = 0; r < 3; ++r) st.c[2][r] = unpack2<T, RELU>(st.d[r]);
    } else {
        constexpr int S = (M - NPRIME) / 14, K = (M - NPRIME) % 14;
        if constexpr (K == 7) {
            if constexpr (S + 1 < NSTEP) {
#pragma unroll
                for (int r = 0; r < 3; ++r) st.d[r] = raw_dword<PW>(smem, raw_addr, S + 2, r);
            }
        } else if constexpr (K < 10) {
            constexpr int TP = K < 7 ? K : K - 1, R = TP / 3, DX = TP % 3;
            tap2<T, TP == 0>(st.o0, st.o1, st.tw[TP], st.c[(S + DX) % 3][R]);
        } else if constexpr (K == 10) {
            *reinterpret_cast<unsigned*>(smem + aw.template at<S>()) = H16<T>::pack2(st.o0, st.o1);
        } else {
            if constexpr (S + 1 < NSTEP) st.c[S % 3][K - 11] = unpack2<T, RELU>(st.d[K - 11]);
        }
    }
}

template <typename T, bool RELU, int PW, int KP, int NSTEP, int LO, int HI, typename AW>
__device__ __forceinline__ void dw_ops(DwState<T>& st, unsigned char* smem, int raw_addr, int tap_addr, const AW& aw) {
    if constexpr (LO < HI) {
        dw_op<T, RELU, PW, KP, NSTEP, LO>(st, smem, raw_addr, tap_addr, aw);
        dw_ops<T, RELU, PW, KP, NSTEP, LO + 1, HI>(st, smem, raw_addr, tap_addr, aw);
    }
}

// Micro-operations issued before MFMA slot q of a chunk of NSLOTQ MFMAs per wave (60, or 40): the taps and the first columns
// go out with the first MFMAs, their unpacking waits a few slots (an LDS round trip), the pixels are spread over the rest.
template <int NSLOTQ, int NDW>
constexpr int dw_before(int q) {
    if (q <= 0) return 0;
    if (q < 4) return 2 * q > 6 ? 6 : 2 * q;
    if (q < 7) return 6;
    if (q < 9) return 6 + 2 * (q - 6) > NPRIME ? NPRIME : 6 + 2 * (q - 6);
    const int done = NPRIME + ((q - 8) * (NDW - NPRIME) + (NSLOTQ - 10)) / (NSLOTQ - 9);
    return done > NDW ? NDW : done;
}
static_assert(dw_before<60, ndw(5)>(60) == ndw(5) && dw_before<60, ndw(5)>(8) == NPRIME, "micro-operation schedule");
static_assert(dw_before<40, ndw(10)>(40) == ndw(10) && dw_before<40, ndw(10)>(8) == NPRIME, "micro-operation schedule");
static_assert(dw_before<40, ndw(5)>(40) == ndw(5) && dw_before<20, ndw(5)>(20) == ndw(5), "micro-operation schedule");

// The MFMA as inline asm with the accumulator tied in place in the accumulator file ("+a"): written through the builtin,
// hipcc gives every v_mfma_f32_16x16x32_bf16 of this loop a destination other than its C operand and pays for it with
// ~450 v_accvgpr_read/write/mov per chunk pair.  `volatile` keeps the MFMAs in source order, which IS the schedule (the
// micro-operations above are written between them).  What hipcc then no longer does for us (it cannot see an MFMA in
// the string): (1) the wait states between the last MFMAs and the epilogue's first accumulator read -- explicit s_nops
// after the loop; (2) the wait states between a VALU write of an A/B operand register and the MFMA reading it -- the
// operands here come straight from ds_read / global_load, and `make` runs check_wide.py over the generated assembly
// to prove that no VALU instruction writes an operand within two instructions of its MFMA.
template <typename T>
__device__ __forceinline__ void mfma16(f32x4v& acc, const uint4& b, const uint4& a) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    if constexpr (H16<T>::F16)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0"
                     : "+a"(acc) : "v"(__builtin_bit_cast(u32x4, b)), "v"(__builtin_bit_cast(u32x4, a)));
    else
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                     : "+a"(acc) : "v"(__builtin_bit_cast(u32x4, b)), "v"(__builtin_bit_cast(u32x4, a)));
}
// The very first k-step of a tile: C is the constant 0, the accumulators need no initialisation (120 writes per wave).
template <typename T>
__device__ __forceinline__ void mfma16_first(f32x4v& acc, const uint4& b, const uint4& a) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    if constexpr (H16<T>::F16)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0"
                     : "=a"(acc) : "v"(__builtin_bit_cast(u32x4, b)), "v"(__builtin_bit_cast(u32x4, a)));
    else
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0"
                     : "=a"(acc) : "v"(__builtin_bit_cast(u32x4, b)), "v"(__builtin_bit_cast(u32x4, a)));
}

// The same with the weight fragment in the accumulator file (the narrow instance keeps a whole chunk's fragments there, see
// BPRE in the kernel)
typedef unsigned u32x4a __attribute__((ext_vector_type(4)));
template <typename T, bool FIRST>
__device__ __forceinline__ void mfma16_ab(f32x4v& acc, const u32x4a& b, const uint4& a) {
    if constexpr (H16<T>::F16) {
        if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(acc) : "a"(b), "v"(__builtin_bit_cast(u32x4a, a)));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "a"(b), "v"(__builtin_bit_cast(u32x4a, a)));
    } else {
        if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "a"(b), "v"(__builtin_bit_cast(u32x4a, a)));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(b), "v"(__builtin_bit_cast(u32x4a, a)));
    }
}

// An LDS base address the compiler must take as it is: offsets beyond the 16-bit immediate of ds_* would otherwise be
// re-associated into one base register per distinct offset (a dozen registers this kernel does not have).
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

template <int LO, int HI, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (LO < HI) {
        f(std::integral_constant<int, LO>{});
        static_for<LO + 1, HI>(f);
    }
}

// RES: the layer has a residual input (compile time: the epilogue's loads are then branch-free)
// LDO / NFTOT: row stride of the output and n-fragments of the layer's weight array when the launch produces a column SLICE
// of a wider layer (block13_sepconv2, 1 024 outputs = two launches of 512: the launcher offsets the pointers)
template <typename T, bool RELU, bool RES, typename G, int KIN, int NOUT = 736, int LDO = NOUT, int NFTOT = NPlan<NOUT>::NFT>
__global__ void __launch_bounds__(64 * WN) sepconv_wide_kernel(const WideParams p) {
    if constexpr (H16<T>::F16) bq_f16_saturate();
    constexpr int KP = KPlan<KIN>::KP, KST = KPlan<KIN>::KST, NCH = KPlan<KIN>::NCH, LASTK = KPlan<KIN>::LASTK;
    constexpr int NP = NPlan<NOUT>::NP, RN = NPlan<NOUT>::RN, NFT = NPlan<NOUT>::NFT, CPW = NPlan<NOUT>::CPW;
    constexpr int MT = G::MT, MF = G::MF, A_BYTES = G::A_BYTES, NSTEP = G::NSTEP, NDW = ndw(NSTEP), NSLOTQ = 2 * MF * RN;
    static_assert(9 * KP * 4 <= G::TAPS_BYTES && NFT * 16 * 4 <= G::SB_HALF && (!RES || (NOUT == 736 && LDO == NOUT)), "tables, residual");
    static_assert(!RES || NRES <= NCH - 2, "the residual prefetch needs one loop iteration of the tile's own per instruction");
    constexpr int IW = G::IW, IH = G::IH, TR = G::TR, TPI = G::TPI, PW = G::PW, NSLOT = G::NSLOT, HPW = G::HPW;
    constexpr int RAW_BYTES = G::RAW_BYTES, OFF_RAW = G::OFF_RAW, OFF_A = G::OFF_A, OFF_TAPS = G::OFF_TAPS, OFF_SB = G::OFF_SB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // Persistent: this workgroup walks the tiles xcd_tile(vb), vb = blockIdx.x, + nwg, + 2 nwg, ... (nwg is a multiple
    // of 8, so all of them come from the XCD's own run of neighbouring tiles)
    const int ntiles = p.n * TPI;
    const int nwg = gridDim.x;
    int vb = blockIdx.x;

#ifdef BQ_EXPERIMENTS
    // (a wave-uniform pointer: every lane stores the same stamp to the same address -- no per-lane registers)
    unsigned long long* stp = (p.stamps && blockIdx.x >= p.stamp_b0 && blockIdx.x < p.stamp_b0 + 64)
                                  ? p.stamps + ((blockIdx.x - p.stamp_b0) * WN + wave) * (32 * STAMP_TILES) : nullptr;
    int stamp_it = 0;
#endif
    WSTAMP(0);
    // ---- per-lane constants (none depends on the tile) ---------------------------------------------------------
    // halo image: piece P = (wave + 8 t) * 64 + lane of the padded image (piece = slot * 8 + 16-byte part), i.e. LDS-DMA
    // instruction wave + 8 t covers pieces [64 (wave + 8t), + 64); the same P is the piece this thread zeroes when its slot
    // lies outside the map.  The slot geometry is recomputed once per tile (divisions by constants) rather than kept in
    // registers across the K loop.
    const unsigned voff = lane * 16;                            // (also the per-lane offset of the weight loads)
    auto piece_geo = [&](int t, int& sy, unsigned& rel) {       // sy: slot row 0 .. TR+1, or -1 (pad column / past the image)
        const int P = (wave + WN * t) * 64 + (opaque((int)voff) >> 4);   // opaque: recomputed per tile, not hoisted and kept
        const int slot = P >> 3, piece = P & 7;
        const int y = slot / PW, sx = slot - y * PW;
        const int x = sx - 1;
        const bool col_ok = slot < NSLOT && (unsigned)x < (unsigned)IW;
        sy = col_ok ? y : -1;
        rel = col_ok ? (unsigned)((y * IW + x) * KP + piece * 8) * 2u : 0u;   // bytes from the tile's map row r0 - 1, column 0
    };
    const unsigned long long tail_lanes = 0x0F0F0F0F0F0F0F0Full;    // pieces 0..3: the 32 channels of the last chunk

    // depthwise: channel pair cp, lane group grp (0..15) = tile row grp / SEG, pixels [5 (grp % SEG), + 5) of it
    const int cp = lane & 31;
    const int grp = wave * 2 + (lane >> 5);
    const int drow = grp / G::SEG, dx0 = (grp - drow * G::SEG) * NSTEP;
    const int raw_lane = (drow * PW + dx0) * 128 + cp * 4;      // slot (row - 1, x0 - 1) of the padded image: column -1, row -1
    // A-chunk address of pixel s of the run: awb + s * A_STR, except where the run runs past the end of its map row (only
    // the last run of a row, from step OOB0 on): those results go to row 80, which nobody reads
    constexpr int OOB0 = IW - (G::SEG - 1) * NSTEP;             // first step of the last run that can lie past the row
    const int awb = (drow * IW + dx0) * A_STR + cp * 4;
    const unsigned long long last_run = __builtin_amdgcn_ballot_w64(dx0 + OOB0 >= IW);
    const int tap_lane = OFF_TAPS + cp * 8;

    // matrix stage: lane -> (row l&15 of a 16-row fragment, 16-byte k-group l>>4)
    const int r16 = lane & 15, kg = lane >> 4;
    const int nfb = wave * RN;
    // weights [k-step][48 n-fragments][64 lanes] x 16 B: a wave's 6 fragments of one k-step are 6 KiB in a row;
    // uniform base per k-step and 4 KiB (the immediate reaches 4 KiB) + one per-lane offset
    const unsigned char* __restrict__ wbase = reinterpret_cast<const unsigned char*>(p.wp) + (size_t)nfb * 1024;
    auto load_b = [&](int ks, int j) {
        // the base is uniform: hand it to the compiler as scalars so that the load takes the scalar-base form (no
        // 64-bit vector address arithmetic per fragment)
        const unsigned long long b = (unsigned long long)(wbase + (size_t)ks * (NFTOT * 1024) + (j >> 2) * 4096);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        typedef const __attribute__((address_space(1))) unsigned char* gptr_t;
        gptr_t sb = (gptr_t)(((unsigned long long)hi << 32) | lo);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = *(const __attribute__((address_space(1))) u32x4*)(sb + voff + (j & 3) * 1024);
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    const int a_lane = r16 * A_STR + kg * 16;
    const int ch0 = wave * CPW;

    // ---- per-tile state ------------------------------------------------------------------------------------------
    const unsigned char* inb = reinterpret_cast<const unsigned char*>(p.in);
    unsigned halo_off[HPW];                 // byte offsets of this thread's halo pieces (0 where masked)
    unsigned long long halo_mask[HPW];      // lanes of DMA instruction wave + 8t that lie inside the map
    unsigned zero_bits = 0;                 // bit t: piece t of this thread is a map-row slot outside the map (wants zeros)
    int t_m0 = 0, t_npix = 0;               // first global pixel and number of valid pixels of the tile being computed
    auto tile_setup = [&](int v, int& m0, int& npix) {
        const int tile = xcd_tile(v, ntiles);
        const int img = tile / TPI, part = tile - img * TPI;
        const int r0 = part * TR;                                   // first map row of the tile
        npix = (IH - r0 < TR ? IH - r0 : TR) * IW;                  // valid output pixels (76 or 57)
        m0 = img * (IH * IW) + r0 * IW;                             // their first global pixel index
        const unsigned base = (unsigned)(m0 - IW) * (unsigned)(KP * 2);   // map row r0 - 1 (wraps for r0 = 0: only used where ok)
        zero_bits = 0;
#pragma unroll
        for (int t = 0; t < HPW; ++t) {
            int sy; unsigned rel;
            piece_geo(t, sy, rel);
            const int row = r0 + sy - 1;
            const bool ok = sy >= 0 && (unsigned)row < (unsigned)IH;
            halo_off[t] = ok ? base + rel : 0u;
            halo_mask[t] = __builtin_amdgcn_ballot_w64(ok);
            zero_bits |= (sy >= 0 && !ok) ? (1u << t) : 0u;
        }
    };
    // chunk cc of the tile the halo constants describe -> raw[buf]; rows of the padded image that lie outside the map
    // are zeroed (a pad column never holds anything but the zeros written at the start)
    auto halo_dma = [&](int cc, int buf) {
        const unsigned long long tl = (LASTK == 1 && cc == NCH - 1) ? tail_lanes : ~0ull;
#pragma unroll
        for (int t = 0; t < HPW; ++t)
            dma16(inb + cc * (KC * 2), halo_off[t], lds0 + OFF_RAW + buf * RAW_BYTES + (wave + WN * t) * 1024, halo_mask[t] & tl);
    };
    auto halo_zero = [&](int buf) {
        const unsigned z = (unsigned)opaque(0);                 // (a zero vector kept across the K loop is four registers)
#pragma unroll
        for (int t = 0; t < HPW; ++t)
            if ((zero_bits >> t) & 1u)
                *reinterpret_cast<uint4*>(smem + OFF_RAW + buf * RAW_BYTES + (wave + WN * t) * 1024 + voff) = make_uint4(z, z, z, z);
    };

    // ---- prologue, once per workgroup: every load the first stages need goes out before anything waits --------
    tile_setup(vb, t_m0, t_npix);
    halo_dma(0, 0);
    {   // depthwise taps (26 496 B) and folded BN (2 x 3 072 B): plain copies, wave w takes instructions w, w+8, ...
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = wave + WN * t;
            const int byte = j * 1024 + lane * 16;
            dma16(p.dw, (unsigned)byte, lds0 + OFF_TAPS + j * 1024, __builtin_amdgcn_ballot_w64(byte < 9 * KP * 4));
        }
        const int k3 = wave < 3 ? wave : wave - 3;              // waves 0..2: 1 KiB of scale each, waves 3..5: of bias
        dma16(wave < 3 ? p.scale : p.bias, (unsigned)(k3 * 1024 + lane * 16), lds0 + OFF_SB + (wave < 3 ? 0 : G::SB_HALF) + k3 * 1024,
              __builtin_amdgcn_ballot_w64(wave < 6 && k3 * 1024 + lane * 16 < NFT * 16 * 4));
    }
    // BPRE (the narrow instance, 2 n-fragments per wave): ALL the weight fragments of a chunk (2 k-steps x 2) are fetched one
    // chunk ahead, at the top of the chunk together with the halo DMA, into the accumulator file (48 of its registers are
    // free here) by asm loads, and the only vector-memory wait of a chunk is the one that closes it.  Why: vmcnt retires in
    // order, so a fragment load issued BEHIND the halo DMA cannot be consumed before that DMA -- an HBM round trip of 3.3 k
    // cycles at this instance's 38 KB per chunk and workgroup -- has landed, and with fragments reloaded k-step by k-step
    // (the wide instances' scheme: no registers for more) every chunk's MFMAs and the depthwise stage interleaved with
    // them waited that trip out: 5.4-6.7 k cycles per chunk where DMA alone takes 3.3 k and the instructions ~3.5 k.
    constexpr bool BPRE = RN == 2;
    uint4 bq[RN];
    u32x4a bb[2][2][RN];
    auto load_bb = [&](int ks, u32x4a (&dst)[RN]) {
        const unsigned long long b = (unsigned long long)(wbase + (size_t)ks * (NFTOT * 1024));
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
        // (the base may just have been written by v_readfirstlane: a vector-memory instruction reads an SGPR 5 wait states
        // behind a VALU write of it, and hipcc does not pad in front of an asm statement)
        asm volatile("s_nop 4" :: "s"(sb));
#pragma unroll
        for (int j = 0; j < RN; ++j)
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=a"(dst[j]) : "v"(voff), "s"(sb), "n"(j * 1024) : "memory");
    };
    if constexpr (BPRE) {
        load_bb(0, bb[0][0]);
        load_bb(1, bb[0][1]);
    } else {
#pragma unroll
        for (int j = 0; j < RN; ++j) bq[j] = load_b(0, j);
    }
    halo_dma(1, 1);
    // zero what the DMA never writes: pad columns and slots past the image once and for all, map rows outside the map
    // for this tile (both halo buffers)
#pragma unroll
    for (int t = 0; t < HPW; ++t) {
        int sy; unsigned rel;
        piece_geo(t, sy, rel);
        if (sy < 0 || ((zero_bits >> t) & 1u)) {
            *reinterpret_cast<uint4*>(smem + OFF_RAW + (wave + WN * t) * 1024 + voff) = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(smem + OFF_RAW + RAW_BYTES + (wave + WN * t) * 1024 + voff) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
#if WIDE_ZERO_PAD
    // A rows past the tile's last pixel (76..79 of 80) are never written: whatever LDS held goes through the MFMAs of every k-step
    // and is thrown away.  Zeros cost the matrix pipe less energy than noise (the kernel runs against the power management).
    if constexpr (TR * IW < MT) {
        constexpr int NB = (MT - TR * IW) * A_STR / 16;          // 16-byte pieces per buffer
        for (int i = tid; i < 2 * NB; i += 64 * WN) {
            const int b = i / NB, k = i - b * NB;
            *reinterpret_cast<uint4*>(smem + OFF_A + b * A_BYTES + TR * IW * A_STR + k * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
#endif
    f32x4v acc[MF][RN];                                          // first written by the first k-step's MFMAs (C = 0)

    WSTAMP(1);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(HPW) : "memory");  // all but the DMAs of halo chunk 1 have landed
    WSTAMP(2);
    __syncthreads();
    WSTAMP(3);
    {   // D(0) of the first tile: nothing to overlap it with
        DwState<T> st;
        AwAddr<NSTEP - OOB0, NSTEP> aw0;
        aw0.base = opaque(awb) + OFF_A;
        aw0.dump = opaque(((tap_lane - OFF_TAPS) >> 1) + (OFF_A + MT * A_STR));
        aw0.last = last_run;
        dw_ops<T, RELU, PW, KP, NSTEP, 0, NDW>(st, smem, opaque(opaque(raw_lane) + OFF_RAW), opaque(tap_lane), aw0);
    }
    WSTAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // halo chunk 1
    __syncthreads();
    WSTAMP(5);

    // ---- K loop ------------------------------------------------------------------------------------------------
    // iteration c: G(c) on A[c & 1]; D(c+1) from raw[(c+1) & 1] into A[(c+1) & 1]; DMA of halo chunk c+2 into raw[c & 1].
    // Chunk indices run on into the NEXT tile of this workgroup: in iteration NCH-2 the DMA fetches the next tile's chunk 0,
    // in iteration NCH-1 its chunk 1 while D builds its first A chunk -- when a tile's epilogue starts, the next tile's
    // whole prologue is already in LDS.
    const unsigned char* resb = reinterpret_cast<const unsigned char*>(p.residual);
    constexpr bool has_res = RES;
    bool has_next = false;
    // residual rows [0, RES_ROWS) of the tile being computed -> this wave's private LDS rows, instruction j of NRES:
    // linear pieces [64 j, 64 j + 64), P = row * 13 + col (col 12 is padding)
    auto res_dma = [&](int j) {
        const int P = j * 64 + (int)(voff >> 4);
        const int row = P / RES_PPR, col = P - row * RES_PPR;
        const bool ok = row < RES_ROWS && col < RES_PPR - 1 && ch0 + col * 8 < NP && row < t_npix;
        const unsigned off = ok ? (unsigned)((t_m0 + row) * NP + ch0 + col * 8) * 2u : 0u;
        dma16(resb, off, lds0 + G::OFF_RES + wave * (RES_ROWS * RES_STR) + j * 1024, __builtin_amdgcn_ballot_w64(ok));
    };
    auto chunk = [&](auto cur_c, auto ksc_c, auto dmode_c, auto first_c, int c) {
        constexpr int CUR = decltype(cur_c)::value;             // c & 1
        constexpr bool FIRST = decltype(first_c)::value;        // chunk 0: its first k-step starts the accumulators
        constexpr int KSC = decltype(ksc_c)::value;             // k-steps of chunk c (2, or 1 for the last)
        constexpr int DMODE = decltype(dmode_c)::value;         // 0: no depthwise stage; 1: over 60 MFMA slots; 2: over 30
        constexpr int NXT = CUR ^ 1;
        const int cd = c + 1 < NCH ? c + 1 : 0;                 // chunk the depthwise stage builds
        {
            const int cc = c + 2 < NCH ? c + 2 : c + 2 - NCH;   // chunk the DMA fetches (c >= NCH-2: of the next tile)
#if !(defined(BQ_EXPERIMENTS) && (WIDE_ABLATE & 8))          // timing ablation (wrong results): 8 = no halo DMA in the loop
            if (c + 2 < NCH || has_next) {
                if (c + 2 >= NCH) halo_zero(CUR);
                halo_dma(cc, CUR);
            }
#endif
        }
        if constexpr (has_res) { if (c < NRES) res_dma(c); }
        if constexpr (BPRE) {                                   // the next chunk's fragments (past the end: the next tile's first chunk)
            const int kn = (c + 1 < NCH ? c + 1 : 0) * (KC / 32);
            load_bb(kn, bb[NXT][0]);
            load_bb(kn + 1, bb[NXT][1]);
        }
        // LDS addresses: the per-lane constant goes through opaque() FIRST, so that base + constant is formed here (or
        // folded into the instruction's immediate) instead of being hoisted out of the tile loop into one register per
        // buffer and use; the halo image's base lies beyond the 16-bit immediate: one add, opaque again
        const int a_cur = opaque(a_lane) + (OFF_A + CUR * A_BYTES);
        const int raw_addr = opaque(opaque(raw_lane) + (OFF_RAW + NXT * RAW_BYTES));
        AwAddr<NSTEP - OOB0, NSTEP> awn;
        awn.base = opaque(awb) + (OFF_A + NXT * A_BYTES);
        awn.dump = opaque(((opaque(tap_lane) - OFF_TAPS) >> 1) + (OFF_A + NXT * A_BYTES + MT * A_STR));   // cp * 4 + row 80
        awn.last = last_run;
        // taps of chunk cd; the last chunk has 32 channels: pairs 16..31 read a clamped (valid, unused) address
        const int tap_l = opaque(tap_lane);
        const int tap_addr = opaque(((LASTK == 1 && cd == NCH - 1 && tap_l >= OFF_TAPS + 128) ? tap_l - 128 : tap_l) + cd * (KC * 4));   // (pairs cp >= 16)
        DwState<T> st;
        uint4 a[MF];
        const int ks0 = c * (KC / 32);
        static_for<0, KSC * MF * RN>([&](auto qc) {
            constexpr int Q = decltype(qc)::value;
            constexpr int D = Q / (MF * RN), J = (Q % (MF * RN)) / MF, I = Q % MF;
#if !(defined(BQ_EXPERIMENTS) && (WIDE_ABLATE & 16))         // timing ablation (wrong results): 16 = no A-fragment reads
            if constexpr (Q == 0) {                             // A fragments of the chunk's first k-step
#pragma unroll
                for (int i = 0; i < MF; ++i) a[i] = *reinterpret_cast<const uint4*>(smem + a_cur + i * 16 * A_STR);
            }
#else
            if constexpr (Q == 0) {
#pragma unroll
                for (int i = 0; i < MF; ++i) asm volatile("" : "=v"(a[i].x), "=v"(a[i].y), "=v"(a[i].z), "=v"(a[i].w));
            }
#endif
#if !(defined(BQ_EXPERIMENTS) && (WIDE_ABLATE & 2))          // timing ablations (wrong results): 2 = no depthwise
            if constexpr (DMODE == 1) dw_ops<T, RELU, PW, KP, NSTEP, dw_before<NSLOTQ, NDW>(Q), dw_before<NSLOTQ, NDW>(Q + 1)>(st, smem, raw_addr, tap_addr, awn);
            if constexpr (DMODE == 2) dw_ops<T, RELU, PW, KP, NSTEP, dw_before<NSLOTQ, NDW>(2 * Q), dw_before<NSLOTQ, NDW>(2 * Q + 2)>(st, smem, raw_addr, tap_addr, awn);
#endif
#if defined(BQ_EXPERIMENTS) && (WIDE_ABLATE & 96)           // 32 = no MFMA on the last m-fragment, 64 = none on n-fragments 4, 5 of six:
            if constexpr (!(((WIDE_ABLATE & 32) && I == MF - 1 && MF == 5) || ((WIDE_ABLATE & 64) && RN == 6 && J >= 4)))   // what the padding costs
#endif
#if !(defined(BQ_EXPERIMENTS) && (WIDE_ABLATE & 4))          // 4 = no MFMA
            if constexpr (BPRE) mfma16_ab<T, FIRST && D == 0>(acc[I][J], bb[CUR][D][J], a[I]);
            else if constexpr (FIRST && D == 0) mfma16_first<T>(acc[I][J], bq[J], a[I]);
            else mfma16<T>(acc[I][J], bq[J], a[I]);
#endif
#if !(defined(BQ_EXPERIMENTS) && (WIDE_ABLATE & 1))          // 1 = weights stay in registers
            if constexpr (I == MF - 1 && !BPRE) {               // the fragment is dead: fetch it for the next k-step
                const int nx = ks0 + D + 1;
                bq[J] = load_b(nx < KST ? nx : 0, J);           // past the end: k-step 0, the next tile's first
            }
#endif
#if !(defined(BQ_EXPERIMENTS) && (WIDE_ABLATE & 16))
            if constexpr (J == RN - 1 && D + 1 < KSC)           // last use of this A fragment: fetch the next k-step's
                a[I] = *reinterpret_cast<const uint4*>(smem + a_cur + I * 16 * A_STR + (D + 1) * 64);
#endif
            __builtin_amdgcn_sched_barrier(0);                  // the source order of this loop IS the schedule
        });
        WSTAMP(6 + c);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(BPRE ? 0 : RN) : "memory");
        __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using K2 = std::integral_constant<int, 2>; using K1 = std::integral_constant<int, 1>;
    unsigned char* outb = reinterpret_cast<unsigned char*>(p.out);
    // Output (and residual) layout of a lane.  The host packs the pointwise weights so that the two 16-wide n-fragments
    // 2q, 2q+1 of a wave INTERLEAVE in groups of four channels (weights.py: pack_fragments16): fragment 2q's row m is
    // channel 32q + 8 (m / 4) + m % 4, fragment 2q+1's row m is channel 32q + 8 (m / 4) + 4 + m % 4.  A lane (pixel
    // l & 15, k-group l >> 4) then holds, for every pair, EIGHT consecutive channels 32q + 8 kg + (0..7) of its pixel:
    // one 16-byte store, and one 16-byte load of the residual, per pair -- no LDS staging, no crumbs.

    for (;;) {
        has_next = vb + nwg < ntiles;
        WSTAMP(18);
        WSTAMP_RT(30);
        if constexpr (NCH > 2) {
            chunk(I0{}, K2{}, I1{}, std::true_type{}, 0);
            chunk(I1{}, K2{}, I1{}, std::false_type{}, 1);
            for (int c = 2; c < NCH - 2; c += 2) {
                chunk(I0{}, K2{}, I1{}, std::false_type{}, c);
                chunk(I1{}, K2{}, I1{}, std::false_type{}, c + 1);
            }
        }
        const int m0 = t_m0, npix = t_npix;                      // of the tile whose accumulators are being finished
        if (has_next) tile_setup(vb + nwg, t_m0, t_npix);        // from here on the halo constants describe the next tile
        // (a plan of two chunks -- 128 input channels -- has nothing in front of these two: the first one starts the accumulators)
        chunk(I0{}, K2{}, I1{}, std::integral_constant<bool, NCH == 2>{}, NCH - 2);
        // (one instance whether or not a tile follows: two copies of the chunk under a branch make hipcc give the
        // accumulator tiles different registers on the two paths and move them between -- a workgroup's very last tile
        // builds an A chunk nobody reads)
        if constexpr (LASTK == 1) chunk(I1{}, K1{}, I2{}, std::false_type{}, NCH - 1);
        else chunk(I1{}, K2{}, I1{}, std::false_type{}, NCH - 1);

        // The last MFMAs have to have written their accumulators before anything reads them (see mfma16), and hipcc must not
        // move an accumulator read up in front of these wait states: every tile is an operand of one of the two statements
        // (an asm statement takes 30 operands; a tied one counts twice).
        static_assert((MF == 5 && (RN == 6 || RN == 4)) || (MF == 10 && RN == 2), "operand lists below");
        if constexpr (RN == 4) {
#define BQ_ACC_ROW(i) "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][RN - 1])
            asm volatile("s_nop 15\n\ts_nop 15" : BQ_ACC_ROW(0), BQ_ACC_ROW(1), BQ_ACC_ROW(2));
            asm volatile("" : BQ_ACC_ROW(3), BQ_ACC_ROW(4));
#undef BQ_ACC_ROW
        } else if constexpr (RN == 6) {
#define BQ_ACC_ROW(i) "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5])
            asm volatile("s_nop 15\n\ts_nop 15" : BQ_ACC_ROW(0), BQ_ACC_ROW(1), "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]));
            asm volatile("" : "+a"(acc[2][3]), "+a"(acc[2][4]), "+a"(acc[2][5]), BQ_ACC_ROW(3), BQ_ACC_ROW(4));
#undef BQ_ACC_ROW
        } else {
#define BQ_ACC_ROW(i) "+a"(acc[i][0]), "+a"(acc[i][1])
            asm volatile("s_nop 15\n\ts_nop 15" : BQ_ACC_ROW(0), BQ_ACC_ROW(1), BQ_ACC_ROW(2), BQ_ACC_ROW(3), BQ_ACC_ROW(4), BQ_ACC_ROW(5), BQ_ACC_ROW(6));
            asm volatile("" : BQ_ACC_ROW(7), BQ_ACC_ROW(8), BQ_ACC_ROW(MF - 1));
#undef BQ_ACC_ROW
        }
        WSTAMP(20);
        // ---- epilogue: folded BN (+ residual) (+ ReLU), 16-bit, straight from the accumulators to HBM -----------------
        {
            const unsigned lo2 = p.relu ? 0u : 0x80008000u;     // packed int16 max with 0 = ReLU, with -32768 = no-op
            const int er16 = (int)(voff >> 4) & 15, ekg = (int)(voff >> 8);   // lane & 15, lane >> 4 (not kept across the K loop)
            const unsigned lane_ch = (unsigned)(ch0 + ekg * 8) * 2u;    // byte offset of the lane's first channel within a pixel row
            // The folded-BN table address goes through opaque(): the table never changes, and hipcc would otherwise hoist
            // its reads out of the tile loop and keep 48 registers across the K loop.
            const int sb_lane = opaque(OFF_SB + (ch0 + ekg * 8) * 4);
            // vmcnt retires in order and a store is only retired when it is acknowledged (~1.8 k cycles here), so a load
            // issued BEHIND a store cannot be consumed before that: every load of the epilogue is issued before its first
            // store.  Residual rows 0..31 are already in LDS (copied under the K loop), rows 32..79 (row fragments 2-4)
            // are fetched now, 9 x 16 bytes per lane, and land while fragments 0 and 1 are finished.
            uint4 rg[RES ? MF - 2 : 1][RES ? RN / 2 : 1];
            if constexpr (has_res) {
                const unsigned q2 = ch0 + 2 * 32 < NP ? 128u : 0u;   // wave 7's last pair is channel padding: any valid address
#pragma unroll
                for (int i = 2; i < MF; ++i) {
                    const int row = i * 16 + er16;
                    const unsigned off = (unsigned)(m0 + (row < npix ? row : 0)) * (unsigned)(NP * 2) + lane_ch;
                    rg[i - 2][0] = *reinterpret_cast<const uint4*>(resb + off);
                    rg[i - 2][1] = *reinterpret_cast<const uint4*>(resb + off + 64);
                    rg[i - 2][2] = *reinterpret_cast<const uint4*>(resb + off + q2);
                }
            }
            const int res_lane = opaque(G::OFF_RES + wave * (RES_ROWS * RES_STR) + er16 * RES_STR + ekg * 16);
            // pair-major: the folded-BN constants of one pair are 16 registers (the next tile's weight fragments and every
            // per-lane constant of the loop stay live across this epilogue)
#pragma unroll
            for (int i = 0; i < MF; ++i) {
#pragma unroll
                for (int q = 0; q < RN / 2; ++q) {
                    if (ch0 + q * 32 >= NP) continue;           // wave 7's last pair lies in the channel padding (wave-uniform)
                    // row-major: the three 64-byte pieces of a pixel row go out back to back, so the halves of a 128-byte
                    // line reach L2 together; the folded-BN constants are re-read from LDS for every piece (16 registers)
                    const int sbq = opaque(sb_lane + q * 128);
                    const float4 s0 = *reinterpret_cast<const float4*>(smem + sbq);
                    const float4 s1 = *reinterpret_cast<const float4*>(smem + sbq + 16);
                    const float4 b0 = *reinterpret_cast<const float4*>(smem + sbq + G::SB_HALF);
                    const float4 b1 = *reinterpret_cast<const float4*>(smem + sbq + G::SB_HALF + 16);
                    const int row = i * 16 + er16;
                    // the accumulators are copied out HERE, tile by tile (volatile: hipcc otherwise copies the 40 registers of a
                    // pair, or all 120, out up front and spills around them)
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x[e]) : "a"(acc[i][2 * q][e]));
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x[4 + e]) : "a"(acc[i][2 * q + 1][e]));
                    }
                    float v0 = fmaf(x[0], s0.x, b0.x), v1 = fmaf(x[1], s0.y, b0.y);
                    float v2 = fmaf(x[2], s0.z, b0.z), v3 = fmaf(x[3], s0.w, b0.w);
                    float v4 = fmaf(x[4], s1.x, b1.x), v5 = fmaf(x[5], s1.y, b1.y);
                    float v6 = fmaf(x[6], s1.z, b1.z), v7 = fmaf(x[7], s1.w, b1.w);
                    if constexpr (has_res) {
                        uint4 u;
                        if (i < 2) u = *reinterpret_cast<const uint4*>(smem + res_lane + i * 16 * RES_STR + q * 64);
                        else u = rg[i < 2 ? 0 : i - 2][q];
                        v0 = H16<T>::add_lo(v0, u.x); v1 = H16<T>::add_hi(v1, u.x);
                        v2 = H16<T>::add_lo(v2, u.y); v3 = H16<T>::add_hi(v3, u.y);
                        v4 = H16<T>::add_lo(v4, u.z); v5 = H16<T>::add_hi(v5, u.z);
                        v6 = H16<T>::add_lo(v6, u.w); v7 = H16<T>::add_hi(v7, u.w);
                    }
                    uint4 o;
                    o.x = H16<T>::pack2(v0, v1); o.y = H16<T>::pack2(v2, v3);
                    o.z = H16<T>::pack2(v4, v5); o.w = H16<T>::pack2(v6, v7);
                    asm("v_pk_max_i16 %0, %1, %2" : "=v"(o.x) : "v"(o.x), "v"(lo2));
                    asm("v_pk_max_i16 %0, %1, %2" : "=v"(o.y) : "v"(o.y), "v"(lo2));
                    asm("v_pk_max_i16 %0, %1, %2" : "=v"(o.z) : "v"(o.z), "v"(lo2));
                    asm("v_pk_max_i16 %0, %1, %2" : "=v"(o.w) : "v"(o.w), "v"(lo2));
                    if (row < npix)
                        *reinterpret_cast<uint4*>(outb + (size_t)((unsigned)(m0 + row) * (unsigned)(LDO * 2) + lane_ch + q * 64)) = o;
                    __builtin_amdgcn_sched_barrier(0);          // (or hipcc copies all 120 accumulators out up front)
                }
            }
        }
        WSTAMP(22);
        WSTAMP_RT(31);
#ifdef BQ_EXPERIMENTS
        ++stamp_it;
#endif
        if (!has_next) break;
        vb += nwg;
    }
}

}  // namespace

using G19 = Geo<19, 4>;     // blocks 5-12 and block13_sepconv1: 5 tiles of 4 (the last: 3) rows per image
using G37 = Geo<37, 2>;     // block4_sepconv2: 19 tiles of 2 (the last: 1) rows per image
using G74 = Geo<74, 2, 256, 256>;   // block3_sepconv1 / 2 (128 / 256 -> 256): 37 tiles of 2 rows = 148 pixels, 160 MFMA rows

// The kernel forms byte offsets into the activation tensors in 32 bits: n * H * W * 736 * 2 must stay below 2^32
// (n < 8 085 images at 19x19, n < 2 132 at 37x37 -- a larger batch falls back to the pipelined kernel).
// Instances: K = 736 (728 -> 728) on 19x19 and 37x37 maps, K = 256 (block4_sepconv1: 256 -> 728) on 37x37 maps, and
// 256 -> 256 on 74x74 maps (block3_sepconv2: no ReLU in front, no residual input).
bool wide_supported(int dtype, int prod, int nfp, int H, int W, int K, int Nstore, int ldi, int ldo, long long M, bool residual) {
    const bool wide = Nstore == 736 && nfp * 2 == NPlan<736>::NFT &&
                      ((K == 736 && (W == G19::IW || (W == G37::IW && !residual))) || (K == 256 && W == G37::IW && !residual));
    // block 3: 256 -> 256 behind the ReLU of the layer before, 128 -> 256 with its own ReLU in front
    const bool b3 = Nstore == 256 && nfp * 2 == NPlan<256>::NFT && W == G74::IW && !residual &&
                    ((K == 256 && prod == PROD_DW) || (K == 128 && prod == PROD_DW_RELU));
    // block13_sepconv2 (728 -> 1 024): two launches of 512 columns each (160 accumulator registers per wave do not exist)
    const bool b13 = Nstore == 1024 && nfp * 2 == 2 * NPlan<512>::NFT && K == 736 && W == G19::IW && !residual && prod == PROD_DW;
    return dtype != 0 && (prod == PROD_DW || prod == PROD_DW_RELU) && H == W && (wide || b3 || b13) &&
           ldi == K && ldo == Nstore && M > 0 && M % (H * W) == 0 && M * Nstore * 2 < (1ll << 32);
}

// wp16: the layer's pointwise weights in 16x16x32 fragment order, n-fragment pairs interleaved (blob entry "<layer>/wp16")
int launch_sepconv_wide(int dtype, int prod, const GemmParams& g, const void* wp16, int num_cus, hipStream_t s) {
    const bool big = g.W == G37::IW;
    const int hw = g.H * g.W;
    if (!wide_supported(dtype, prod, g.NFp, g.H, g.W, g.K, g.Nstore, g.ldi, g.ldo, g.M, g.residual != nullptr) || g.k_off != 0 ||
        !g.scale || !g.bias || !wp16)
        return (int)hipErrorInvalidValue;
    WideParams p;
    p.in = reinterpret_cast<const h16_t*>(g.in);
    p.wp = reinterpret_cast<const uint4*>(wp16);
    p.dw = g.dw; p.scale = g.scale; p.bias = g.bias;
    p.residual = reinterpret_cast<const h16_t*>(g.residual);
    p.out = reinterpret_cast<h16_t*>(g.out);
    p.n = g.M / hw;
    p.relu = g.relu;
    const bool relu_in = prod == PROD_DW_RELU;
#define BQ_WIDE_SET(T, RES) sepconv_wide_kernel<T, false, RES, G19, 736>, sepconv_wide_kernel<T, true, RES, G19, 736>, \
                            sepconv_wide_kernel<T, false, RES, G37, 736>, sepconv_wide_kernel<T, true, RES, G37, 736>
    void (*const kerns[26])(const WideParams) = {BQ_WIDE_SET(bf16_t, false), BQ_WIDE_SET(f16_t, false),
                                                 BQ_WIDE_SET(bf16_t, true), BQ_WIDE_SET(f16_t, true),
                                                 sepconv_wide_kernel<bf16_t, false, false, G37, 256>,
                                                 sepconv_wide_kernel<bf16_t, true, false, G37, 256>,
                                                 sepconv_wide_kernel<f16_t, false, false, G37, 256>,
                                                 sepconv_wide_kernel<f16_t, true, false, G37, 256>,
                                                 sepconv_wide_kernel<bf16_t, false, false, G74, 256, 256>,
                                                 sepconv_wide_kernel<f16_t, false, false, G74, 256, 256>,
                                                 sepconv_wide_kernel<bf16_t, true, false, G74, 128, 256>,
                                                 sepconv_wide_kernel<f16_t, true, false, G74, 128, 256>,
                                                 sepconv_wide_kernel<bf16_t, false, false, G19, 736, 512, 1024, 64>,
                                                 sepconv_wide_kernel<f16_t, false, false, G19, 736, 512, 1024, 64>};
#undef BQ_WIDE_SET
    const bool b3 = g.W == G74::IW, b13 = g.Nstore == 1024;
    const int ki = b13 ? 24 + (dtype == 2 ? 1 : 0)
                 : b3 ? 20 + (g.K == 128 ? 2 : 0) + (dtype == 2 ? 1 : 0)
                 : g.K == 256 ? 16 + (dtype == 2 ? 2 : 0) + (relu_in ? 1 : 0)
                              : (p.residual ? 8 : 0) + (dtype == 2 ? 4 : 0) + (big ? 2 : 0) + (relu_in ? 1 : 0);
    auto kern = kerns[ki];
    const int tpi = b3 ? G74::TPI : big ? G37::TPI : G19::TPI;
    const bool res = p.residual != nullptr;
    const int lds = b3 ? G74::lds_bytes(false) : big ? G37::lds_bytes(res) : G19::lds_bytes(res);   // (b13: G19 without a residual)
    static BqLdsAttr attr[26];
    if (const int e = attr[ki].ensure(reinterpret_cast<const void*>(kern), lds)) return e;
    // one persistent workgroup per CU (a multiple of 8: every workgroup stays inside its XCD's run of tiles)
    const int ntiles = p.n * tpi;
    int nwg = (num_cus > 0 ? num_cus : 256) & ~7;
    if (nwg < 8) nwg = 8;
    if (nwg > ntiles) nwg = ntiles;         // fewer tiles than CUs: one tile each (any count: nobody takes a second tile)
    p.nwg = nwg;
#ifdef BQ_EXPERIMENTS
    // BQ_STAMPS_WIDE=<file>: in-kernel s_memtime stamps of the first launch with (BQ_STAMPS_NORES: without) a residual
    static const char* stamp_file = bq_exp_env("BQ_STAMPS_WIDE");
    static const bool want_res = bq_exp_env("BQ_STAMPS_NORES") == nullptr;
    static int state = 0;
    static unsigned long long* d_stamps = nullptr;
    p.stamps = nullptr;
    p.stamp_b0 = bq_exp_env("BQ_STAMPS_B0") ? (unsigned)atoi(bq_exp_env("BQ_STAMPS_B0")) : 0u;
    if (stamp_file && state == 0 && (p.residual != nullptr) == want_res && p.n >= 64 &&
        hipMalloc(&d_stamps, 64 * WN * 32 * STAMP_TILES * 8) == hipSuccess) {
        (void)hipMemsetAsync(d_stamps, 0, 64 * WN * 32 * STAMP_TILES * 8, s);
        p.stamps = d_stamps;
        state = 1;
    }
#endif
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * WN), lds, s, p);
    if (b13) {                              // the second 512 columns: weights 32 n-fragments on, tables and output 512 channels on
        p.wp = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(wp16) + (size_t)NPlan<512>::NFT * 1024);
        p.scale += 512; p.bias += 512; p.out += 512;
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * WN), lds, s, p);
    }
#ifdef BQ_EXPERIMENTS
    if (state == 1) {
        std::vector<unsigned long long> h(64 * WN * 32 * STAMP_TILES);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost);
        if (FILE* f = fopen(stamp_file, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
        state = 2;
    }
#endif
    return (int)hipGetLastError();
}
