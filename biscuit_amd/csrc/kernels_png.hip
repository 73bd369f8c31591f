// PNG scanline un-filtering on the GPU: the second half of the tile reader's PNG decode.
//
// A PNG decoder is zlib inflate -- a serial Huffman stream, which stays on the host cores (csrc/inflate_fast.h) -- followed
// by the reversal of the per-row prediction filter (None / Sub / Up / Average / Paeth), a byte recurrence on the left, upper
// and upper-left neighbours that costs the host a third of a photo-like tile's decode time.  The recurrence is serial along
// a row and from row to row, but row r at column x only needs row r-1 up to column x: 64 rows run as a skewed wavefront in
// one wave (lane l works on column s - l at step s; the upper pixel is lane l-1's result of the step before, handed over by
// one cross-lane move, the upper-left one is what that move delivered a step earlier).  One workgroup of one wave per tile,
// bands of 64 rows staged through LDS (coalesced reads of the filtered bytes, in-place results, coalesced writes of the RGB
// tile); 256 tiles = one wave on every CU.
//
// Input: what libbiscuit_io's bqio_decode_rows leaves in (pinned) host memory -- per tile px rows of 1 filter-type byte +
// 3 px filtered bytes, exactly the inflated IDAT stream of an 8-bit RGB, non-interlaced PNG (tiles of any other kind are
// un-filtered by the reader and arrive as filter type 0).  Output: uint8 NHWC [n][px][px][3], what bq_stage takes.
#include "bq_common.h"

namespace {

constexpr int BAND = 64;
constexpr int NTHR = 256;           // wave 0 runs the wavefront; all four waves move the band in and out of LDS
constexpr int PAD = 4;              // a row in LDS: 3 unused bytes, the filter-type byte, then the pixel bytes dword-aligned

__device__ __forceinline__ int row_pitch(int px) {                 // a whole, ODD number of dwords: the 64 rows a wave
    const int dw = 3 * ((px + 3) / 4) + PAD / 4 + 1;               // touches in one step lie in different banks
    return 4 * (dw | 1);
}

// the previous lane's value: one DPP move, not a trip through the LDS crossbar
__device__ __forceinline__ unsigned from_lane_above(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// Three channels at once, one byte each in bits 0..23 of a dword:
__device__ __forceinline__ unsigned add3(unsigned f, unsigned p) {      // byte-wise (f + p) mod 256
    return ((f & 0x7f7f7fu) + (p & 0x7f7f7fu)) ^ ((f ^ p) & 0x808080u);
}
__device__ __forceinline__ unsigned avg3(unsigned a, unsigned b) {      // byte-wise floor((a + b) / 2)
    return (a & b) + (((a ^ b) & 0xfefefeu) >> 1);
}
__device__ __forceinline__ unsigned paeth3(unsigned a, unsigned b, unsigned c) {
    unsigned r = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const int av = (a >> (8 * ch)) & 255, bv = (b >> (8 * ch)) & 255, cv = (c >> (8 * ch)) & 255;
        const int pa = abs(bv - cv), pb = abs(av - cv), pc = abs(av + bv - 2 * cv);
        r |= (unsigned)((pa <= pb && pa <= pc) ? av : (pb <= pc ? bv : cv)) << (8 * ch);     // ties: left, then up, then upper-left
    }
    return r;
}

// One workgroup = one tile; its first wave runs the wavefront.  Lane l works on row r0 + l of the band, one BLOCK of four pixels (three dwords) per
// step, block s - l in step s: every lane is at the same place inside its block, so nothing in a step depends on the lane
// but its data, the upper block is the three dwords lane l-1 produced the step before (three DPP moves), and LDS is
// touched with aligned dwords only.
__global__ void __launch_bounds__(NTHR) png_unfilter_kernel(const unsigned char* __restrict__ rows, unsigned char* __restrict__ out,
                                                          int px, size_t in_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int W3 = 3 * px, RS = W3 + 1;
    const int RSP = row_pitch(px);
    unsigned char* prev = lds + BAND * RSP;                             // the row above the band (its last result row), dword-aligned
    const int tid = threadIdx.x, lane = tid & 63;
    const bool wave0 = tid < 64;
    const unsigned char* src = rows + (size_t)blockIdx.x * in_stride;      // (px * RS when the tiles' rows are packed)
    unsigned char* dst = out + (size_t)blockIdx.x * px * W3;
    const int nblk = (px + 3) / 4;                                      // blocks of 4 pixels = 3 dwords (the pitch covers the last, partial one)

    for (int r0 = 0; r0 < px; r0 += BAND) {
        const int nb = px - r0 < BAND ? px - r0 : BAND;
        // ---- the band's filtered rows -> LDS.  One wave has no other wave to hide a load's latency behind, so the loads go
        // out in batches (4 rows x as many 64-byte pieces as a row is long) before the first byte is written
        {
            const unsigned char* s0 = src + (size_t)r0 * RS;
            constexpr int RB = 4, CB = 4;                               // rows per batch, 256-byte pieces per row (<= 1024 bytes)
            for (int row = 0; row < nb; row += RB) {
                unsigned char v[RB][CB];
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int c = 0; c < CB; ++c) {
                        const int col = tid + NTHR * c;
                        v[r][c] = (row + r < nb && col < RS) ? s0[(row + r) * RS + col] : (unsigned char)0;
                    }
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int c = 0; c < CB; ++c) {
                        const int col = tid + NTHR * c;
                        if (row + r < nb && col < RS) lds[(row + r) * RSP + (PAD - 1) + col] = v[r][c];
                    }
            }
        }
        __syncthreads();
        const bool mine = wave0 && lane < nb;
        const int ft = mine ? lds[lane * RSP + PAD - 1] : 0;
        unsigned* my = reinterpret_cast<unsigned*>(lds + lane * RSP + PAD);
        const unsigned* pv = reinterpret_cast<const unsigned*>(prev);
        const bool top = r0 == 0;                                       // no row above the image's first
        const bool any_paeth = __any(ft == 4);                          // wave-uniform: the per-channel Paeth code only where a row asks for it
        unsigned left = 0, upleft = 0;                                  // last pixel of the lane's / the upper row's previous block
        unsigned o0 = 0, o1 = 0, o2 = 0;                                // the lane's results of the step before: the next lane's upper block
        if (wave0)
        for (int s = 0; s < nblk + nb - 1; ++s) {
            const int blk = s - lane;
            const bool live = mine && blk >= 0 && blk < nblk;
            unsigned u0 = from_lane_above(o0), u1 = from_lane_above(o1), u2 = from_lane_above(o2);
            if (lane == 0) {
                u0 = u1 = u2 = 0;
                if (!top && live) { u0 = pv[3 * blk]; u1 = pv[3 * blk + 1]; u2 = pv[3 * blk + 2]; }
            }
            if (live) {
                const unsigned i0 = my[3 * blk], i1 = my[3 * blk + 1], i2 = my[3 * blk + 2];
                const unsigned f[4] = {i0 & 0xffffffu, (i0 >> 24) | ((i1 & 0xffffu) << 8), (i1 >> 16) | ((i2 & 0xffu) << 16), i2 >> 8};
                const unsigned u[4] = {u0 & 0xffffffu, (u0 >> 24) | ((u1 & 0xffffu) << 8), (u1 >> 16) | ((u2 & 0xffu) << 16), u2 >> 8};
                if (blk == 0) { left = 0; upleft = 0; }
                unsigned r[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned p = ft == 1 ? left : 0u;
                    p = ft == 2 ? u[q] : p;
                    p = ft == 3 ? avg3(left, u[q]) : p;
                    if (any_paeth) p = ft == 4 ? paeth3(left, u[q], upleft) : p;
                    r[q] = add3(f[q], p);
                    left = r[q];
                    upleft = u[q];
                }
                o0 = r[0] | (r[1] << 24);
                o1 = (r[1] >> 8) | (r[2] << 16);
                o2 = (r[2] >> 16) | (r[3] << 8);
                my[3 * blk] = o0; my[3 * blk + 1] = o1; my[3 * blk + 2] = o2;
            }
        }
        __syncthreads();
        // ---- results out (row-major RGB, no filter bytes) and the band's last row kept for the next band
        {
            unsigned char* d0 = dst + (size_t)r0 * W3;
            constexpr int RB = 4, CB = 4;
            for (int row = 0; row < nb; row += RB) {
                unsigned char v[RB][CB];
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int c = 0; c < CB; ++c) {
                        const int col = tid + NTHR * c;
                        v[r][c] = (row + r < nb && col < W3) ? lds[(row + r) * RSP + PAD + col] : (unsigned char)0;
                    }
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int c = 0; c < CB; ++c) {
                        const int col = tid + NTHR * c;
                        if (row + r < nb && col < W3) d0[(row + r) * W3 + col] = v[r][c];
                    }
            }
            const unsigned* lastrow = reinterpret_cast<const unsigned*>(lds + (nb - 1) * RSP + PAD);
            unsigned* pw = reinterpret_cast<unsigned*>(prev);
            for (int i = tid; i < 3 * nblk; i += NTHR) pw[i] = lastrow[i];
        }
        __syncthreads();
    }
}

}  // namespace

// in_stride: bytes between the row data of consecutive tiles (0: packed, px * (1 + 3 px); the device inflate pads to a dword)
int launch_png_unfilter(const unsigned char* rows, int n, int px, unsigned char* out, hipStream_t s, size_t in_stride) {
    if (n <= 0) return 0;
    if (in_stride == 0) in_stride = (size_t)px * (3 * px + 1);
    if (in_stride < (size_t)px * (3 * px + 1)) return (int)hipErrorInvalidValue;
    if (3 * px + 1 > 1024) return (int)hipErrorInvalidValue;         // (the band loader's batch: rows of up to 1 KiB, px <= 341)
    const int dw = 3 * ((px + 3) / 4) + PAD / 4 + 1;
    const int RSP = 4 * (dw | 1);
    const int lds = BAND * RSP + 12 * ((px + 3) / 4) + 16;
    if (lds > 64 * 1024) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(png_unfilter_kernel, dim3(n), dim3(NTHR), lds, s, rows, out, px, in_stride);
    return (int)hipGetLastError();
}
