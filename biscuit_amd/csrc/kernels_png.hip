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
constexpr int PAD = 4;              // a row in LDS: 3 unused bytes, the filter-type byte, then the pixel bytes dword-aligned

__device__ __forceinline__ int row_pitch(int px) {                 // a whole, ODD number of dwords: the 64 rows a wave
    const int dw = (3 * px + PAD + 3) / 4 + 1;                     // touches in one step lie in different banks
    return 4 * (dw | 1);
}

// the previous lane's value (lane 0: the `first` argument): one DPP move, not a trip through the LDS crossbar
__device__ __forceinline__ unsigned from_lane_above(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// one channel: filtered byte + predictor, by filter type -- every predictor is computed, the type selects (no divergence)
__device__ __forceinline__ unsigned unfilter1(int ft, int f, int a, int b, int c) {
    const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
    const int paeth = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);          // ties: left, then up, then upper-left
    int p = ft == 1 ? a : 0;
    p = ft == 2 ? b : p;
    p = ft == 3 ? (a + b) >> 1 : p;
    p = ft == 4 ? paeth : p;
    return (unsigned)((f + p) & 255);
}

__global__ void __launch_bounds__(64) png_unfilter_kernel(const unsigned char* __restrict__ rows, unsigned char* __restrict__ out,
                                                          int px) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int W3 = 3 * px, RS = W3 + 1;
    const int RSP = row_pitch(px);
    unsigned char* prev = lds + BAND * RSP;                             // the row above the band (its last result row), dword-aligned
    const int lane = threadIdx.x;
    const unsigned char* src = rows + (size_t)blockIdx.x * px * RS;
    unsigned char* dst = out + (size_t)blockIdx.x * px * W3;
    const int nblk = (px + 3) / 4;                                      // blocks of 4 pixels = 3 dwords

    for (int r0 = 0; r0 < px; r0 += BAND) {
        const int nb = px - r0 < BAND ? px - r0 : BAND;
        // ---- the band's filtered rows -> LDS.  One wave has no other wave to hide a load's latency behind, so the loads go out
        // in batches of 32 (8 rows x 4 pieces of 64 bytes... as many as the row is long) before the first byte is written
        {
            const unsigned char* s0 = src + (size_t)r0 * RS;
            constexpr int RB = 4, CB = 16;                              // rows per batch, 64-byte pieces per row (<= 1024 bytes)
            for (int row = 0; row < nb; row += RB) {
                unsigned char v[RB][CB];
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int c = 0; c < CB; ++c) {
                        const int col = lane + 64 * c;
                        v[r][c] = (row + r < nb && col < RS) ? s0[(row + r) * RS + col] : (unsigned char)0;
                    }
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int c = 0; c < CB; ++c) {
                        const int col = lane + 64 * c;
                        if (row + r < nb && col < RS) lds[(row + r) * RSP + (PAD - 1) + col] = v[r][c];
                    }
            }
        }
        __syncthreads();
        // ---- the wavefront: lane l = row r0 + l, column x = s - l at step s; a lane keeps 4 pixels (3 dwords) of its row's
        // input, of its results and (lane 0) of the row above in registers and goes to LDS once per block
        const bool mine = lane < nb;
        const int ft = mine ? lds[lane * RSP + PAD - 1] : 0;
        unsigned* my = reinterpret_cast<unsigned*>(lds + lane * RSP + PAD);
        const unsigned* pv = reinterpret_cast<const unsigned*>(prev);
        const bool top = r0 == 0;                                       // no row above the image's first
        unsigned left = 0, last = 0, upleft = 0;                        // packed r | g << 8 | b << 16
        unsigned in0 = 0, in1 = 0, in2 = 0, o0 = 0, o1 = 0, o2 = 0, p0 = 0, p1 = 0, p2 = 0;
        for (int s = 0; s < px + nb - 1; ++s) {
            const int x = s - lane;
            unsigned up = from_lane_above(last);                        // lane l-1's result at column x (its previous step)
            const bool live = mine && x >= 0 && x < px;
            const int q = x & 3, blk = x >> 2;
            if (live && q == 0) {                                       // a new block of 4 pixels
                in0 = my[3 * blk]; in1 = my[3 * blk + 1]; in2 = my[3 * blk + 2];
                if (lane == 0 && !top) { p0 = pv[3 * blk]; p1 = pv[3 * blk + 1]; p2 = pv[3 * blk + 2]; }
            }
            // pixel q of the block: bytes 3q .. 3q+2 of the 12
            const unsigned long long lo = ((unsigned long long)in1 << 32) | in0, hi = ((unsigned long long)in2 << 32) | in1;
            const unsigned f = q == 0 ? in0 : q == 1 ? (unsigned)(lo >> 24) : q == 2 ? (unsigned)(hi >> 16) : (in2 >> 8);
            if (lane == 0) {
                const unsigned long long plo = ((unsigned long long)p1 << 32) | p0, phi = ((unsigned long long)p2 << 32) | p1;
                up = top ? 0u : (q == 0 ? p0 : q == 1 ? (unsigned)(plo >> 24) : q == 2 ? (unsigned)(phi >> 16) : (p2 >> 8));
            }
            if (live) {
                const unsigned c = x == 0 ? 0u : upleft;
                const unsigned res = unfilter1(ft, f & 255, left & 255, up & 255, c & 255) |
                                     (unfilter1(ft, (f >> 8) & 255, (left >> 8) & 255, (up >> 8) & 255, (c >> 8) & 255) << 8) |
                                     (unfilter1(ft, (f >> 16) & 255, (left >> 16) & 255, (up >> 16) & 255, (c >> 16) & 255) << 16);
                left = res;
                last = res;
                upleft = up & 0xffffffu;                                // this column's upper pixel is the next column's upper-left
                // the 3 result bytes into the block's 12
                if (q == 0) o0 = res;
                else if (q == 1) { o0 |= res << 24; o1 = res >> 8; }
                else if (q == 2) { o1 |= res << 16; o2 = res >> 16; }
                else o2 |= res << 8;
                if (q == 3 || x == px - 1) { my[3 * blk] = o0; my[3 * blk + 1] = o1; my[3 * blk + 2] = o2; }
            }
        }
        (void)nblk;
        __syncthreads();
        // ---- results out (row-major RGB, no filter bytes) and the band's last row kept for the next band
        {
            unsigned char* d0 = dst + (size_t)r0 * W3;
            for (int row = 0; row < nb; ++row)
                for (int col = lane; col < W3; col += 64) d0[row * W3 + col] = lds[row * RSP + PAD + col];
            const unsigned* lastrow = reinterpret_cast<const unsigned*>(lds + (nb - 1) * RSP + PAD);
            unsigned* pw = reinterpret_cast<unsigned*>(prev);
            for (int i = lane; i < (W3 + 3) / 4; i += 64) pw[i] = lastrow[i];
        }
        __syncthreads();
    }
}

}  // namespace

int launch_png_unfilter(const unsigned char* rows, int n, int px, unsigned char* out, hipStream_t s) {
    if (n <= 0) return 0;
    if (3 * px + 1 > 1024) return (int)hipErrorInvalidValue;         // (the band loader's batch: rows of up to 1 KiB, px <= 341)
    const int dw = (3 * px + PAD + 3) / 4 + 1;
    const int RSP = 4 * (dw | 1);
    const int lds = BAND * RSP + (3 * px + 3) / 4 * 4 + 16;
    if (lds > 64 * 1024) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(png_unfilter_kernel, dim3(n), dim3(64), lds, s, rows, out, px);
    return (int)hipGetLastError();
}
