// Tile shapes for the 728 -> 728 @19x19 kernel, costed before anything is built (round-4 review, item 4): the matrix loop of
// tools/ubench/wide_ceiling.hip -- v_mfma_f32_16x16x32_f16 with the accumulators tied in place, A fragments from LDS (two
// k-steps per 64-channel chunk, one workgroup barrier per chunk), weight fragments streamed from L2 one k-step ahead, 23
// k-steps per tile, persistent workgroups, one per CU -- as a template over
//   WM x WN   the arrangement of the workgroup's eight waves (rows x columns of wave tiles)
//   MI x NJ   16-row x 16-column fragments per wave (accumulators: 4 MI NJ registers)
//   VPM       vector-ALU instructions issued between consecutive MFMAs: a stand-in for the depthwise stage, which the real kernel
//             interleaves at ~4 per MFMA; a shape whose two column halves are computed by two workgroups builds every A row twice
//             (8 per MFMA)
// so that the candidates can be compared at the clock each holds:
//   80 x 768   (1 x 8 waves of 5 x 6)   the shipped shape: 1 280 tiles of 4 map rows (76 of 80 rows used), 5 per workgroup
//   96 x 768   (1 x 8 waves of 6 x 6)   5 map rows per tile (95 of 96 rows; an image = 4 tiles of 5 + 5 + 5 + 4 rows): 1 024 tiles,
//                                       4 per workgroup, weights streamed 4 instead of 5 times per image, 144 accumulators
//   192 x 384  (2 x 4 waves of 6 x 6)   the review's shape: 10 + 9 map rows x a column half, 1 024 tiles, 4 per workgroup; a weight
//                                       fragment is fetched by two waves (the second from the vector L1), the A build is doubled
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o wide_shapes wide_shapes.hip && ./wide_shapes
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KST = 23, CH = 64;
constexpr int AST = CH * 2 + 32;                    // 10 slots of 16 B: conflict-free fragment reads

__device__ __forceinline__ void mfma(f32x4v& acc, const u32x4& b, const u32x4& a) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(b), "v"(a));
}

template <int WM, int WN, int MI, int NJ, int VPM>
__global__ void __launch_bounds__(512) shape_kernel(const uint4* __restrict__ wp, float* __restrict__ out, int tiles_per_wg, int nft,
                                                    unsigned long long* __restrict__ stamps) {
    static_assert(WM * WN == 8, "eight waves");
    constexpr int ROWS = WM * MI * 16, A_BYTES = ROWS * AST;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // two A chunks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * A_BYTES / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003800u + (i * 2654435761u >> 20);
    __syncthreads();
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, kg = lane >> 4;
    const uint4* wq = wp + (size_t)(wn * NJ) * 64 + lane;
    const int row0 = wm * MI * 16;
    float v[8];                                                              // the vector-ALU stand-in: eight independent fma chains
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + 0.001f * (float)(lane + i);
    const float vm = 0.9999f, va = 0.0001f;
    unsigned long long t0 = 0, r0 = 0;
    if (stamps && tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int t = 0; t < tiles_per_wg; ++t) {
        f32x4v acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        u32x4 bq[NJ], bn[NJ], a[MI];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bq[j] = __builtin_bit_cast(u32x4, wq[(size_t)j * 64]);
#pragma unroll 1
        for (int ks = 0; ks < KST; ++ks) {
            if ((ks & 1) == 0) __syncthreads();                                  // one barrier per 64-channel chunk
            const unsigned char* ab = smem + ((ks >> 1) & 1) * A_BYTES + (ks & 1) * 64;
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const u32x4*>(ab + (row0 + i * 16 + r16) * AST + kg * 16);
            const int kn = ks + 1 < KST ? ks + 1 : 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) bn[j] = __builtin_bit_cast(u32x4, wq[((size_t)kn * nft + j) * 64]);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    mfma(acc[i][j], bq[j], a[i]);
#pragma unroll
                    for (int q = 0; q < VPM; ++q)
                        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i * NJ + j + q) & 7]) : "v"(vm), "v"(va));
                }
#pragma unroll
            for (int j = 0; j < NJ; ++j) bq[j] = bn[j];
        }
        asm volatile("s_nop 7\n s_nop 7\n s_nop 3" ::: "memory");
        f32x4v s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) s += acc[i][j];
        float vs = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) vs += v[i];
        out[((size_t)(blockIdx.x * tiles_per_wg + t) * 512 + tid)] = s[0] + s[1] + s[2] + s[3] + vs;
    }
    if (stamps && tid == 0) {
        stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
        stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// tiles: workgroup tiles per launch (all 256 images x 361 pixels x 768 padded columns); used_rows: map pixels per tile on average
template <int WM, int WN, int MI, int NJ, int VPM>
int run(const char* name, const uint4* wp, float* out, unsigned long long* stamps, int tiles_per_wg, int nft, int reps) {
    constexpr int ROWS = WM * MI * 16, COLS = WN * NJ * 16, A_BYTES = ROWS * AST;
    const int grid = 256;
    auto kern = shape_kernel<WM, WN, MI, NJ, VPM>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * A_BYTES));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 2 * A_BYTES, 0, wp, out, tiles_per_wg, nft, (unsigned long long*)nullptr);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 2 * A_BYTES, 0, wp, out, tiles_per_wg, nft, (unsigned long long*)nullptr);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 2 * A_BYTES, 0, wp, out, tiles_per_wg, nft, stamps);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(grid * 2);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (int i = 0; i < grid; ++i) if (h[2 * i + 1]) ghz.push_back((double)h[2 * i] / h[2 * i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double clk = ghz.empty() ? 0 : ghz[ghz.size() / 2];
    const double flop_exec = (double)grid * tiles_per_wg * ROWS * COLS * (KST * 32) * 2.0;
    const double flop_layer = 256.0 * 361 * (2.0 * 728 * 728 + 18.0 * 728);
    const double mfma_per_simd = (double)tiles_per_wg * KST * MI * NJ * 2.0;
    printf("%-34s VALU/MFMA %d  %3d x %3d, %d tiles/wg  %.4f ms  %5.0f TFLOP/s executed;  as the layer: %.3f of 2.5 PF;  clock %.2f GHz,  %.1f cycles per MFMA and SIMD;"
           "  at 1.45 GHz: %.4f ms = %.3f\n", name, VPM, ROWS, COLS, tiles_per_wg, ms, flop_exec / ms / 1e9, flop_layer / ms / 1e9 / 2500.0, clk,
           ms * 1e-3 * clk * 1e9 / mfma_per_simd, ms * clk / 1.45, flop_layer / (ms * clk / 1.45) / 1e9 / 2500.0);
    return 0;
}

int main(int argc, char** argv) {
    uint4* wp; float* out; unsigned long long* stamps;
    const int NFT = 48;
    const size_t wbytes = (size_t)KST * NFT * 1024;
    CK(hipMalloc(&wp, wbytes)); CK(hipMalloc(&out, (size_t)256 * 8 * 512 * 4)); CK(hipMalloc(&stamps, 256 * 16));
    {
        std::vector<unsigned short> h(wbytes / 2);
        unsigned s = 12345;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = 0x2c00 + ((s >> 9) & 0x3ff) + ((s >> 31) << 15); }
        CK(hipMemcpy(wp, h.data(), wbytes, hipMemcpyHostToDevice));
    }
    const int reps = 1500;
    // the shipped shape, the taller one, the review's -- each without vector work, with the real kernel's ~4 per MFMA, and (for
    // the shape that builds A twice) with 8
    if (run<1, 8, 5, 6, 0>("80 x 768 (shipped)", wp, out, stamps, 5, NFT, reps)) return 1;
    if (run<1, 8, 5, 6, 4>("80 x 768 (shipped)", wp, out, stamps, 5, NFT, reps)) return 1;
    if (run<1, 8, 6, 6, 0>("96 x 768 (5 map rows)", wp, out, stamps, 4, NFT, reps)) return 1;
    if (run<1, 8, 6, 6, 4>("96 x 768 (5 map rows)", wp, out, stamps, 4, NFT, reps)) return 1;
    // 192 x 384: a workgroup streams 24 of the 48 n-fragments (the column half it owns): nft stays 48 (the array's stride)
    if (run<2, 4, 6, 6, 0>("192 x 384 (review)", wp, out, stamps, 4, NFT, reps)) return 1;
    if (run<2, 4, 6, 6, 4>("192 x 384 (review)", wp, out, stamps, 4, NFT, reps)) return 1;
    if (run<2, 4, 6, 6, 8>("192 x 384 (review), A built twice", wp, out, stamps, 4, NFT, reps)) return 1;
    // 160 x 384 (two 4-row tiles x a column half): 120 accumulators, rows 152 of 160
    if (run<2, 4, 5, 6, 8>("160 x 384, A built twice", wp, out, stamps, 5, NFT, reps)) return 1;
    return 0;
}
