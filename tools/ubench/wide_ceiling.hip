// Structural ceiling of the 728 -> 728 @19x19 kernel (csrc/kernels_wide.hip), measured (round-3 review, item 2): the SAME matrix
// loop -- 8 waves x (80 rows x 96 channels) of v_mfma_f32_16x16x32_f16 with the accumulators tied in place, A fragments read
// from LDS (ds_read_b128, two k-steps per 64-channel chunk, one workgroup barrier per chunk), B fragments (the pointwise
// weights in [k-step][n-fragment][lane] order) streamed from L2 one k-step ahead, 23 k-steps per tile, five tiles per
// persistent workgroup, one workgroup per CU -- with NO depthwise stage, no halo DMA, no epilogue arithmetic (the
// accumulators are summed into one store per tile so that nothing is dead).  What this loop reaches at the clock the chip
// holds is the most the real kernel could reach if its depthwise stage, halo traffic and epilogue were free.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o wide_ceiling wide_ceiling.hip && ./wide_ceiling
// Variants (argv[1]): 0 = as described; 1 = no barrier per chunk; 2 = A fragments kept in registers (no LDS reads);
// 3 = B fragments kept in registers (no L2 stream); 4 = neither (bare MFMA issue).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int MI = 5, NJ = 6, KST = 23, NFT = 48, NWAVE = 8, ROWS = 80, CH = 64;
constexpr int AST = CH * 2 + 16;                    // A chunk row stride (odd number of 16-byte slots)
constexpr int A_BYTES = ROWS * AST;

__device__ __forceinline__ void mfma(f32x4v& acc, const u32x4& b, const u32x4& a) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(b), "v"(a));
}

template <int VAR>
__global__ void __launch_bounds__(512) ceiling_kernel(const uint4* __restrict__ wp, float* __restrict__ out, int tiles_per_wg,
                                                      unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // two A chunks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * A_BYTES / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003800u + (i * 2654435761u >> 20);   // f16 values near 1
    __syncthreads();
    const int r16 = lane & 15, kg = lane >> 4;
    const uint4* wq = wp + (size_t)(wave * NJ) * 64 + lane;
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
        if (VAR == 2 || VAR == 4)
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const u32x4*>(smem + (i * 16 + r16) * AST + kg * 16);
#pragma unroll 1
        for (int ks = 0; ks < KST; ++ks) {
            if ((ks & 1) == 0 && VAR != 1 && VAR != 4) __syncthreads();          // one barrier per 64-channel chunk
            const unsigned char* ab = smem + ((ks >> 1) & 1) * A_BYTES + (ks & 1) * 64;
            if (VAR != 2 && VAR != 4)
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const u32x4*>(ab + (i * 16 + r16) * AST + kg * 16);
            const int kn = ks + 1 < KST ? ks + 1 : 0;                            // next k-step's fragments (wraps to the next tile's first)
            if (VAR != 3 && VAR != 4)
#pragma unroll
                for (int j = 0; j < NJ; ++j) bn[j] = __builtin_bit_cast(u32x4, wq[((size_t)kn * NFT + j) * 64]);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) mfma(acc[i][j], bq[j], a[i]);
            if (VAR != 3 && VAR != 4)
#pragma unroll
                for (int j = 0; j < NJ; ++j) bq[j] = bn[j];
        }
        asm volatile("s_nop 7\n s_nop 7\n s_nop 3" ::: "memory");
        f32x4v s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) s += acc[i][j];
        out[((size_t)(blockIdx.x * tiles_per_wg + t) * 512 + tid)] = s[0] + s[1] + s[2] + s[3];
    }
    if (stamps && tid == 0) {
        stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
        stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int VAR>
int run(const uint4* wp, float* out, unsigned long long* stamps, int reps) {
    const int tiles_per_wg = 5, grid = 256;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto kern = ceiling_kernel<VAR>;
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 2 * A_BYTES, 0, wp, out, tiles_per_wg, (unsigned long long*)nullptr);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 2 * A_BYTES, 0, wp, out, tiles_per_wg, (unsigned long long*)nullptr);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    // clock: stamped in a run of its own, behind >= 1 s of back-to-back launches (the chip has settled)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 2 * A_BYTES, 0, wp, out, tiles_per_wg, stamps);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(grid * 2);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (int i = 0; i < grid; ++i) if (h[2 * i + 1]) ghz.push_back((double)h[2 * i] / h[2 * i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double clk = ghz.empty() ? 0 : ghz[ghz.size() / 2];
    const double flop_exec = (double)grid * tiles_per_wg * ROWS * (NFT * 16) * (KST * 32) * 2.0;      // executed (80 x 768 x 736)
    const double flop_layer = 256.0 * 361 * (2.0 * 728 * 728 + 18.0 * 728);                            // the layer's algorithmic FLOPs
    const char* names[] = {"LDS A + L2 B + barrier per chunk (the real loop, no depthwise)", "no barrier", "A in registers", "B in registers", "bare MFMA issue"};
    printf("variant %d  %-62s %.4f ms  %.0f TFLOP/s executed = %.3f of 2.5 PF;  as the layer (99.2 GFLOP): %.3f;  clock %.2f GHz,  %.1f cycles per MFMA and SIMD\n",
           VAR, names[VAR], ms, flop_exec / ms / 1e9, flop_exec / ms / 1e9 / 2500.0, flop_layer / ms / 1e9 / 2500.0, clk,
           ms * 1e-3 * clk * 1e9 / (tiles_per_wg * KST * MI * NJ * 2.0));
    return 0;
}

int main(int argc, char** argv) {
    uint4* wp; float* out; unsigned long long* stamps;
    const size_t wbytes = (size_t)KST * NFT * 1024;
    CK(hipMalloc(&wp, wbytes)); CK(hipMalloc(&out, (size_t)256 * 5 * 512 * 4)); CK(hipMalloc(&stamps, 256 * 16));
    {
        std::vector<unsigned short> h(wbytes / 2);
        unsigned s = 12345;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = 0x2c00 + ((s >> 9) & 0x3ff) + ((s >> 31) << 15); }   // random f16 in +-[0.06, 0.12)
        CK(hipMemcpy(wp, h.data(), wbytes, hipMemcpyHostToDevice));
    }
    const int reps = 2000;                      // ~0.3 s per variant: long enough for the clock to settle
    if (run<0>(wp, out, stamps, reps) || run<1>(wp, out, stamps, reps) || run<2>(wp, out, stamps, reps) ||
        run<3>(wp, out, stamps, reps) || run<4>(wp, out, stamps, reps)) return 1;
    return 0;
}
