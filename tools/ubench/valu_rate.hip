// VALU issue-rate microbenchmark for gfx950: cycles per instruction per SIMD for the vector instructions the depthwise
// stages are built from, at 1 and 2 waves per SIMD, alone and next to an MFMA-only wave on the same SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));

enum Op { FMA, FMA_MIX, PK_FMA, CVT_PK_BF16, CVT_PK_F16, PK_MAX_I16, LSHL, AND, CVT_F32_F16, CVT_F32_F16_SDWA, PERMLANE16_SWAP, NOPS };
static const char* kNames[] = {"v_fma_f32", "v_fma_mix_f32", "v_pk_fma_f32", "v_cvt_pk_bf16_f32", "v_cvt_pk_f16_f32", "v_pk_max_i16",
                               "v_lshlrev_b32", "v_and_b32", "v_cvt_f32_f16", "v_cvt_f32_f16_sdwa", "v_permlane16_swap", "?"};

template <int OP>
__device__ __forceinline__ void op(float (&x)[8], int i) {
    float& d = x[i & 7]; float& s = x[(i + 3) & 7];
    if constexpr (OP == FMA) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(d) : "v"(s));
    else if constexpr (OP == FMA_MIX) asm volatile("v_fma_mix_f32 %0, %1, %1, %0 op_sel_hi:[1,0,0]" : "+v"(d) : "v"(s));
    else if constexpr (OP == PK_FMA) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2& dd = *reinterpret_cast<f2*>(&x[(i & 3) * 2]);
        asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(dd));
    }
    else if constexpr (OP == CVT_PK_BF16) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(d) : "v"(s));
    else if constexpr (OP == CVT_PK_F16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(d) : "v"(s));
    else if constexpr (OP == PK_MAX_I16) asm volatile("v_pk_max_i16 %0, %1, %1" : "=v"(d) : "v"(s));
    else if constexpr (OP == LSHL) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(d) : "v"(s));
    else if constexpr (OP == AND) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(d) : "v"(s));
    else if constexpr (OP == CVT_F32_F16) asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(d) : "v"(s));
    else if constexpr (OP == CVT_F32_F16_SDWA) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(d) : "v"(s));
    else if constexpr (OP == PERMLANE16_SWAP) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(d), "+v"(s));
}

// role: waves with (wave & mfma_mask) != 0 run MFMAs only, the others the vector op only
template <int OP>
__global__ void __launch_bounds__(512) kern(unsigned long long* out, int iters, float* sink, int mfma_waves_from) {
    f4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
    s8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    const int wave = threadIdx.x >> 6;
    const bool mf = wave >= mfma_waves_from;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (mf) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int q = 0; q < 32; ++q) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[q & 15]) : "v"(a), "v"(b));
    } else {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int q = 0; q < 128; ++q) op<OP>(x, q);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 1.2345f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int OP>
void run(int threads, int mfma_from, const char* what) {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 256 * 8 * 8); hipMalloc(&sink, 4);
    const int iters = 1000;
    auto k = kern<OP>;
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, iters, sink, mfma_from);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, iters, sink, mfma_from); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    double tv = 0, tm = 0; int nv = 0, nm = 0;
    for (int bidx = 0; bidx < 256; ++bidx) for (int w = 0; w < nw; ++w) { if (w >= mfma_from) { tm += h[bidx * 8 + w]; ++nm; } else { tv += h[bidx * 8 + w]; ++nv; } }
    // cycle counter ticks at 100 MHz; SIMD clock assumed 2.4 GHz -> 24 cycles per tick
    const double cyc = 24.0;
    printf("%-22s %-34s kernel %.3f ms", kNames[OP], what, ms);
    if (nv) printf("  VALU wave: %.2f cycles/instr", tv / nv * cyc / (iters * 128.0));
    if (nm) printf("  MFMA wave: %.2f cycles/MFMA", tm / nm * cyc / (iters * 32.0));
    printf("\n");
    hipFree(d); hipFree(sink);
}

template <int OP> void all() {
    run<OP>(256, 99, "1 wave/SIMD, VALU only");
    run<OP>(512, 99, "2 waves/SIMD, VALU only");
    run<OP>(512, 4, "1 VALU wave + 1 MFMA wave per SIMD");
}

int main() {
    run<FMA>(256, 0, "1 wave/SIMD, MFMA only");
    run<FMA>(512, 0, "2 waves/SIMD, MFMA only");
    all<FMA>(); all<FMA_MIX>(); all<PK_FMA>(); all<CVT_PK_BF16>(); all<CVT_PK_F16>(); all<PK_MAX_I16>(); all<LSHL>(); all<AND>();
    all<CVT_F32_F16>(); all<CVT_F32_F16_SDWA>(); all<PERMLANE16_SWAP>();
    // permlane16_swap semantics: lane l holds (l, 1000 + l) in (v0, v1)
    {
        int* dv; hipMalloc(&dv, 64 * 2 * 4);
        auto sem = +[](int* o) {};
        (void)sem;
        hipFree(dv);
    }
    return 0;
}
