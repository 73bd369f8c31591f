// v_permlane16_swap_b32 / v_permlane32_swap_b32 semantics on gfx950 (through the builtins, so that hipcc pads the
// hazards): prints what each lane holds afterwards.  Lane l starts with (v0, v1) = (l, 1000 + l).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
    unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
    u2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r.x; o[64 + threadIdx.x] = r.y;
    u2 q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[128 + threadIdx.x] = q.x; o[192 + threadIdx.x] = q.y;
    unsigned c = threadIdx.x, d = 1000 + threadIdx.x;
    asm volatile("s_nop 4\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 4" : "+v"(c), "+v"(d));
    o[256 + threadIdx.x] = c; o[320 + threadIdx.x] = d;
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 384 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[384]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[6] = {"permlane16_swap .x", "permlane16_swap .y", "permlane32_swap .x", "permlane32_swap .y", "asm p16swap v0", "asm p16swap v1"};
    for (int r = 0; r < 6; ++r) { printf("%-20s:", nm[r]); for (int i = 0; i < 64; i += 4) printf(" %4u", h[r * 64 + i]); printf("\n"); }
    return 0;
}
