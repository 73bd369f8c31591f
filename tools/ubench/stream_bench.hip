// Standalone timing harness of csrc/kernels_stream.hip (GPU only): random input, 128 -> 128 and 64 -> 128 @147x147 at batch
// 256, HIP events over 20 launches, a checksum of the output (variants that only change the schedule must agree), and a
// plain 16-byte copy of the same bytes as the box's calibration.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DSTREAM_NW=16 -DSTREAM_ABL=1 ...] -o stream_bench stream_bench.hip
#include "../../biscuit_amd/csrc/kernels_stream.hip"
#include "../../biscuit_amd/csrc/kernels_front.hip"

#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void copy16(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void fill_rand(unsigned short* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = ((h & 0xffff) / 65536.0f - 0.5f) * 4.0f;
        p[i] = __builtin_bit_cast(unsigned short, (_Float16)v);
    }
}
__global__ void checksum(const unsigned* p, size_t n, unsigned long long* out) {
    unsigned long long s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i] * (unsigned long long)(i % 1000003 + 1);
    atomicAdd(out, s);
}

// chip clock while a kernel runs: one wave on a non-blocking stream reads s_memtime (core cycles) and s_memrealtime (100 MHz)
// around ~2 ms of s_sleep; printed as GHz.  (The clock the power management holds depends on the instruction mix.)
__global__ void clock_probe(unsigned long long* out, int spins) {
    unsigned long long c0, r0, c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0));
    for (int i = 0; i < spins; ++i) asm volatile("s_sleep 127");
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1));
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}
static hipStream_t g_ps;
static unsigned long long* g_pc;
static void probe_start() { hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, g_ps, g_pc, 400); }
static double probe_ghz() {
    unsigned long long h[2];
    hipStreamSynchronize(g_ps);
    hipMemcpy(h, g_pc, 16, hipMemcpyDeviceToHost);
    return h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256, H = 147, W = 147;
    const char* tag = argc > 2 ? argv[2] : "";
    const size_t px = (size_t)n * H * W;
    unsigned short *in, *out; uint4* wp; float *dw, *sc, *bi; unsigned long long* cs;
    CK(hipMalloc(&in, px * 128 * 2 + 8192)); CK(hipMalloc(&out, px * 128 * 2)); CK(hipMalloc(&wp, 32768)); CK(hipMalloc(&dw, 9 * 128 * 4));
    CK(hipMalloc(&sc, 512)); CK(hipMalloc(&bi, 512)); CK(hipMalloc(&cs, 8));
    in += 2048;   // front pad
    fill_rand<<<2048, 256>>>(in, px * 128, 1);
    fill_rand<<<64, 256>>>((unsigned short*)wp, 16384, 2);
    {
        std::vector<float> h(9 * 128), s(128), b(128);
        for (int i = 0; i < 9 * 128; ++i) h[i] = ((i * 37 % 101) / 101.0f - 0.5f) * 0.6f;
        for (int i = 0; i < 128; ++i) { s[i] = 0.2f + (i % 7) * 0.01f; b[i] = (i % 5) * 0.1f - 0.2f; }
        CK(hipMemcpy(dw, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(sc, s.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(bi, b.data(), 512, hipMemcpyHostToDevice));
    }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipStreamCreateWithFlags(&g_ps, hipStreamNonBlocking)); CK(hipMalloc(&g_pc, 16));
    { probe_start(); printf("%-28s clock, idle chip: %.2f GHz\n", tag, probe_ghz()); }
    for (int cin : {128, 64}) {
        const double gb = (double)px * (cin + 128) * 2 / 1e9;
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < 3; ++i) if (launch_sepconv_stream(2, cin, 128, false, in, wp, dw, sc, bi, out, n, H, W, 1, 256, 0)) { printf("launch failed\n"); return 1; }
            CK(hipEventRecord(a));
            for (int i = 0; i < 20; ++i) { launch_sepconv_stream(2, cin, 128, false, in, wp, dw, sc, bi, out, n, H, W, 1, 256, 0); if (i == 4) probe_start(); }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); const double ghz = probe_ghz();
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
            CK(hipMemset(cs, 0, 8));
            checksum<<<1024, 256>>>((const unsigned*)out, px * 64, cs);
            unsigned long long h; CK(hipMemcpy(&h, cs, 8, hipMemcpyDeviceToHost));
            printf("%-28s cin %3d: %.4f ms  %.0f GB/s  checksum %016llx  clock %.2f GHz\n", tag, cin, ms, gb / ms * 1e3, h, ghz);
        }
    }
    {   // fused block tail: y1 = in (128 ch), x = second tensor (64 ch), out 74 x 74 x 128
        unsigned short* x; uint4* wr; unsigned short* out2;
        CK(hipMalloc(&x, px * 64 * 2)); CK(hipMalloc(&wr, 16384)); CK(hipMalloc(&out2, (size_t)n * 74 * 74 * 128 * 2));
        fill_rand<<<2048, 256>>>(x, px * 64, 7);
        fill_rand<<<64, 256>>>((unsigned short*)wr, 8192, 9);
        const double gb = ((double)px * 128 + (double)n * 74 * 74 * (64 + 128)) * 2 / 1e9;
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < 3; ++i) if (launch_block_tail(2, 128, 128, 64, in, wp, dw, sc, bi, x, wr, sc, bi, out2, n, H, W, 256, 0)) { printf("tail launch failed\n"); return 1; }
            CK(hipEventRecord(a));
            for (int i = 0; i < 20; ++i) { launch_block_tail(2, 128, 128, 64, in, wp, dw, sc, bi, x, wr, sc, bi, out2, n, H, W, 256, 0); if (i == 4) probe_start(); }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); const double ghz = probe_ghz();
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
            CK(hipMemset(cs, 0, 8));
            checksum<<<1024, 256>>>((const unsigned*)out2, (size_t)n * 74 * 74 * 64, cs);
            unsigned long long h; CK(hipMemcpy(&h, cs, 8, hipMemcpyDeviceToHost));
            printf("%-28s tail    : %.4f ms  %.0f GB/s  checksum %016llx  clock %.2f GHz\n", tag, ms, gb / ms * 1e3, h, ghz);
        }
    }
    {   // cooperative block-3 tail: y1 74 x 74 x 256, x 74 x 74 x 128 -> 37 x 37 x 256
        const size_t p3 = (size_t)n * 74 * 74;
        unsigned short *y3, *x3, *o3; uint4 *w3, *wr3; float *sc3, *dw3;
        CK(hipMalloc(&y3, p3 * 256 * 2 + 8192)); CK(hipMalloc(&x3, p3 * 128 * 2)); CK(hipMalloc(&o3, (size_t)n * 37 * 37 * 256 * 2));
        CK(hipMalloc(&w3, 131072)); CK(hipMalloc(&wr3, 65536)); CK(hipMalloc(&sc3, 1024)); CK(hipMalloc(&dw3, 9 * 256 * 4));
        y3 += 2048;
        fill_rand<<<2048, 256>>>(y3, p3 * 256, 21); fill_rand<<<2048, 256>>>(x3, p3 * 128, 22);
        fill_rand<<<64, 256>>>((unsigned short*)w3, 65536, 23); fill_rand<<<64, 256>>>((unsigned short*)wr3, 32768, 24);
        {
            std::vector<float> h(9 * 256), sv(256);
            for (int i = 0; i < 9 * 256; ++i) h[i] = ((i * 37 % 101) / 101.0f - 0.5f) * 0.6f;
            for (int i = 0; i < 256; ++i) sv[i] = 0.1f + (i % 7) * 0.01f;
            CK(hipMemcpy(dw3, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sc3, sv.data(), 1024, hipMemcpyHostToDevice));
        }
        const double gb = ((double)p3 * 256 + (double)n * 37 * 37 * (128 + 256)) * 2 / 1e9;
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < 3; ++i) if (launch_block_tail(2, 256, 256, 128, y3, w3, dw3, sc3, sc3, x3, wr3, sc3, sc3, o3, n, 74, 74, 256, 0)) { printf("coop launch failed\n"); return 1; }
            CK(hipEventRecord(a));
            for (int i = 0; i < 20; ++i) { launch_block_tail(2, 256, 256, 128, y3, w3, dw3, sc3, sc3, x3, wr3, sc3, sc3, o3, n, 74, 74, 256, 0); if (i == 4) probe_start(); }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); const double ghz = probe_ghz();
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
            CK(hipMemset(cs, 0, 8));
            checksum<<<1024, 256>>>((const unsigned*)o3, (size_t)n * 37 * 37 * 128, cs);
            unsigned long long h; CK(hipMemcpy(&h, cs, 8, hipMemcpyDeviceToHost));
            printf("%-28s coop    : %.4f ms  %.0f GB/s  checksum %016llx  clock %.2f GHz\n", tag, ms, gb / ms * 1e3, h, ghz);
        }
    }
    {   // fused front: uint8 tiles -> conv2 output (147 x 147 x 64)
        uint8_t* u8; unsigned long long* st; uint4 *ws16, *wc16; unsigned short* o3;
        const size_t tb = (size_t)n * 299 * 299 * 3;
        CK(hipMalloc(&u8, tb)); CK(hipMalloc(&st, (size_t)n * 16)); CK(hipMalloc(&ws16, 4096)); CK(hipMalloc(&wc16, 36864));
        CK(hipMalloc(&o3, px * 64 * 2));
        fill_rand<<<2048, 256>>>((unsigned short*)u8, tb / 2, 11);
        fill_rand<<<16, 256>>>((unsigned short*)ws16, 2048, 12);
        fill_rand<<<64, 256>>>((unsigned short*)wc16, 18432, 13);
        {
            std::vector<unsigned long long> h(2 * n);
            for (int i = 0; i < n; ++i) { h[2 * i] = 128ull * 268203; h[2 * i + 1] = (128ull * 128 + 5400) * 268203; }
            CK(hipMemcpy(st, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        }
        const double gb = ((double)tb + (double)px * 64 * 2) / 1e9;
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < 3; ++i) if (launch_front(2, u8, st, ws16, sc, bi, wc16, sc, bi, o3, n, 256, 0)) { printf("front launch failed\n"); return 1; }
            CK(hipEventRecord(a));
            for (int i = 0; i < 20; ++i) { launch_front(2, u8, st, ws16, sc, bi, wc16, sc, bi, o3, n, 256, 0); if (i == 4) probe_start(); }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); const double ghz = probe_ghz();
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
            CK(hipMemset(cs, 0, 8));
            checksum<<<1024, 256>>>((const unsigned*)o3, px * 32, cs);
            unsigned long long h; CK(hipMemcpy(&h, cs, 8, hipMemcpyDeviceToHost));
            printf("%-28s front   : %.4f ms  %.0f GB/s  checksum %016llx  clock %.2f GHz\n", tag, ms, gb / ms * 1e3, h, ghz);
        }
    }
    {   // calibration: copy of 1.42 GB (read + write 2.83 GB)
        const size_t n16 = px * 128 * 2 / 16;
        for (int i = 0; i < 3; ++i) copy16<<<256 * 8, 256>>>((const uint4*)in, (uint4*)out, n16);
        CK(hipEventRecord(a));
        for (int i = 0; i < 20; ++i) { copy16<<<256 * 8, 256>>>((const uint4*)in, (uint4*)out, n16); if (i == 4) probe_start(); }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); const double ghz = probe_ghz();
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
        printf("%-28s copy16 of the same bytes: %.4f ms  %.0f GB/s\n", tag, ms, (double)n16 * 32 / 1e9 / ms * 1e3);
    }
    return 0;
}
