// Device-side ROC / Youden's J for the consumer's threshold search (SURVEY.md section 8f row 3;
// biscuit/threshold.py:145-155 tile-level prediction threshold, :417-426 tile-level uncertainty threshold over
// up to 1.6 M rows).  Result contract = the reference's `thresh[argmax(tpr - fpr)]` on
// sklearn.metrics.roc_curve(y_true, y_score): first maximum, the curve starting at (0, 0) with threshold +inf,
// one point per distinct score in descending order, rates = integer counts / totals in float64.
// (roc_curve drops collinear interior points first; such a point is never a maximum of tpr - fpr -- J is
// strictly monotone along a horizontal or vertical run and a run's first point is kept -- so the first
// maximum over all distinct points is the same threshold.)
//
// HBM-bound integer/byte work: one radix sort of (float64 score, uint8 label) pairs (rocPRIM), one inclusive
// scan of the labels, one pass that turns run ends into (J, index) candidates and reduces them with a
// first-maximum operator.  No matrix cores.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "bq_common.h"

namespace {

struct Cand {
    double j;
    long long idx;      // position of the run end in the sorted order; -1 = the (0, 0) point
};

struct FirstMax {
    __host__ __device__ Cand operator()(const Cand& a, const Cand& b) const {
        // larger J wins; equal J: the earlier point of the curve (smaller index) wins; NaN never wins
        if (a.j > b.j) return a;
        if (b.j > a.j) return b;
        if (a.j == b.j) return a.idx <= b.idx ? a : b;
        return (a.j == a.j) ? a : b;
    }
};

__global__ void __launch_bounds__(256) roc_candidates_kernel(const double* __restrict__ score,
                                                             const unsigned* __restrict__ cum_pos, long long n,
                                                             Cand* __restrict__ cand) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double neg_inf = -__builtin_inf();
    Cand c{neg_inf, i};
    const bool run_end = (i == n - 1) || (score[i + 1] != score[i]);
    if (run_end) {
        const double P = (double)cum_pos[n - 1], N = (double)(n - (long long)cum_pos[n - 1]);
        const double tps = (double)cum_pos[i], fps = (double)(i + 1 - (long long)cum_pos[i]);
        c.j = tps / P - fps / N;        // float64 division of exact integers, like tps / tps[-1] in sklearn
    }
    cand[i] = c;
}

__global__ void roc_finish_kernel(const Cand* __restrict__ best, const double* __restrict__ score,
                                  const unsigned* __restrict__ cum_pos, long long n, double* __restrict__ out) {
    // out: [threshold, J, tpr, fpr, n_pos, n_neg]
    const double P = (double)cum_pos[n - 1], N = (double)(n - (long long)cum_pos[n - 1]);
    Cand b = *best;
    const Cand origin{0.0, -1};
    b = FirstMax()(origin, b);          // the curve's first point (0, 0, +inf) precedes every run end
    out[4] = P;
    out[5] = N;
    if (b.idx < 0) {
        out[0] = __builtin_inf(); out[1] = 0.0; out[2] = 0.0; out[3] = 0.0;
    } else {
        out[0] = score[b.idx];
        out[1] = b.j;
        out[2] = (double)cum_pos[b.idx] / P;
        out[3] = (double)(b.idx + 1 - (long long)cum_pos[b.idx]) / N;
    }
}

struct U8ToU32 {
    __host__ __device__ unsigned operator()(unsigned char v) const { return v ? 1u : 0u; }
};

}  // namespace

size_t roc_workspace_bytes(long long n) {
    if (n <= 0) return 0;
    size_t sort_b = 0, scan_b = 0, red_b = 0;
    (void)rocprim::radix_sort_pairs_desc(nullptr, sort_b, (double*)nullptr, (double*)nullptr, (unsigned char*)nullptr,
                                         (unsigned char*)nullptr, (size_t)n, 0, 64, (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, scan_b,
                                  rocprim::make_transform_iterator((unsigned char*)nullptr, U8ToU32()),
                                  (unsigned*)nullptr, (size_t)n, rocprim::plus<unsigned>(), (hipStream_t)0);
    (void)rocprim::reduce(nullptr, red_b, (Cand*)nullptr, (Cand*)nullptr, Cand{0.0, 0}, (size_t)n, FirstMax(), (hipStream_t)0);
    size_t tmp = sort_b > scan_b ? sort_b : scan_b;
    if (red_b > tmp) tmp = red_b;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return al(tmp) + al((size_t)n * 8) + al((size_t)n) + al((size_t)n * 4) + al((size_t)n * sizeof(Cand)) + 256;
}

// score [n] float64, label [n] uint8 (non-zero = positive), all on the device; out [6] float64 on the device.
int launch_roc_youden(const double* score, const unsigned char* label, long long n, unsigned char* ws, size_t ws_bytes,
                      double* out, hipStream_t s) {
    if (n <= 0) return (int)hipErrorInvalidValue;
    if (ws_bytes < roc_workspace_bytes(n)) return (int)hipErrorInvalidValue;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t sort_b = 0, scan_b = 0, red_b = 0;
    (void)rocprim::radix_sort_pairs_desc(nullptr, sort_b, (double*)nullptr, (double*)nullptr, (unsigned char*)nullptr,
                                         (unsigned char*)nullptr, (size_t)n, 0, 64, s);
    (void)rocprim::inclusive_scan(nullptr, scan_b, rocprim::make_transform_iterator((unsigned char*)nullptr, U8ToU32()),
                                  (unsigned*)nullptr, (size_t)n, rocprim::plus<unsigned>(), s);
    (void)rocprim::reduce(nullptr, red_b, (Cand*)nullptr, (Cand*)nullptr, Cand{0.0, 0}, (size_t)n, FirstMax(), s);
    size_t tmp = sort_b > scan_b ? sort_b : scan_b;
    if (red_b > tmp) tmp = red_b;
    unsigned char* p = ws;
    void* d_tmp = p; p += al(tmp);
    double* k_sorted = (double*)p; p += al((size_t)n * 8);
    unsigned char* v_sorted = p; p += al((size_t)n);
    unsigned* cum = (unsigned*)p; p += al((size_t)n * 4);
    Cand* cand = (Cand*)p; p += al((size_t)n * sizeof(Cand));
    Cand* best = (Cand*)p;

    hipError_t e = rocprim::radix_sort_pairs_desc(d_tmp, sort_b, score, k_sorted, label, v_sorted, (size_t)n, 0, 64, s);
    if (e != hipSuccess) return (int)e;
    e = rocprim::inclusive_scan(d_tmp, scan_b, rocprim::make_transform_iterator(v_sorted, U8ToU32()), cum, (size_t)n,
                                rocprim::plus<unsigned>(), s);
    if (e != hipSuccess) return (int)e;
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(roc_candidates_kernel, dim3(grid), dim3(256), 0, s, k_sorted, cum, n, cand);
    e = rocprim::reduce(d_tmp, red_b, cand, best, Cand{-__builtin_inf(), (long long)n}, (size_t)n, FirstMax(), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(roc_finish_kernel, dim3(1), dim3(1), 0, s, best, k_sorted, cum, n, out);
    return (int)hipGetLastError();
}
