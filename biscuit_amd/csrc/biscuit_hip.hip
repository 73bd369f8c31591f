// C ABI of libbiscuit_hip.so (see include/biscuit_hip.h): context, BQW1 weight blob,
// the Xception launch schedule, the MC-dropout head and event-based per-kernel timing.
#include "../../include/biscuit_hip.h"
#include "bq_common.h"

#include <math.h>
#include <cmath>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

namespace {

std::string g_create_error;

struct Blob { const unsigned char* p = nullptr; size_t n = 0; };

struct GemmLayer {
    std::string name;
    const void* wp = nullptr;
    const void* wp16 = nullptr;   // the same weights in 16x16x32 fragment order (728 -> 728 layers, kernels_wide.hip)
    const void* wp32 = nullptr;   // ... in 32x32x16 fragment order (strided shortcuts, kernels_respool.hip)
    const float* scale = nullptr;
    const float* bias = nullptr;
    const float* dw = nullptr;
    int cin = 0, cout = 0;   // true channel counts
    int kpad = 0;            // padded contraction length
    int nfp = 0;             // padded n-frags in wp
};

struct HeadLayer { const void* wh = nullptr; const void* wl = nullptr; const float* bias = nullptr; int k = 0; };

struct ProfRec { int cls; hipEvent_t a, b; };

}  // namespace

struct bq_ctx {
    bq_config cfg{};
    int device = 0;
    std::string err;
    unsigned char* d_blob = nullptr;
    size_t blob_bytes = 0;
    std::map<std::string, Blob> entries;
    std::map<std::string, GemmLayer> layers;
    const float* stem_w = nullptr; const float* stem_s = nullptr; const float* stem_b = nullptr;
    const void* front_ws16 = nullptr;   // "block1_conv1/w16" + "block1_conv2/wp16": the fused front kernel (kernels_front.hip)
    const void* front_wc16 = nullptr;
    const float* logits_w = nullptr; const float* logits_b = nullptr;
    HeadLayer head[2];             // hidden_0, hidden_1: weights split into two halves (kernels_head.hip)
    bool loaded = false;
    int num_cus = 256;
    float* d_srgb_lut = nullptr;   // tables of the Reinhard normaliser
    const long long* d_tile0 = nullptr;   // bq_set_tile_index_ptr
    const long long* d_tile_idx = nullptr;   // bq_set_tile_index_array
    int inflate_variant = 5;       // bq_set_option("inflate_variant"): 5 = rounds of a literal-only fast phase + a general phase (LDS), 0 = the
                                   // kernel without LDS, tables in global memory (kernels_inflate.hip; profiles/r05_inflate.txt)
    float feat_mul = 1.f;          // "act/feat_mul" of the blob: 2^k of the pooled tensor's activation exponent (weights.py: pack_blob)
    double* d_stage_stats = nullptr;   // 2 x 64-bit integer sums per tile for the staging kernel pair
    // profiling
    bool prof = false;
    std::vector<std::string> prof_names;
    std::vector<double> prof_flops, prof_bytes;
    std::vector<int64_t> prof_launches;
    std::vector<double> prof_ms;
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
};

namespace {

int fail(bq_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define HIPCHK(c, expr)                                                                 \
    do {                                                                                \
        hipError_t _e = (hipError_t)(expr);                                             \
        if (_e != hipSuccess)                                                           \
            return fail((c), BQ_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct DeviceGuard {
    int prev = -1; bool ok = false;
    explicit DeviceGuard(int dev) { ok = hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int pad16(int c) { return (c + 15) / 16 * 16; }
inline bool is16(int dtype) { return dtype == BQ_DTYPE_BF16 || dtype == BQ_DTYPE_F16; }
inline size_t esize(const bq_ctx* c) { return is16(c->cfg.dtype) ? 2 : 4; }

constexpr long long kStaged = 3LL * 299 * 299;
constexpr long long kMaxAct = 147LL * 147 * 128;
constexpr long long kMaxRes = 74LL * 74 * 128;

struct WsLayout {
    size_t a, b, c, r, staged, feat, h0, h1, state, total;
};

WsLayout ws_layout(const bq_ctx* c, int n, int mc) {
    const size_t es = esize(c);
    WsLayout L{};
    // The streaming and tail kernels (kernels_stream.hip, kernels_front.hip) read -- and mask -- a few pixels OUTSIDE the tensor
    // they walk: the 16-bit pixel in front of an image's first row (window column -1) and up to three pixels (<= 768 B) behind
    // the last row of the last image.  Every activation buffer therefore has kActPad bytes of the workspace on both sides: the
    // pad in front of A, and a pad behind each of A, B, C and R (which is also the front pad of the next one).  run_conv hands
    // these kernels workspace buffers only.
    constexpr size_t kActPad = 4096;
    size_t off = kActPad;
    auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
    L.a = take((size_t)n * kMaxAct * es + kActPad);
    L.b = take((size_t)n * kMaxAct * es + kActPad);
    L.c = take((size_t)n * kMaxAct * es + kActPad);
    L.r = take((size_t)n * kMaxRes * es + kActPad);
    L.staged = take((size_t)n * kStaged * es);
    L.feat = take((size_t)n * 2048 * 4);
    const size_t rows = (size_t)n * (mc > 0 ? mc : 1);
    L.h0 = take(rows * 1024 * 4);
    L.h1 = take(rows * 1024 * 4);
    L.state = take((size_t)n * 5 * 4);
    L.total = off;
    return L;
}

// ---- profiling -----------------------------------------------------------------
// A class sums the algorithmic FLOPs and bytes of its launches: the instances of one class differ (8 of the 25
// 728 -> 728 layers read a residual, 406 against 270 MB), and bq_profile_read reports the launch-weighted average.
int prof_class(bq_ctx* c, const std::string& name, double flops, double bytes) {
    int k = -1;
    for (size_t i = 0; i < c->prof_names.size(); ++i)
        if (c->prof_names[i] == name) { k = (int)i; break; }
    if (k < 0) {
        c->prof_names.push_back(name);
        c->prof_flops.push_back(0.0);
        c->prof_bytes.push_back(0.0);
        c->prof_launches.push_back(0);
        c->prof_ms.push_back(0.0);
        k = (int)c->prof_names.size() - 1;
    }
    c->prof_flops[k] += flops;
    c->prof_bytes[k] += bytes;
    return k;
}

struct ProfScope {
    bq_ctx* c; hipStream_t s; int cls = -1; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(bq_ctx* c_, hipStream_t s_, const std::string& name, double flops, double bytes)
        : c(c_), s(s_) {
        if (!c->prof) return;
        if (c->ev_used + 2 > c->ev_pool.size()) {
            for (int i = 0; i < 256; ++i) {
                hipEvent_t e;
                if (hipEventCreate(&e) != hipSuccess) return;
                c->ev_pool.push_back(e);
            }
        }
        cls = prof_class(c, name, flops, bytes);
        a = c->ev_pool[c->ev_used++];
        b = c->ev_pool[c->ev_used++];
        (void)hipEventRecord(a, s);
    }
    ~ProfScope() {
        if (cls < 0) return;
        (void)hipEventRecord(b, s);
        c->prof_recs.push_back({cls, a, b});
    }
};

// ---- GEMM layer launch -----------------------------------------------------------
int pick_shape(const bq_ctx* c, int prod, int nfp) {
    if (prod == PROD_IM2COL) return SHAPE_A;
    if (is16(c->cfg.dtype)) {
        static const bool s2_small = !bq_exp_env("BQ_S2_BIG");
        // 64-row tiles halve the staging tile: four workgroups per CU instead of two for N = 256
        // (128->256 @37x37: 0.161 -> 0.111 ms); for N = 128 they measured slower (0.194 -> 0.234 ms)
        if (prod == PROD_S2 && s2_small && nfp == 8) return SHAPE_K;
        switch (nfp) {
            case 4: return SHAPE_B;
            case 8: return SHAPE_C;
            case 24: return SHAPE_D;
            case 32: return SHAPE_E;
            case 48: return SHAPE_F;
            case 64: return SHAPE_G;
        }
        return -1;
    }
    if (nfp == 4) return SHAPE_B;
    if (nfp == 8) return SHAPE_C;
    return SHAPE_H;
}

struct ConvArgs {
    const char* layer;
    int prod;
    const void* in; void* out; const void* residual; void* split_tmp;
    int n, H, W, Hi, Wi;   // output / input spatial dims
    int ldi, ldo;          // row strides (elements)
    int relu;
    void* dwtmp = nullptr; // scratch for the two-kernel (depthwise + GEMM) form, >= n*H*W*ldi elements
    float* gap_out = nullptr;   // round 4: global average pool as the GEMM's epilogue -> fp32 [n][ldo]; `out` is not written.
    bool* gap_done = nullptr;   // set when that form ran (it needs the two-kernel path and H * W <= 128)
};

int run_conv(bq_ctx* c, const ConvArgs& a, hipStream_t s) {
    auto it = c->layers.find(a.layer);
    if (it == c->layers.end()) return fail(c, BQ_ERR_WEIGHTS, std::string("layer not loaded: ") + a.layer);
    const GemmLayer& L = it->second;
    const int dtype = c->cfg.dtype;
    const int shape = pick_shape(c, a.prod, L.nfp);
    if (shape < 0) return fail(c, BQ_ERR_ARG, std::string("no kernel shape for ") + a.layer);
    const int vec = is16(dtype) ? 8 : 4;
    int nsplit = 1;
    while (gemm_lds_bytes(dtype, shape, L.kpad / nsplit) > 160 * 1024) {
        nsplit *= 2;
        if ((L.kpad / nsplit) % (2 * vec) != 0 || nsplit > 8)
            return fail(c, BQ_ERR_ARG, std::string("cannot split K for ") + a.layer);
    }
    if (nsplit > 1 && (a.residual || !a.split_tmp))
        return fail(c, BQ_ERR_ARG, std::string("split-K needs a temp and no residual: ") + a.layer);

    GemmParams p{};
    p.in = a.in; p.wp = L.wp; p.dw = L.dw;
    p.M = a.n * a.H * a.W;
    p.KBtot = L.kpad / (2 * vec);
    p.NFp = L.nfp; p.Nstore = a.ldo; p.ldo = a.ldo; p.ldi = a.ldi;
    p.H = a.H; p.W = a.W; p.Hi = a.Hi; p.Wi = a.Wi;
    const double es = (double)esize(c);
    const double M = (double)p.M;
    const bool dwp = a.prod == PROD_DW || a.prod == PROD_DW_RELU;
    const double flops = 2.0 * M * L.cin * L.cout + (dwp ? 18.0 * M * L.cin : 0.0);
    const double in_rows = (double)a.n * a.Hi * a.Wi;
    const double kin = a.prod == PROD_IM2COL ? (double)a.ldi : (double)L.cin;
    const double bytes = es * (in_rows * kin * (a.prod == PROD_S2 ? 0.25 : 1.0) + M * L.cout * (a.residual ? 2.0 : 1.0)) +
                         es * (double)L.cin * L.cout;
    char cls[96];
    snprintf(cls, sizeof cls, "%s_k%d_n%d_%dx%d",
             a.prod == PROD_S2 ? "res1x1s2" : (a.prod == PROD_IM2COL ? "conv3x3" : "sepconv"), L.cin,
             L.cout, a.H, a.W);
    // Two-kernel form (depthwise kernel + 128x128-tile GEMM): always for the wide exit-flow layers
    // (K >= 1024: the fused kernel can only hold 32-64 rows of A in LDS there and re-streams the
    // 3-6 MB weight matrix per 32-64 rows); BQ_SPLIT=1 forces it for every separable conv.
    static const bool split_env = bq_exp_env("BQ_SPLIT") != nullptr;
    static const bool no_split = bq_exp_env("BQ_NO_SPLIT") != nullptr;
    const bool will_split = !no_split && (split_env || L.kpad >= 1024) && is16(dtype) && dwp &&
                            L.nfp % 4 == 0 && a.dwtmp && nsplit == 1;
    ProfScope ps(c, s, will_split ? std::string("split_") + cls : std::string(cls), will_split ? 0.0 : flops, will_split ? 0.0 : bytes);
    // round 4: the 147x147 separable convolutions of block 2 on the streaming kernel (kernels_stream.hip)
    static const bool no_stream = bq_exp_env("BQ_NO_STREAM") != nullptr;
    if (!no_stream && dwp && L.wp16 && !a.residual && nsplit == 1 && a.H == a.Hi && a.W == a.Wi && a.ldi == L.kpad &&
        a.ldo == L.cout && stream_supported(dtype, L.kpad, L.cout, a.prod == PROD_DW_RELU, a.n, a.H, a.W)) {
        const int e = launch_sepconv_stream(dtype, L.kpad, L.cout, a.prod == PROD_DW_RELU, a.in, L.wp16, L.dw, L.scale, L.bias,
                                            a.out, a.n, a.H, a.W, a.relu, c->num_cus, s);
        if (e != 0) return fail(c, BQ_ERR_HIP, std::string("launch(stream) ") + a.layer + ": " +
                                                   hipGetErrorString((hipError_t)e));
        return BQ_OK;
    }
    static const bool no_tile = bq_exp_env("BQ_NO_TILE") != nullptr;
    static const int tile_mask = bq_exp_env("BQ_TILE_MASK") ? atoi(bq_exp_env("BQ_TILE_MASK")) : 15;  // kinds enabled (bit k)
    if (!no_tile && is16(dtype) && !a.residual) {
        int kind = -1;
        if (a.prod == PROD_IM2COL && L.cin == 32 && L.cout == 64) kind = 0;
        else if (a.prod == PROD_DW && L.cin == 64 && L.cout == 128) kind = 1;
        else if (a.prod == PROD_DW && L.cin == 128 && L.cout == 128) kind = 2;
        else if (a.prod == PROD_DW_RELU && L.cin == 128 && L.cout == 256 &&
                 !(L.wp16 && a.H == a.Hi && a.W == a.Wi && nsplit == 1 &&
                   wide_supported(dtype, a.prod, L.nfp, a.H, a.W, L.kpad, a.ldo, a.ldi, a.ldo, p.M, false)))
            kind = 3;               // (block3_sepconv1: the wide kernel's 74x74 instance when its weights are there)
        if (kind >= 0 && ((tile_mask >> kind) & 1)) {
            const int e = launch_tile_conv(dtype, kind, a.in, L.wp, L.dw, L.scale, L.bias, a.out, a.n, a.H, a.W, a.Hi, a.Wi,
                                           a.relu, c->num_cus, s);
            if (e != 0) return fail(c, BQ_ERR_HIP, std::string("launch(tile) ") + a.layer + ": " +
                                                       hipGetErrorString((hipError_t)e));
            return BQ_OK;
        }
    }
    if (will_split) {
        {
            ProfScope pd(c, s, std::string("dw3x3_") + cls, 18.0 * M * L.cin, 2.0 * es * M * L.cin);
            const int e = launch_dw3x3(dtype, a.in, L.dw, a.dwtmp, a.n, a.H, a.W, a.ldi, a.prod == PROD_DW_RELU, s);
            if (e != 0) return fail(c, BQ_ERR_HIP, std::string("launch(dw3x3) ") + a.layer);
        }
        const bool gap = a.gap_out && a.gap_done && !a.residual && a.H * a.W <= 128 && a.ldo == L.cout;
        ProfScope pg(c, s, std::string(gap ? "gemm_gap_" : "gemm_") + cls, 2.0 * M * L.cin * L.cout + (gap ? M * L.cout : 0.0),
                     gap ? es * M * L.cin + 4.0 * a.n * L.cout
                         : es * (M * L.cin + M * L.cout * (a.residual ? 2.0 : 1.0)));
        p.in = a.dwtmp; p.K = L.kpad; p.k_off = 0; p.kb0 = 0; p.gap_mul = c->feat_mul;
        p.scale = L.scale; p.bias = L.bias; p.relu = a.relu; p.residual = a.residual; p.out = gap ? (void*)a.gap_out : a.out;
        if (gap) *a.gap_done = true;
        // round 4: one image's pixels x 256 channels per workgroup on 16x16x32 fragments (kernels_exit.hip) when the layer's
        // weights are there in that order -- block 14
        static const bool no_exit = bq_exp_env("BQ_NO_EXIT") != nullptr;
        if (!no_exit && L.wp16 && !a.residual && a.ldi == L.kpad && a.ldo == L.cout &&
            exit_supported(dtype, L.kpad, L.cout, a.H * a.W, a.n)) {
            const int e = launch_exit_gemm(dtype, a.dwtmp, L.wp16, L.scale, L.bias, a.out, gap ? a.gap_out : nullptr, a.n,
                                           a.H * a.W, L.kpad, L.cout, a.relu, c->feat_mul, s);
            if (e != 0) return fail(c, BQ_ERR_HIP, std::string("launch(exit_gemm) ") + a.layer + ": " +
                                                       hipGetErrorString((hipError_t)e));
            return BQ_OK;
        }
        const int e = launch_gemm_tile(dtype, p, false, s, gap ? 1 : 0);
        if (e != 0) return fail(c, BQ_ERR_HIP, std::string("launch(gemm_tile) ") + a.layer + ": " +
                                                   hipGetErrorString((hipError_t)e));
        return BQ_OK;
    }
    // strided shortcut convolutions with many channels (blocks 4 and 13: K = 256 / 736): a plain GEMM whose A rows are the
    // even pixels of the input map -- the 128 x 128-tile kernel (block 13: 0.13 -> 0.085 ms against the fused-producer form; block 4: the same 0.09 ms)
    static const bool no_s2tile = bq_exp_env("BQ_NO_S2TILE") != nullptr;
    if (!no_s2tile && a.prod == PROD_S2 && is16(dtype) && L.kpad >= 256 && L.nfp % 4 == 0 && nsplit == 1 && !a.residual) {
        p.K = L.kpad; p.k_off = 0; p.kb0 = 0;
        p.scale = L.scale; p.bias = L.bias; p.relu = a.relu; p.residual = nullptr; p.out = a.out;
        const int e = launch_gemm_tile(dtype, p, true, s);
        if (e != 0) return fail(c, BQ_ERR_HIP, std::string("launch(gemm_tile s2) ") + a.layer + ": " +
                                                   hipGetErrorString((hipError_t)e));
        return BQ_OK;
    }
    static const bool no_wide = bq_exp_env("BQ_NO_WIDE") != nullptr;
    if (!no_wide && nsplit == 1 && L.wp16 && a.H == a.Hi && a.W == a.Wi &&
        wide_supported(dtype, a.prod, L.nfp, a.H, a.W, L.kpad, a.ldo, a.ldi, a.ldo, p.M, a.residual != nullptr)) {
        p.K = L.kpad; p.k_off = 0; p.kb0 = 0;
        p.scale = L.scale; p.bias = L.bias; p.relu = a.relu; p.residual = a.residual; p.out = a.out;
        const int e = launch_sepconv_wide(dtype, a.prod, p, L.wp16, c->num_cus, s);
        if (e != 0) return fail(c, BQ_ERR_HIP, std::string("launch(wide) ") + a.layer + ": " +
                                                   hipGetErrorString((hipError_t)e));
        return BQ_OK;
    }
    static const bool no_pipe = bq_exp_env("BQ_NO_PIPE") != nullptr;
    if (!no_pipe && nsplit == 1 && pipe_supported(dtype, a.prod, L.nfp, a.W, L.kpad)) {
        p.K = L.kpad; p.k_off = 0; p.kb0 = 0;
        p.scale = L.scale; p.bias = L.bias; p.relu = a.relu; p.residual = a.residual; p.out = a.out;
        const int e = launch_sepconv_pipe(dtype, a.prod, p, s);
        if (e != 0) return fail(c, BQ_ERR_HIP, std::string("launch(pipe) ") + a.layer + ": " +
                                                   hipGetErrorString((hipError_t)e));
        return BQ_OK;
    }
    for (int sp = 0; sp < nsplit; ++sp) {
        const bool last = sp == nsplit - 1;
        p.K = L.kpad / nsplit;
        p.k_off = sp * p.K;
        p.kb0 = sp * (p.K / (2 * vec));
        p.scale = L.scale;
        p.bias = last ? L.bias : nullptr;
        p.relu = last ? a.relu : 0;
        p.residual = last ? (nsplit > 1 ? a.split_tmp : a.residual) : (sp > 0 ? a.split_tmp : nullptr);
        p.out = last ? a.out : a.split_tmp;
        const int e = launch_gemm(dtype, a.prod, shape, p, s);
        if (e != 0) return fail(c, BQ_ERR_HIP, std::string("launch ") + a.layer + ": " +
                                                   hipGetErrorString((hipError_t)e));
    }
    return BQ_OK;
}

struct Tap {
    const char* want = nullptr;   // requested activation name (null: none)
    float* out = nullptr;
    size_t out_elems = 0;
    int64_t written = -1;
};

// returns 1 if the tap matched (caller stops), 0 otherwise, <0 on error
int tap_nhwc(bq_ctx* c, Tap* t, const char* name, const void* buf, int n, int H, int W, int C, int ld,
             hipStream_t s) {
    if (!t || !t->want || strcmp(t->want, name) != 0) return 0;
    const long long rows = (long long)n * H * W;
    if ((size_t)(rows * C) > t->out_elems) return fail(c, BQ_ERR_ARG, "debug output too small");
    const int e = launch_to_f32_nhwc(buf, rows, C, ld, t->out, c->cfg.dtype, s);
    if (e) return fail(c, BQ_ERR_HIP, "debug copy failed");
    t->written = rows * C;
    return 1;
}

#define RUN(expr) do { int _r = (expr); if (_r != BQ_OK) return _r; } while (0)
#define TAP(name, buf, H, W, C, ld) \
    do { int _t = tap_nhwc(c, tap, name, buf, n, H, W, C, ld, s); if (_t) return _t < 0 ? _t : BQ_OK; } while (0)

// End of a block with a strided shortcut: out = maxpool3x3/s2(y) + BN(conv1x1/s2(x)).  bf16: one kernel
// (kernels_respool.hip) unless the shortcut tensor itself was asked for; otherwise the shortcut goes to `out` first and
// the pooling pass adds to it in place.  x and out must not overlap.
int block_end(bq_ctx* c, const char* res_name, const char* pool_name, const void* x, const void* y, void* out, int n,
              int Hi, int ci, int co, int cout, hipStream_t s, Tap* tap, int* tapped) {
    const int Ho = (Hi + 1) / 2;
    const double es = (double)esize(c);
    auto it = c->layers.find(res_name);
    if (it == c->layers.end()) return fail(c, BQ_ERR_WEIGHTS, std::string("layer not loaded: ") + res_name);
    const GemmLayer& L = it->second;
    const bool want_res = tap && tap->want && strcmp(tap->want, res_name) == 0;
    static const bool no_fuse = bq_exp_env("BQ_NO_RESPOOL") != nullptr;
    *tapped = 0;
    // measured per batch of 256 (one stream): block 2 0.62 -> 0.46 ms, block 3 0.34 -> 0.27 ms; block 4 (K = 256, six
    // 128-channel workgroups per pixel tile) 0.26 -> 0.34 ms and block 13 0.19 -> 0.20 ms stay on the two-kernel path
    static const bool fuse_all = bq_exp_env("BQ_RESPOOL_ALL") != nullptr;
    // round 4: the shortcuts of blocks 3, 4 and 13 (K = 128 / 256 / 736) as the tiled GEMM with the pooling pass as its store pass
    // (kernels_split.hip, EPI_POOL): the shortcut tensor never goes to HBM, one launch instead of two
    static const bool no_poolgemm = bq_exp_env("BQ_NO_POOLGEMM") != nullptr;
#ifndef POOLGEMM_MINK
#define POOLGEMM_MINK 128     // block 3 (K = 128) too: 0.273 -> 0.242 ms against kernels_respool.hip; block 2 (K = 64) lives in the fused tail
#endif
    if (is16(c->cfg.dtype) && !want_res && !no_poolgemm && L.kpad >= POOLGEMM_MINK && L.nfp % 4 == 0) {
        const double px = (double)n * Ho * Ho;
        GemmParams p{};
        p.in = x; p.wp = L.wp; p.scale = L.scale; p.bias = L.bias; p.residual = y; p.out = out;
        p.M = n * Ho * Ho; p.K = L.kpad; p.KBtot = L.kpad / 16; p.kb0 = 0; p.k_off = 0;
        p.NFp = L.nfp; p.Nstore = co; p.ldo = co; p.ldi = ci;
        p.H = Ho; p.W = Ho; p.Hi = Hi; p.Wi = Hi; p.relu = 0;
        ProfScope ps(c, s, std::string("respool_") + std::to_string(Hi) + "_c" + std::to_string(cout),
                     2.0 * px * L.cin * L.cout + 9.0 * px * co,
                     es * ((double)n * Hi * Hi * co + px * co + px * ci) + es * (double)L.cin * L.cout);
        const int e = launch_gemm_tile(c->cfg.dtype, p, true, s, 2);
        if (e) return fail(c, BQ_ERR_HIP, std::string("launch(gemm_tile pool) ") + res_name + ": " + hipGetErrorString((hipError_t)e));
        return BQ_OK;
    }
    if (is16(c->cfg.dtype) && L.wp32 && !want_res && !no_fuse && (L.kpad <= 128 || fuse_all)) {
        const double px = (double)n * Ho * Ho;
        ProfScope ps(c, s, std::string("respool_") + std::to_string(Hi) + "_c" + std::to_string(cout),
                     2.0 * px * L.cin * L.cout + 9.0 * px * co,
                     es * ((double)n * Hi * Hi * co + px * co + px * ci) + es * (double)L.cin * L.cout);
        const int e = launch_respool(c->cfg.dtype, x, L.wp32, L.scale, L.bias, y, out, n, Hi, Hi, L.kpad, ci, co, L.nfp, s);
        if (e) return fail(c, BQ_ERR_HIP, std::string("launch(respool) ") + res_name + ": " + hipGetErrorString((hipError_t)e));
        return BQ_OK;
    }
    {
        const int r = run_conv(c, {res_name, PROD_S2, x, out, nullptr, nullptr, n, Ho, Ho, Hi, Hi, ci, co, 0}, s);
        if (r != BQ_OK) return r;
        const int t = tap_nhwc(c, tap, res_name, out, n, Ho, Ho, cout, co, s);
        if (t) { *tapped = 1; return t < 0 ? t : BQ_OK; }
    }
    const double px = (double)n * Ho * Ho * co;
    ProfScope ps(c, s, pool_name, 9.0 * px, es * ((double)n * Hi * Hi * co + 2.0 * px));
    if (launch_pool_add(y, out, out, n, Hi, Hi, co, c->cfg.dtype, s)) return fail(c, BQ_ERR_HIP, "pool_add launch failed");
    return BQ_OK;
}

// Stem + entry flow (blocks 1-4) of n tiles.  A/B/C/R are scratch for n tiles; the block-4 output
// (19x19x736 per tile) goes to out4.  Returns 1 if a debug tap matched (caller stops).
// u8 != nullptr: the uint8 tiles themselves -- staging, block1_conv1 and block1_conv2 run as ONE kernel (kernels_front.hip;
// `in_nchw` is not read) when its weights are loaded
int entry_flow(bq_ctx* c, const void* in_nchw, int n, void* out4, void* A, void* B, void* C, void* R,
               hipStream_t s, Tap* tap, const uint8_t* u8 = nullptr) {
    const int dt = c->cfg.dtype;
    const double es = (double)esize(c);
    if (u8) {
        auto l2 = c->layers.find("block1_conv2");
        if (!c->front_ws16 || !c->front_wc16 || l2 == c->layers.end() || !is16(dt))
            return fail(c, BQ_ERR_ARG, "the fused front kernel needs a 16-bit context with its weights loaded");
        if (tap && tap->want && (strcmp(tap->want, "staged") == 0 || strcmp(tap->want, "block1_conv1") == 0))
            return fail(c, BQ_ERR_ARG, "the fused front kernel does not materialise the staged tile or block1_conv1");
        {
            ProfScope ps(c, s, "stage_stats", 2.0 * n * kStaged, (double)n * kStaged);
            if (launch_stage_stats(u8, n, 299, c->d_stage_stats, s)) return fail(c, BQ_ERR_HIP, "stage stats launch failed");
        }
        {
            const double p1 = (double)n * 149 * 149, p2 = (double)n * 147 * 147;
            ProfScope ps(c, s, "front_stage_stem_conv2", 2.0 * p1 * 27 * 32 + 2.0 * p2 * 288 * 64 + 4.0 * n * kStaged,
                         (double)n * kStaged + es * p2 * 64);
            const int e = launch_front(dt, u8, reinterpret_cast<const unsigned long long*>(c->d_stage_stats), c->front_ws16, c->stem_s,
                                       c->stem_b, c->front_wc16, l2->second.scale, l2->second.bias, B, n, c->num_cus, s);
            if (e) return fail(c, BQ_ERR_HIP, std::string("launch(front): ") + hipGetErrorString((hipError_t)e));
        }
    } else {
    if (tap && tap->want && strcmp(tap->want, "staged") == 0) {
        if ((size_t)n * kStaged > tap->out_elems) return fail(c, BQ_ERR_ARG, "debug output too small");
        if (launch_nchw_to_f32_nhwc(in_nchw, n, 3, 299 * 299, tap->out, dt, s))
            return fail(c, BQ_ERR_HIP, "debug copy failed");
        tap->written = (int64_t)n * kStaged;
        return BQ_OK;
    }
    {   // block1_conv1 + bn + relu  (vector ALU)
        const double px = (double)n * 149 * 149;
        ProfScope ps(c, s, "stem_conv1_3x3s2", 2.0 * px * 27 * 32, es * ((double)n * kStaged + px * 32));
        if (launch_stem1(in_nchw, n, c->stem_w, c->stem_s, c->stem_b, A, dt, s))
            return fail(c, BQ_ERR_HIP, "stem1 launch failed");
    }
    TAP("block1_conv1", A, 149, 149, 32, 32);
    RUN(run_conv(c, {"block1_conv2", PROD_IM2COL, A, B, nullptr, nullptr, n, 147, 147, 149, 149, 32, 64, 1}, s));
    }
    TAP("block1_conv2", B, 147, 147, 64, 64);

    // entry flow: blocks 2-4.  The block's input lives in `cur`, its output goes to `nxt` (block 4: out4)
    struct Entry { int block, cin, cout, Hi; };
    const Entry entry[3] = {{2, 64, 128, 147}, {3, 128, 256, 74}, {4, 256, 728, 37}};
    void* cur = B; void* nxt = R;
    for (const Entry& e : entry) {
        const int Ho = (e.Hi + 1) / 2;
        const int ci = pad16(e.cin), co = pad16(e.cout);
        char nm[64], rn[64], tn[64];
        snprintf(nm, sizeof nm, "block%d_sepconv1", e.block);
        RUN(run_conv(c, {nm, e.block == 2 ? PROD_DW : PROD_DW_RELU, cur, A, nullptr, nullptr, n, e.Hi, e.Hi,
                         e.Hi, e.Hi, ci, co, 1}, s));
        TAP(nm, A, e.Hi, e.Hi, e.cout, co);
        snprintf(nm, sizeof nm, "block%d_sepconv2", e.block);
        snprintf(rn, sizeof rn, "block%d_res", e.block);
        void* dst = e.block == 4 ? out4 : nxt;
        {   // round 4: the block's tail in one kernel (kernels_stream.hip) -- sepconv2 + BN, max-pool, shortcut conv + BN, add --
            // unless the tensors it no longer writes were asked for
            static const bool no_tail = bq_exp_env("BQ_NO_TAIL") != nullptr;
            auto l2 = c->layers.find(nm), lr = c->layers.find(rn);
            const bool want_mid = tap && tap->want && (strcmp(tap->want, nm) == 0 || strcmp(tap->want, rn) == 0);
            if (!no_tail && !want_mid && l2 != c->layers.end() && lr != c->layers.end() && l2->second.wp16 && lr->second.wp16 &&
                e.cout == co && e.cin == ci && tail_supported(dt, co, co, ci, n, e.Hi, e.Hi)) {
                const double M = (double)n * e.Hi * e.Hi, Mo = (double)n * Ho * Ho;
                char cls[64];
                snprintf(cls, sizeof cls, "blocktail_%d_c%d", e.Hi, e.cout);
                {
                    ProfScope ps(c, s, cls, 2.0 * M * co * co + 18.0 * M * co + 2.0 * Mo * ci * co + 9.0 * Mo * co,
                                 es * (M * co + Mo * ci + Mo * co) + es * ((double)co * co + (double)ci * co));
                    const int er = launch_block_tail(dt, co, co, ci, A, l2->second.wp16, l2->second.dw, l2->second.scale,
                                                     l2->second.bias, cur, lr->second.wp16, lr->second.scale, lr->second.bias, dst,
                                                     n, e.Hi, e.Hi, c->num_cus, s);
                    if (er) return fail(c, BQ_ERR_HIP, std::string("launch(block tail) ") + nm + ": " + hipGetErrorString((hipError_t)er));
                }
                snprintf(nm, sizeof nm, "block%d_out", e.block);
                TAP(nm, dst, Ho, Ho, e.cout, co);
                void* t = cur; cur = nxt; nxt = t;
                continue;
            }
        }
        RUN(run_conv(c, {nm, PROD_DW, A, C, nullptr, nullptr, n, e.Hi, e.Hi, e.Hi, e.Hi, co, co, 0}, s));
        TAP(nm, C, e.Hi, e.Hi, e.cout, co);
        snprintf(tn, sizeof tn, "maxpool_add_%d_c%d", e.Hi, e.cout);
        int tapped = 0;
        RUN(block_end(c, rn, tn, cur, C, dst, n, e.Hi, ci, co, e.cout, s, tap, &tapped));
        if (tapped) return BQ_OK;
        snprintf(nm, sizeof nm, "block%d_out", e.block);
        TAP(nm, dst, Ho, Ho, e.cout, co);
        void* t = cur; cur = nxt; nxt = t;
    }
    return BQ_OK;
}

// part: bit 0 = stem + entry flow (blocks 1-4; its output stays in the workspace), bit 1 = middle + exit flow (reads it there):
// (experiments build: bq_mc_infer_part lets a scheduler run the two halves of two batches against each other)
int backbone_impl(bq_ctx* c, const void* in_nchw, int n, float* feat, unsigned char* ws, hipStream_t s,
                  Tap* tap, const uint8_t* u8 = nullptr, int part = 3) {
    const WsLayout L = ws_layout(c, n, 1);
    void* A = ws + L.a; void* B = ws + L.b; void* C = ws + L.c; void* R = ws + L.r;
    const int dt = c->cfg.dtype;
    const double es = (double)esize(c);
    // Entry flow in sub-batches: its activations are the big ones (up to 5.5 MB per tile and layer);
    // with a small sub-batch every intermediate buffer is re-used at the same addresses and stays in
    // the 256 MiB Infinity Cache instead of round-tripping through HBM.  The block-4 outputs of all
    // sub-batches are gathered in the upper half of buffer B (the sub-batches only touch the front).
    static const int env_sub = bq_exp_env("BQ_SUB") ? atoi(bq_exp_env("BQ_SUB")) : 0;
    int sub = (tap && tap->want) ? n : env_sub;
    if (sub <= 0 || sub > n / 2 || part != 3) sub = n;
    const size_t tile4 = (size_t)361 * 736 * esize(c);
    // Block 4 reads its input from B (blocks 2-4 alternate between B and R) and must not write over it: in one piece
    // its output goes to R, and B is the scratch buffer S of the middle and exit flow; in sub-batches it goes to the
    // upper half of B, which no sub-batch touches.
    unsigned char* X4 = sub == n ? (unsigned char*)R : (unsigned char*)B + (size_t)(n / 2) * kMaxAct * esize(c);
    void* S = sub == n ? B : R;
    if (sub == n) {
        if (part & 1) RUN(entry_flow(c, in_nchw, n, X4, A, B, C, R, s, tap, u8));
        if (tap && tap->written >= 0) return BQ_OK;
        if (!(part & 2)) return BQ_OK;
    } else {
        for (int i0 = 0; i0 < n; i0 += sub) {
            const int ns = n - i0 < sub ? n - i0 : sub;
            const unsigned char* in_i = (const unsigned char*)in_nchw + (size_t)i0 * kStaged * esize(c);
            RUN(entry_flow(c, in_i, ns, X4 + (size_t)i0 * tile4, A, B, C, R, s, nullptr, u8 ? u8 + (size_t)i0 * kStaged : nullptr));
        }
    }
    // middle flow: blocks 5-12 at 19x19x728 (stride 736)
    void* X = X4; void* Y = A;
    for (int block = 5; block <= 12; ++block) {
        char nm[64];
        snprintf(nm, sizeof nm, "block%d_sepconv1", block);
        RUN(run_conv(c, {nm, PROD_DW_RELU, X, Y, nullptr, nullptr, n, 19, 19, 19, 19, 736, 736, 1, S}, s));
        TAP(nm, Y, 19, 19, 728, 736);
        snprintf(nm, sizeof nm, "block%d_sepconv2", block);
        RUN(run_conv(c, {nm, PROD_DW, Y, C, nullptr, nullptr, n, 19, 19, 19, 19, 736, 736, 1, S}, s));
        TAP(nm, C, 19, 19, 728, 736);
        snprintf(nm, sizeof nm, "block%d_sepconv3", block);
        RUN(run_conv(c, {nm, PROD_DW, C, Y, X, nullptr, n, 19, 19, 19, 19, 736, 736, 0, S}, s));
        void* t = X; X = Y; Y = t;
        snprintf(nm, sizeof nm, "block%d_out", block);
        TAP(nm, X, 19, 19, 728, 736);
    }
    // exit flow
    RUN(run_conv(c, {"block13_sepconv1", PROD_DW_RELU, X, Y, nullptr, nullptr, n, 19, 19, 19, 19, 736, 736, 1}, s));
    TAP("block13_sepconv1", Y, 19, 19, 728, 736);
    RUN(run_conv(c, {"block13_sepconv2", PROD_DW, Y, C, nullptr, nullptr, n, 19, 19, 19, 19, 736, 1024, 0}, s));
    TAP("block13_sepconv2", C, 19, 19, 1024, 1024);
    {   // the block's output goes to S (X is its input); X is scratch from here on
        int tapped = 0;
        RUN(block_end(c, "block13_res", "maxpool_add_19_c1024", X, C, S, n, 19, 736, 1024, 1024, s, tap, &tapped));
        if (tapped) return BQ_OK;
    }
    TAP("block13_out", S, 10, 10, 1024, 1024);
    RUN(run_conv(c, {"block14_sepconv1", PROD_DW, S, Y, nullptr, C, n, 10, 10, 10, 10, 1024, 1536, 1, X}, s));
    TAP("block14_sepconv1", Y, 10, 10, 1536, 1536);
    {   // round 4: the global average pool is the epilogue of block14_sepconv2's GEMM (one workgroup owns an image's 100
        // pixels) unless the convolution's own output was asked for
        const bool want14 = tap && tap->want && strcmp(tap->want, "block14_sepconv2") == 0;
        static const bool no_gapfuse = bq_exp_env("BQ_NO_GAPFUSE") != nullptr;
        bool gap_done = false;
        ConvArgs a14{"block14_sepconv2", PROD_DW, Y, C, nullptr, S, n, 10, 10, 10, 10, 1536, 2048, 1, X};
        if (!want14 && !no_gapfuse) { a14.gap_out = feat; a14.gap_done = &gap_done; }
        RUN(run_conv(c, a14, s));
        if (!gap_done) {
            TAP("block14_sepconv2", C, 10, 10, 2048, 2048);
            ProfScope ps(c, s, "global_avg_pool", (double)n * 100 * 2048, es * (double)n * 100 * 2048 + 4.0 * n * 2048);
            if (launch_gap(C, n, 100, 2048, 2048, feat, c->feat_mul, dt, s)) return fail(c, BQ_ERR_HIP, "gap launch failed");
        }
    }
    if (tap && tap->want) return fail(c, BQ_ERR_ARG, std::string("unknown activation: ") + tap->want);
    return BQ_OK;
}

int head_impl(bq_ctx* c, const float* feat, int n, int64_t tile0, int mc_n, int pass0, uint64_t seed,
              int init, int finalize, float* state, float* mean2, float* std2, unsigned char* ws,
              hipStream_t s) {
    const WsLayout L = ws_layout(c, n, mc_n);
    float* h0 = (float*)(ws + L.h0);
    float* h1 = (float*)(ws + L.h1);
    const double rate = (double)c->cfg.dropout;
    double t = floor(rate * 4294967296.0);
    if (t < 0) t = 0; if (t > 4294967295.0) t = 4294967295.0;
    const unsigned thresh = (unsigned)t;
    const float dscale = (float)(1.0 / (1.0 - rate));
    const int rows = n * mc_n;
    for (int layer = 0; layer < 2; ++layer) {
        const HeadLayer& G = c->head[layer];
        if (!G.wh) return fail(c, BQ_ERR_WEIGHTS, "head weights not loaded");
        const int K = G.k;                       // 2048 / 1024
        // three f16 MFMAs per fp32 product (kernels_head.hip): 6 x the nominal FLOPs of the layer
        ProfScope ps(c, s, layer == 0 ? "mc_head_dense0" : "mc_head_dense1", 2.0 * rows * (double)K * 1024,
                     4.0 * ((layer == 0 ? (double)n : (double)rows) * K + (double)rows * 1024 + (double)K * 1024));
        const int e = launch_head_dense(layer == 0 ? feat : h0, G.wh, G.wl, G.bias, layer == 0 ? h0 : h1, rows, K, mc_n, pass0,
                                        layer == 0 ? 1 : 0, layer, (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32), thresh,
                                        dscale, tile0, c->d_tile0, c->d_tile_idx, s);
        if (e) return fail(c, BQ_ERR_HIP, std::string("head dense launch: ") + hipGetErrorString((hipError_t)e));
    }
    {
        ProfScope ps(c, s, "mc_head_softmax_welford", 2.0 * rows * 1024 * 2, 4.0 * rows * 1024);
        if (launch_head_final(h1, n, mc_n, pass0, tile0, c->d_tile0, c->d_tile_idx, (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32),
                              thresh, dscale, c->logits_w, c->logits_b, init, finalize, state, mean2, std2, s))
            return fail(c, BQ_ERR_HIP, "head_final launch failed");
    }
    return BQ_OK;
}

const float* entry_f32(bq_ctx* c, const std::string& name) {
    auto it = c->entries.find(name);
    return it == c->entries.end() ? nullptr : reinterpret_cast<const float*>(it->second.p);
}

int register_gemm_layer(bq_ctx* c, const std::string& name, int cin, int cout, int kpad, bool has_dw,
                        int vec, int elt) {
    GemmLayer L;
    L.name = name; L.cin = cin; L.cout = cout; L.kpad = kpad;
    auto w = c->entries.find(name + "/wp");
    if (w == c->entries.end()) return fail(c, BQ_ERR_WEIGHTS, "missing " + name + "/wp");
    const size_t per_nf = (size_t)(kpad / (2 * vec)) * 64 * vec * elt;
    if (per_nf == 0 || w->second.n % per_nf) return fail(c, BQ_ERR_WEIGHTS, "bad size for " + name + "/wp");
    L.nfp = (int)(w->second.n / per_nf);
    if (L.nfp * 32 < cout) return fail(c, BQ_ERR_WEIGHTS, "too few output fragments in " + name);
    L.wp = w->second.p;
    auto w16 = c->entries.find(name + "/wp16");
    if (w16 != c->entries.end()) {
        if (kpad % 32 || w16->second.n != (size_t)(kpad / 32) * ((size_t)L.nfp * 2) * 1024)
            return fail(c, BQ_ERR_WEIGHTS, "bad size for " + name + "/wp16");
        L.wp16 = w16->second.p;
    }
    auto w32 = c->entries.find(name + "/wp32");
    if (w32 != c->entries.end()) {
        if (w32->second.n != (size_t)(kpad / 16) * (size_t)L.nfp * 1024)
            return fail(c, BQ_ERR_WEIGHTS, "bad size for " + name + "/wp32");
        L.wp32 = w32->second.p;
    }
    L.scale = entry_f32(c, name + "/scale");
    L.bias = entry_f32(c, name + "/bias");
    if (!L.scale || !L.bias) return fail(c, BQ_ERR_WEIGHTS, "missing scale/bias for " + name);
    if (c->entries[name + "/scale"].n < (size_t)cout * 4 || c->entries[name + "/bias"].n < (size_t)cout * 4)
        return fail(c, BQ_ERR_WEIGHTS, "scale/bias of " + name + " shorter than its output channels");
    if (has_dw) {
        L.dw = entry_f32(c, name + "/dw");
        if (!L.dw) return fail(c, BQ_ERR_WEIGHTS, "missing " + name + "/dw");
        if (c->entries[name + "/dw"].n < (size_t)9 * kpad * 4)
            return fail(c, BQ_ERR_WEIGHTS, "depthwise taps of " + name + " shorter than 9 x its padded input channels");
    }
    c->layers[name] = L;
    return BQ_OK;
}

}  // namespace

// =================================================================== C ABI
extern "C" {

bq_ctx* bq_create(int device_id, const bq_config* cfg) {
    if (!cfg) { g_create_error = "cfg is null"; return nullptr; }
    if (cfg->tile_px != 299 || cfg->n_classes != 2 ||
        (cfg->dtype != BQ_DTYPE_F32 && cfg->dtype != BQ_DTYPE_BF16 && cfg->dtype != BQ_DTYPE_F16) ||
        !(cfg->dropout >= 0.f) || !(cfg->dropout < 1.f)) {
        g_create_error = "unsupported config (need tile_px=299, n_classes=2, dtype f32|bf16|f16, 0<=dropout<1)";
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) {
        g_create_error = "no such HIP device";
        return nullptr;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { g_create_error = "hipGetDeviceProperties failed"; return nullptr; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("libbiscuit_hip is built for gfx950 only, device is ") + prop.gcnArchName;
        return nullptr;
    }
    bq_ctx* c = new (std::nothrow) bq_ctx();
    if (!c) { g_create_error = "out of host memory"; return nullptr; }
    DeviceGuard guard(device_id);            // allocations below go to the context's device; the caller's stays current
    if (!guard.ok) { g_create_error = "hipSetDevice failed"; delete c; return nullptr; }
    c->cfg = *cfg;
    c->device = device_id;
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* e = bq_exp_env("BQ_NUM_CUS")) { const int v = atoi(e); if (v > 0) c->num_cus = v; }   // persistent-grid sizing (experiments)
    {   // tables of the Reinhard normaliser (oracle/stain.py states the same arithmetic):
        // [0,256)   sRGB -> linear, float64 evaluation rounded to float32
        // [256,511) linear -> 8-bit sRGB as 255 switching points: entry v-1 is the smallest float32 c for which
        //           clip(trunc(255 * clip(gamma(c), 0, 1)), 0, 255) >= v, gamma(c) = c > 0.0031308 ?
        //           1.055f * float(pow(double(c), 1/2.4)) - 0.055f : 12.92f * c, found by bisection on the
        //           float bit pattern (the function is monotone)
        float lut[512];
        for (int v = 0; v < 256; ++v) {
            const double x = (double)v / 255.0;
            lut[v] = (float)(x > 0.04045 ? std::pow((x + 0.055) / 1.055, 2.4) : x / 12.92);
        }
        auto level = [](float cf) {
            float g;
            if (cf > 0.0031308f) { const float p = (float)std::pow((double)cf, 1.0 / 2.4); g = 1.055f * p - 0.055f; }
            else g = cf * 12.92f;
            g = g < 0.f ? 0.f : (g > 1.f ? 1.f : g);
            const float t = truncf(g * 255.0f);
            return (int)(t < 0.f ? 0.f : (t > 255.f ? 255.f : t));
        };
        for (int v = 1; v <= 255; ++v) {
            uint32_t lo = 0, hi = 0x3F800000u;               // bit patterns of 0.0f and 1.0f; level(1.0f) = 255
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo) / 2;
                float f;
                memcpy(&f, &mid, 4);
                if (level(f) >= v) hi = mid; else lo = mid + 1;
            }
            memcpy(&lut[256 + v - 1], &lo, 4);
        }
        lut[511] = 0.f;
        if (hipMalloc(&c->d_srgb_lut, sizeof lut) != hipSuccess ||
            hipMemcpy(c->d_srgb_lut, lut, sizeof lut, hipMemcpyHostToDevice) != hipSuccess) {
            g_create_error = "cannot allocate the sRGB tables";
            delete c;
            return nullptr;
        }
    }
    if (hipMalloc(&c->d_stage_stats, (size_t)(cfg->max_batch > 0 ? cfg->max_batch : 1) * 16) != hipSuccess) {
        g_create_error = "cannot allocate the staging statistics";
        (void)hipFree(c->d_srgb_lut);
        delete c;
        return nullptr;
    }
    return c;
}

void bq_destroy(bq_ctx* c) {
    if (!c) return;
    if (c->d_blob) (void)hipFree(c->d_blob);
    if (c->d_srgb_lut) (void)hipFree(c->d_srgb_lut);
    if (c->d_stage_stats) (void)hipFree(c->d_stage_stats);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    delete c;
}

const char* bq_last_error(bq_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

size_t bq_workspace_bytes(bq_ctx* c, int batch, int mc_n) {
    if (!c || batch <= 0) return 0;
    return ws_layout(c, batch, mc_n).total;
}

int bq_load_weights(bq_ctx* c, const void* host_blob, size_t nbytes) {
    if (!c || !host_blob || nbytes < 16) return fail(c, BQ_ERR_ARG, "bad weight blob");
    const unsigned char* hb = (const unsigned char*)host_blob;
    uint32_t ver, cnt, dt;
    if (memcmp(hb, "BQW1", 4) != 0) return fail(c, BQ_ERR_WEIGHTS, "bad magic (want BQW1)");
    memcpy(&ver, hb + 4, 4); memcpy(&cnt, hb + 8, 4); memcpy(&dt, hb + 12, 4);
    if (ver != 1 || (size_t)16 + (size_t)cnt * 64 > nbytes) return fail(c, BQ_ERR_WEIGHTS, "bad header");
    if ((int)dt != c->cfg.dtype) return fail(c, BQ_ERR_WEIGHTS, "blob dtype does not match context dtype");
    // validate the directory before anything is allocated (wrap-free bounds: off and len are untrusted 64-bit values)
    for (uint32_t i = 0; i < cnt; ++i) {
        const unsigned char* e = hb + 16 + (size_t)i * 64;
        char name[49]; memcpy(name, e, 48); name[48] = 0;
        uint64_t off, len; memcpy(&off, e + 48, 8); memcpy(&len, e + 56, 8);
        if (off > nbytes || len > nbytes - off || (off & 255)) return fail(c, BQ_ERR_WEIGHTS, std::string("bad entry ") + name);
    }
    DeviceGuard guard(c->device);            // restores the caller's current device on every exit path
    if (!guard.ok) return fail(c, BQ_ERR_HIP, "hipSetDevice failed");
    if (c->d_blob) { (void)hipFree(c->d_blob); c->d_blob = nullptr; }
    c->entries.clear(); c->layers.clear(); c->loaded = false;
    c->head[0] = c->head[1] = HeadLayer{};
    HIPCHK(c, hipMalloc((void**)&c->d_blob, nbytes));
    HIPCHK(c, hipMemcpy(c->d_blob, hb, nbytes, hipMemcpyHostToDevice));
    c->blob_bytes = nbytes;
    for (uint32_t i = 0; i < cnt; ++i) {
        const unsigned char* e = hb + 16 + (size_t)i * 64;
        char name[49]; memcpy(name, e, 48); name[48] = 0;
        uint64_t off, len; memcpy(&off, e + 48, 8); memcpy(&len, e + 56, 8);
        c->entries[name] = Blob{c->d_blob + off, (size_t)len};
    }
    const int vec = is16(c->cfg.dtype) ? 8 : 4;
    const int elt = is16(c->cfg.dtype) ? 2 : 4;
    c->stem_w = entry_f32(c, "block1_conv1/w");
    c->stem_s = entry_f32(c, "block1_conv1/scale");
    c->stem_b = entry_f32(c, "block1_conv1/bias");
    c->logits_w = entry_f32(c, "logits/w");
    c->logits_b = entry_f32(c, "logits/bias");
    if (!c->stem_w || !c->stem_s || !c->stem_b || !c->logits_w || !c->logits_b)
        return fail(c, BQ_ERR_WEIGHTS, "missing stem/logits tensors");
    c->feat_mul = 1.f;
    {
        auto fm = c->entries.find("act/feat_mul");
        if (fm != c->entries.end()) {
            if (fm->second.n < 4) return fail(c, BQ_ERR_WEIGHTS, "bad size for act/feat_mul");
            memcpy(&c->feat_mul, hb + (fm->second.p - c->d_blob), 4);
            if (!(c->feat_mul > 0.f) || !std::isfinite(c->feat_mul)) return fail(c, BQ_ERR_WEIGHTS, "act/feat_mul must be a positive finite number");
        }
    }
    c->front_ws16 = c->front_wc16 = nullptr;
    {
        auto a = c->entries.find("block1_conv1/w16"), b = c->entries.find("block1_conv2/wp16");
        if (a != c->entries.end() && b != c->entries.end()) {
            if (a->second.n != 2 * 2 * 1024 || b->second.n != 9 * 4 * 1024)
                return fail(c, BQ_ERR_WEIGHTS, "bad size for block1_conv1/w16 or block1_conv2/wp16");
            c->front_ws16 = a->second.p;
            c->front_wc16 = b->second.p;
        }
    }
    RUN(register_gemm_layer(c, "block1_conv2", 32, 64, 288, false, vec, elt));
    const int res[4][3] = {{2, 64, 128}, {3, 128, 256}, {4, 256, 728}, {13, 728, 1024}};
    for (auto& r : res)
        RUN(register_gemm_layer(c, "block" + std::to_string(r[0]) + "_res", r[1], r[2], pad16(r[1]), false, vec, elt));
    struct S { int block, idx, cin, cout; };
    std::vector<S> seps = {{2, 1, 64, 128}, {2, 2, 128, 128}, {3, 1, 128, 256}, {3, 2, 256, 256},
                           {4, 1, 256, 728}, {4, 2, 728, 728}};
    for (int b = 5; b <= 12; ++b) for (int i = 1; i <= 3; ++i) seps.push_back({b, i, 728, 728});
    seps.push_back({13, 1, 728, 728}); seps.push_back({13, 2, 728, 1024});
    seps.push_back({14, 1, 1024, 1536}); seps.push_back({14, 2, 1536, 2048});
    for (auto& sp : seps)
        RUN(register_gemm_layer(c, "block" + std::to_string(sp.block) + "_sepconv" + std::to_string(sp.idx),
                                sp.cin, sp.cout, pad16(sp.cin), true, vec, elt));
    for (int layer = 0; layer < 2; ++layer) {
        const std::string name = layer == 0 ? "hidden_0" : "hidden_1";
        const int K = layer == 0 ? 2048 : 1024;
        auto wh = c->entries.find(name + "/wph"), wl = c->entries.find(name + "/wpl"), bi = c->entries.find(name + "/bias");
        const size_t want = (size_t)32 * (K / 16) * 64 * 16;       // [1024 / 32][K / 16][64] x 16 B
        if (wh == c->entries.end() || wl == c->entries.end() || bi == c->entries.end() || wh->second.n != want ||
            wl->second.n != want || bi->second.n < 1024 * 4)
            return fail(c, BQ_ERR_WEIGHTS, "missing or malformed head tensors of " + name);
        c->head[layer] = HeadLayer{wh->second.p, wl->second.p, reinterpret_cast<const float*>(bi->second.p), K};
    }
    c->loaded = true;
    return BQ_OK;
}

int bq_stage(bq_ctx* c, const uint8_t* d_tiles, int n, void* d_out, bq_stream_t stream) {
    if (!c || !d_tiles || !d_out || n < 0 || n > c->cfg.max_batch) return fail(c, BQ_ERR_ARG, "bq_stage: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(c, s, "stage_u8_standardize", 4.0 * n * kStaged, (double)n * kStaged * (1.0 + esize(c)));
    static const bool one_kernel = bq_exp_env("BQ_STAGE_1K") != nullptr;
    if (launch_stage_u8(d_tiles, n, 299, d_out, c->cfg.dtype, one_kernel ? nullptr : c->d_stage_stats, s))
        return fail(c, BQ_ERR_HIP, "stage launch failed");
    return BQ_OK;
}

namespace {
// float32 colour constants of the Reinhard normaliser: XYZ<-RGB, RGB<-XYZ (its float64 inverse rounded),
// D65 white (oracle/stain.py: constants())
const float kReinhardConsts[21] = {
    0.412452996f, 0.357580006f, 0.180423006f, 0.212670997f, 0.715160012f, 0.0721689984f, 0.0193339996f,
    0.119193003f, 0.950227022f,
    3.24048138f, -1.53715158f, -0.498536319f, -0.969254971f, 1.87599003f, 0.0415559262f, 0.0556466393f,
    -0.204041332f, 1.05731106f,
    0.950469971f, 1.0f, 1.08882999f};
}  // namespace

int bq_stain_reinhard_fast(bq_ctx* c, const uint8_t* d_tiles, int n, const float* target_means3,
                           const float* target_stds3, uint8_t* d_out, bq_stream_t stream) {
    if (!c || !d_tiles || !d_out || !target_means3 || !target_stds3 || n < 0)
        return fail(c, BQ_ERR_ARG, "bq_stain_reinhard_fast: bad argument");
    for (int i = 0; i < 3; ++i)
        if (!(target_stds3[i] == target_stds3[i]) || !(target_means3[i] == target_means3[i]))
            return fail(c, BQ_ERR_ARG, "bq_stain_reinhard_fast: NaN in the target statistics");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(c, s, "stain_reinhard_fast", 300.0 * n * 299 * 299, 3.0 * n * kStaged);
    if (launch_reinhard(d_tiles, n, 299, c->d_srgb_lut, kReinhardConsts, target_means3, target_stds3, d_out, nullptr, s))
        return fail(c, BQ_ERR_HIP, "reinhard launch failed");
    return BQ_OK;
}

int bq_stain_lab_stats(bq_ctx* c, const uint8_t* d_tiles, int n, float* d_stats6, bq_stream_t stream) {
    if (!c || !d_tiles || !d_stats6 || n < 0) return fail(c, BQ_ERR_ARG, "bq_stain_lab_stats: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (launch_reinhard(d_tiles, n, 299, c->d_srgb_lut, kReinhardConsts, nullptr, nullptr, nullptr, d_stats6, s))
        return fail(c, BQ_ERR_HIP, "lab stats launch failed");
    return BQ_OK;
}

int bq_png_unfilter(bq_ctx* c, const uint8_t* d_rows, int n, int px, uint8_t* d_out, bq_stream_t stream) {
    if (!c || !d_rows || !d_out || n < 0 || px <= 0) return fail(c, BQ_ERR_ARG, "bq_png_unfilter: bad argument");
    if (launch_png_unfilter(d_rows, n, px, d_out, (hipStream_t)stream)) return fail(c, BQ_ERR_HIP, "png unfilter launch failed");
    return BQ_OK;
}

size_t bq_png_inflate_scratch_bytes(int n) { return inflate_scratch_bytes(n); }

int bq_png_inflate(bq_ctx* c, const uint8_t* d_z, const uint32_t* d_off, const uint32_t* d_len, int n, int px, uint8_t* d_rows,
                   size_t rows_stride, void* d_scratch, size_t scratch_bytes, int32_t* d_status, bq_stream_t stream) {
    if (!c || !d_z || !d_off || !d_len || !d_rows || !d_scratch || !d_status || n < 0 || px <= 0 || px > 4096)
        return fail(c, BQ_ERR_ARG, "bq_png_inflate: bad argument");
    const size_t row_bytes = (size_t)px * (3 * (size_t)px + 1);
    if (rows_stride < row_bytes + 4 || (rows_stride & 3) || rows_stride > 0xffffffffull) return fail(c, BQ_ERR_ARG, "bq_png_inflate: rows_stride must be a multiple of 4, >= px (1 + 3 px) + 4");
    if (scratch_bytes < inflate_scratch_bytes(n)) return fail(c, BQ_ERR_WORKSPACE, "bq_png_inflate: scratch too small");
    ProfScope ps(c, (hipStream_t)stream, "png_inflate", 0.0, (double)n * row_bytes * 2.0);
    const int e = launch_inflate(d_z, d_off, d_len, n, d_rows, (unsigned)row_bytes, (unsigned)rows_stride, d_scratch, d_status, (hipStream_t)stream,
                                 c->inflate_variant, (unsigned)(1 + 3 * px));
    if (e) return fail(c, BQ_ERR_HIP, std::string("png inflate launch: ") + hipGetErrorString((hipError_t)e));
    return BQ_OK;
}

int bq_png_unfilter_strided(bq_ctx* c, const uint8_t* d_rows, size_t rows_stride, int n, int px, uint8_t* d_out, bq_stream_t stream) {
    if (!c || !d_rows || !d_out || n < 0 || px <= 0) return fail(c, BQ_ERR_ARG, "bq_png_unfilter_strided: bad argument");
    if (launch_png_unfilter(d_rows, n, px, d_out, (hipStream_t)stream, rows_stride)) return fail(c, BQ_ERR_HIP, "png unfilter launch failed");
    return BQ_OK;
}

int bq_stream_create_masked(bq_ctx* c, const uint32_t* cu_mask, int mask_words, bq_stream_t* out) {
    if (!c || !cu_mask || mask_words <= 0 || !out) return fail(c, BQ_ERR_ARG, "bq_stream_create_masked: bad argument");
    hipStream_t s = nullptr;
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(c, BQ_ERR_HIP, "hipSetDevice failed");
    HIPCHK(c, hipExtStreamCreateWithCUMask(&s, (uint32_t)mask_words, cu_mask));
    *out = (bq_stream_t)s;
    return BQ_OK;
}

int bq_set_option(bq_ctx* c, const char* name, int value) {
    if (!c || !name) return fail(c, BQ_ERR_ARG, "bq_set_option: bad argument");
    if (strcmp(name, "inflate_variant") == 0 && (value == 0 || value == 5)) { c->inflate_variant = value; return BQ_OK; }
    return fail(c, BQ_ERR_ARG, std::string("bq_set_option: unknown option or value: ") + name);
}

int bq_set_num_cus(bq_ctx* c, int n) {
    if (!c || n < 0 || n > 1024) return fail(c, BQ_ERR_ARG, "bq_set_num_cus: bad argument");
    int dev_cus = 256;
    if (n == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) dev_cus = prop.multiProcessorCount;
    }
    c->num_cus = n > 0 ? n : dev_cus;
    return BQ_OK;
}

int bq_stream_destroy(bq_ctx* c, bq_stream_t stream) {
    if (!c || !stream) return fail(c, BQ_ERR_ARG, "bq_stream_destroy: bad argument");
    HIPCHK(c, hipStreamDestroy((hipStream_t)stream));
    return BQ_OK;
}

int bq_stage_f32(bq_ctx* c, const float* d_tiles, int n, void* d_out, bq_stream_t stream) {
    if (!c || !d_tiles || !d_out || n < 0 || n > c->cfg.max_batch) return fail(c, BQ_ERR_ARG, "bq_stage_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(c, s, "stage_f32_to_planar", 0.0, (double)n * kStaged * (4.0 + esize(c)));
    if (launch_stage_f32(d_tiles, n, 299, d_out, c->cfg.dtype, s)) return fail(c, BQ_ERR_HIP, "stage launch failed");
    return BQ_OK;
}

int bq_backbone(bq_ctx* c, const void* d_in, int n, float* d_feat, void* d_ws, size_t ws_bytes,
                bq_stream_t stream) {
    if (!c || !d_in || !d_feat || !d_ws || n <= 0 || n > c->cfg.max_batch) return fail(c, BQ_ERR_ARG, "bq_backbone: bad argument");
    if (!c->loaded) return fail(c, BQ_ERR_WEIGHTS, "weights not loaded");
    if (ws_bytes < ws_layout(c, n, 1).total) return fail(c, BQ_ERR_WORKSPACE, "workspace too small");
    return backbone_impl(c, d_in, n, d_feat, (unsigned char*)d_ws, (hipStream_t)stream, nullptr);
}

int bq_mc_head(bq_ctx* c, const float* d_feat, int n, int64_t tile_idx0, int mc_n, int pass0, uint64_t seed,
               int init, int finalize, float* d_state, float* d_mean2, float* d_std2, void* d_ws,
               size_t ws_bytes, bq_stream_t stream) {
    if (!c || !d_feat || !d_state || !d_ws || n <= 0 || mc_n <= 0 || n > c->cfg.max_batch ||
        mc_n > c->cfg.max_mc || pass0 < 0 || (finalize && (!d_mean2 || !d_std2)))
        return fail(c, BQ_ERR_ARG, "bq_mc_head: bad argument");
    if (!c->loaded) return fail(c, BQ_ERR_WEIGHTS, "weights not loaded");
    if (ws_bytes < ws_layout(c, n, mc_n).total) return fail(c, BQ_ERR_WORKSPACE, "workspace too small");
    return head_impl(c, d_feat, n, tile_idx0, mc_n, pass0, seed, init, finalize, d_state, d_mean2, d_std2,
                     (unsigned char*)d_ws, (hipStream_t)stream);
}

int bq_set_tile_index_ptr(bq_ctx* c, const int64_t* d_tile_idx0) {
    if (!c) return BQ_ERR_ARG;
    c->d_tile0 = reinterpret_cast<const long long*>(d_tile_idx0);
    return BQ_OK;
}

int bq_set_tile_index_array(bq_ctx* c, const int64_t* d_tile_idx) {
    if (!c) return BQ_ERR_ARG;
    c->d_tile_idx = reinterpret_cast<const long long*>(d_tile_idx);
    return BQ_OK;
}

// uint8 tiles -> pooled features, the kernels bq_mc_infer runs: in a 16-bit context with the front weights loaded staging +
// block1_conv1 + block1_conv2 are ONE kernel straight from the bytes (kernels_front.hip), otherwise bq_stage + the backbone on
// the planar tensor.  bq_mc_infer and bq_backbone_u8 share it, so a tile's features do not depend on which of the two a
// caller used for its batch.
static int features_from_u8(bq_ctx* c, const uint8_t* d_tiles, int n, float* feat, unsigned char* ws, const WsLayout& L,
                            bq_stream_t stream, int part = 3) {
    hipStream_t s = (hipStream_t)stream;
    static const bool no_front = bq_exp_env("BQ_NO_FRONT") != nullptr;
    const bool front = !no_front && is16(c->cfg.dtype) && c->front_ws16 && c->front_wc16;
    if (front) return backbone_impl(c, nullptr, n, feat, ws, s, nullptr, d_tiles, part);
    void* staged = ws + L.staged;
    if (part & 1) RUN(bq_stage(c, d_tiles, n, staged, stream));
    return backbone_impl(c, staged, n, feat, ws, s, nullptr, nullptr, part);
}

int bq_mc_infer(bq_ctx* c, const uint8_t* d_tiles, int n, int64_t tile_idx0, int mc_n, uint64_t seed,
                int mc_mode, float* d_mean2, float* d_std2, void* d_ws, size_t ws_bytes, bq_stream_t stream) {
    if (!c || !d_tiles || !d_mean2 || !d_std2 || !d_ws || n <= 0 || mc_n <= 0 || n > c->cfg.max_batch ||
        mc_n > c->cfg.max_mc || (mc_mode != BQ_MC_HEAD && mc_mode != BQ_MC_FULL))
        return fail(c, BQ_ERR_ARG, "bq_mc_infer: bad argument");
    if (!c->loaded) return fail(c, BQ_ERR_WEIGHTS, "weights not loaded");
    const WsLayout L = ws_layout(c, n, mc_n);
    if (ws_bytes < L.total) return fail(c, BQ_ERR_WORKSPACE, "workspace too small");
    unsigned char* ws = (unsigned char*)d_ws;
    hipStream_t s = (hipStream_t)stream;
    float* feat = (float*)(ws + L.feat);
    float* state = (float*)(ws + L.state);
    if (mc_mode == BQ_MC_HEAD) {
        RUN(features_from_u8(c, d_tiles, n, feat, ws, L, stream));
        return head_impl(c, feat, n, tile_idx0, mc_n, 0, seed, 1, 1, state, d_mean2, d_std2, ws, s);
    }
    // BQ_MC_FULL: the reference's loop structure -- the whole network once per pass.
    for (int p = 0; p < mc_n; ++p) {
        RUN(features_from_u8(c, d_tiles, n, feat, ws, L, stream));
        RUN(head_impl(c, feat, n, tile_idx0, 1, p, seed, p == 0, p == mc_n - 1, state, d_mean2, d_std2, ws, s));
    }
    return BQ_OK;
}

#ifdef BQ_EXPERIMENTS
// bq_mc_infer (BQ_MC_HEAD) in two parts, for a scheduler that runs the entry parts and the rests of two batches against each other
// (tools/phased_pool.py): ENTRY = staging, stem and entry flow, whose output stays in the workspace; REST = middle and exit flow + the
// MC head.  Round 5 measured every such schedule equal to or slower than free-running streams (profiles/r05_schedules_steps.log), so
// the entry point is not part of the product library: `make EXPERIMENTS=1` builds it.
enum { BQ_PART_ENTRY = 1, BQ_PART_REST = 2, BQ_PART_ALL = 3 };
extern "C" int bq_mc_infer_part(bq_ctx* c, const uint8_t* d_tiles, int n, int64_t tile_idx0, int mc_n, uint64_t seed, int part,
                                float* d_mean2, float* d_std2, void* d_ws, size_t ws_bytes, bq_stream_t stream) {
    if (!c || !d_tiles || !d_mean2 || !d_std2 || !d_ws || n <= 0 || mc_n <= 0 || n > c->cfg.max_batch ||
        mc_n > c->cfg.max_mc || (part != BQ_PART_ENTRY && part != BQ_PART_REST && part != BQ_PART_ALL))
        return fail(c, BQ_ERR_ARG, "bq_mc_infer_part: bad argument");
    if (!c->loaded) return fail(c, BQ_ERR_WEIGHTS, "weights not loaded");
    const WsLayout L = ws_layout(c, n, mc_n);
    if (ws_bytes < L.total) return fail(c, BQ_ERR_WORKSPACE, "workspace too small");
    unsigned char* ws = (unsigned char*)d_ws;
    float* feat = (float*)(ws + L.feat);
    RUN(features_from_u8(c, d_tiles, n, feat, ws, L, stream, part));
    if (!(part & BQ_PART_REST)) return BQ_OK;
    return head_impl(c, feat, n, tile_idx0, mc_n, 0, seed, 1, 1, (float*)(ws + L.state), d_mean2, d_std2, ws, (hipStream_t)stream);
}
#endif

int bq_backbone_u8(bq_ctx* c, const uint8_t* d_tiles, int n, float* d_feat, void* d_ws, size_t ws_bytes, bq_stream_t stream) {
    if (!c || !d_tiles || !d_feat || !d_ws || n <= 0 || n > c->cfg.max_batch) return fail(c, BQ_ERR_ARG, "bq_backbone_u8: bad argument");
    if (!c->loaded) return fail(c, BQ_ERR_WEIGHTS, "weights not loaded");
    const WsLayout L = ws_layout(c, n, 1);
    if (ws_bytes < L.total) return fail(c, BQ_ERR_WORKSPACE, "workspace too small");
    return features_from_u8(c, d_tiles, n, d_feat, (unsigned char*)d_ws, L, stream);
}

int bq_slide_reduce(bq_ctx* c, const float* d_mean2, const float* d_std2, const int32_t* d_slide_idx, int n,
                    int n_slides, float tile_uq, int64_t* d_acc_pred, int64_t* d_acc_unc, int32_t* d_count,
                    bq_stream_t stream) {
    if (!c || !d_mean2 || !d_std2 || !d_slide_idx || !d_acc_pred || !d_acc_unc || !d_count || n < 0 || n_slides <= 0)
        return fail(c, BQ_ERR_ARG, "bq_slide_reduce: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(c, s, "slide_reduce", 2.0 * n, 20.0 * n);
    if (launch_slide_reduce(d_mean2, d_std2, d_slide_idx, n, n_slides, tile_uq, (long long*)d_acc_pred,
                            (long long*)d_acc_unc, d_count, s))
        return fail(c, BQ_ERR_HIP, "slide_reduce launch failed");
    return BQ_OK;
}

int bq_slide_finish(bq_ctx* c, const int64_t* d_acc_pred, const int64_t* d_acc_unc, const int32_t* d_count,
                    int n_slides, double* d_mean_pred, double* d_mean_unc, bq_stream_t stream) {
    if (!c || !d_acc_pred || !d_acc_unc || !d_count || !d_mean_pred || !d_mean_unc || n_slides <= 0)
        return fail(c, BQ_ERR_ARG, "bq_slide_finish: bad argument");
    if (launch_slide_finish((const long long*)d_acc_pred, (const long long*)d_acc_unc, d_count, n_slides,
                            d_mean_pred, d_mean_unc, (hipStream_t)stream))
        return fail(c, BQ_ERR_HIP, "slide_finish launch failed");
    return BQ_OK;
}

size_t bq_roc_workspace_bytes(int64_t n) { return roc_workspace_bytes((long long)n); }

int bq_roc_youden(bq_ctx* c, const double* d_score, const uint8_t* d_label, int64_t n, void* d_ws, size_t ws_bytes,
                  double* d_out6, bq_stream_t stream) {
    if (!c || !d_score || !d_label || !d_ws || !d_out6 || n <= 0 || n > 0x7fffffffLL)
        return fail(c, BQ_ERR_ARG, "bq_roc_youden: bad argument");
    if (ws_bytes < roc_workspace_bytes(n)) return fail(c, BQ_ERR_ARG, "bq_roc_youden: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(c, s, "roc_youden", 4.0 * n, 60.0 * n);
    const int e = launch_roc_youden(d_score, d_label, (long long)n, (unsigned char*)d_ws, ws_bytes, d_out6, s);
    if (e) return fail(c, BQ_ERR_HIP, std::string("roc_youden: ") + hipGetErrorString((hipError_t)e));
    return BQ_OK;
}

int bq_profile_enable(bq_ctx* c, int on) {
    if (!c) return BQ_ERR_ARG;
    c->prof = on != 0;
    if (on) {
        c->prof_recs.clear(); c->ev_used = 0;
        c->prof_names.clear(); c->prof_flops.clear(); c->prof_bytes.clear();
        c->prof_launches.clear(); c->prof_ms.clear();
    }
    return BQ_OK;
}

int bq_profile_read(bq_ctx* c, bq_prof_entry* out, int max_entries) {
    if (!c || !out || max_entries <= 0) return BQ_ERR_ARG;
    for (const ProfRec& r : c->prof_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) return fail(c, BQ_ERR_HIP, "event sync failed");
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return fail(c, BQ_ERR_HIP, "event elapsed failed");
        c->prof_ms[r.cls] += ms;
        c->prof_launches[r.cls] += 1;
    }
    c->prof_recs.clear();
    c->ev_used = 0;
    int k = 0;
    for (size_t i = 0; i < c->prof_names.size() && k < max_entries; ++i, ++k) {
        memset(&out[k], 0, sizeof out[k]);
        strncpy(out[k].name, c->prof_names[i].c_str(), sizeof out[k].name - 1);
        out[k].launches = c->prof_launches[i];
        out[k].ms = c->prof_ms[i];
        const double nl = c->prof_launches[i] > 0 ? (double)c->prof_launches[i] : 1.0;
        out[k].flops = c->prof_flops[i] / nl;      // per launch, averaged over the class's launches
        out[k].bytes = c->prof_bytes[i] / nl;
    }
    return k;
}

int64_t bq_debug_activation(bq_ctx* c, const char* name, const void* d_in, int n, void* d_ws, size_t ws_bytes,
                            float* d_out, size_t out_elems, bq_stream_t stream) {
    if (!c || !name || !d_in || !d_ws || !d_out || n <= 0 || n > c->cfg.max_batch) return fail(c, BQ_ERR_ARG, "bq_debug_activation: bad argument");
    if (!c->loaded) return fail(c, BQ_ERR_WEIGHTS, "weights not loaded");
    const WsLayout L = ws_layout(c, n, 1);
    if (ws_bytes < L.total) return fail(c, BQ_ERR_WORKSPACE, "workspace too small");
    Tap t; t.want = name; t.out = d_out; t.out_elems = out_elems;
    unsigned char* ws = (unsigned char*)d_ws;
    const int r = backbone_impl(c, d_in, n, (float*)(ws + L.feat), ws, (hipStream_t)stream, &t);
    if (r != BQ_OK) return r;
    return t.written;
}

int64_t bq_debug_activation_u8(bq_ctx* c, const char* name, const uint8_t* d_tiles, int n, void* d_ws, size_t ws_bytes,
                               float* d_out, size_t out_elems, bq_stream_t stream) {
    if (!c || !name || !d_tiles || !d_ws || !d_out || n <= 0 || n > c->cfg.max_batch) return fail(c, BQ_ERR_ARG, "bq_debug_activation_u8: bad argument");
    if (!c->loaded) return fail(c, BQ_ERR_WEIGHTS, "weights not loaded");
    const WsLayout L = ws_layout(c, n, 1);
    if (ws_bytes < L.total) return fail(c, BQ_ERR_WORKSPACE, "workspace too small");
    Tap t; t.want = name; t.out = d_out; t.out_elems = out_elems;
    unsigned char* ws = (unsigned char*)d_ws;
    const int r = backbone_impl(c, nullptr, n, (float*)(ws + L.feat), ws, (hipStream_t)stream, &t, d_tiles);
    if (r != BQ_OK) return r;
    return t.written;
}

}  // extern "C"
