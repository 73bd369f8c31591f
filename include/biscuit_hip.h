/* biscuit_hip.h -- C ABI of libbiscuit_hip.so: the MI355X (gfx950) implementation of
 * BISCUIT's tile-level MC-dropout inference hot path.
 *
 * The reference (jamesdolezal/biscuit) has no FFI/plugin interface: the path is entered
 * through Python calls into Slideflow/TensorFlow and leaves through a tile-prediction
 * table.  Each entry point below names the reference call site it replaces:
 *
 *   bq_stain_reinhard_fast  interface.wsi_normalizer.rgb_to_rgb(image)   results.py:251-252, hp.py:19
 *   bq_stage        tf.image.per_image_standardization(norm_image)      results.py:256
 *   bq_backbone     keras Xception(include_top=False, pooling='avg')     biscuit/hp.py:4,20,22
 *   bq_mc_head      the UQ loop behind UncertaintyInterface(model)(batch) -> (mean, std)
 *                                                                        results.py:234,257-258
 *                   (dropout 0.1 hard-wired on, 2x Dense(1024)           biscuit/hp.py:11,13,21;
 *                    hp.uq = True                                        biscuit/experiment.py:849)
 *   bq_mc_infer     Project.evaluate(model, outcome, ..., save_predictions=True) inner loop
 *                                                                        biscuit/experiment.py:917-922
 *   bq_slide_reduce groupby(level).mean() of y_pred / uncertainty after the
 *                   `uncertainty < tile_uq` filter                       biscuit/threshold.py:191-204,297-298
 *
 * Conventions: every device buffer and the stream belong to the caller; the library
 * allocates only the weights (bq_load_weights) and small per-context scratch at
 * bq_create.  All work is enqueued on the caller's stream with no hidden
 * synchronisation (except bq_profile_read).  Return 0 on success, <0 on error; the
 * message is available from bq_last_error.  One context per device per process; a
 * context is not re-entrant.  No C++ exception crosses this boundary.
 */
#ifndef BISCUIT_HIP_H
#define BISCUIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bq_ctx bq_ctx;
typedef void* bq_stream_t; /* hipStream_t */

/* Storage / matrix-core type of the backbone activations and weights (accumulation, folded BN, the MC head and
 * every statistic are fp32 in all three).  F16 = IEEE half: the matrix-core rate of BF16 with 8x finer rounding;
 * values beyond +-65504 saturate (MODE.FP16_OVFL) instead of overflowing to inf. */
enum { BQ_DTYPE_F32 = 0, BQ_DTYPE_BF16 = 1, BQ_DTYPE_F16 = 2 };
enum { BQ_MC_HEAD = 0, BQ_MC_FULL = 1 };

enum {
    BQ_OK = 0,
    BQ_ERR_ARG = -1,      /* bad argument / shape */
    BQ_ERR_HIP = -2,      /* a HIP runtime call failed */
    BQ_ERR_WEIGHTS = -3,  /* weight blob malformed or not loaded */
    BQ_ERR_WORKSPACE = -4 /* workspace too small */
};

typedef struct bq_config {
    int32_t dtype;        /* BQ_DTYPE_*: storage/matrix-core type of the backbone activations */
    int32_t tile_px;      /* 299 (biscuit/hp.py:5) */
    int32_t n_classes;    /* 2 (LUAD vs LUSC) */
    float dropout;        /* 0.1 (biscuit/hp.py:11) */
    int32_t max_batch;    /* largest n any call will pass (sizes the workspace) */
    int32_t max_mc;       /* largest mc_n any call will pass */
} bq_config;

/* Lifetime. */
bq_ctx* bq_create(int device_id, const bq_config* cfg);
void bq_destroy(bq_ctx* ctx);
const char* bq_last_error(bq_ctx* ctx); /* ctx may be NULL: last creation error */

/* Bytes of device workspace bq_backbone/bq_mc_head/bq_mc_infer need for `batch` tiles
 * and `mc_n` passes. */
size_t bq_workspace_bytes(bq_ctx* ctx, int batch, int mc_n);

/* Upload a "BQW1" weight blob (biscuit_amd/weights.py:pack_blob; folded-BN scale/bias,
 * matrix-core weights pre-swizzled into MFMA fragment order).  The blob's dtype must
 * match cfg.dtype. */
int bq_load_weights(bq_ctx* ctx, const void* host_blob, size_t nbytes);

/* K0: uint8 NHWC tiles [n,px,px,3] -> per-image standardised planar NCHW [n,3,px,px]
 * of the context dtype.  (x - mean) / max(std, 1/sqrt(N)). */
int bq_stage(bq_ctx* ctx, const uint8_t* d_tiles_nhwc, int n, void* d_out_nchw,
             bq_stream_t stream);

/* A HIP stream restricted to the compute units whose bits are set in cu_mask (mask_words 32-bit
 * words, bit i = CU i): lets two batches in flight own disjoint halves of the chip instead of
 * interleaving workgroups on every CU.  No reference counterpart (scheduling only). */
int bq_stream_create_masked(bq_ctx* ctx, const uint32_t* cu_mask, int mask_words, bq_stream_t* out_stream);
int bq_stream_destroy(bq_ctx* ctx, bq_stream_t stream);
/* Number of compute units the persistent kernels of this context size their grids for (default: all of the device's; 0 restores
 * that).  A context whose launches go to a CU-masked stream sets it to the CUs of the mask: a grid sized for the whole chip runs
 * there as two rounds of workgroups, each with its own prologue.  Results do not depend on it. */
int bq_set_num_cus(bq_ctx* ctx, int n);
/* Tuning knobs that change no result.  "inflate_variant": 5 (default) = bq_png_inflate decodes in rounds of a literal-only fast
 * phase (a 7-bit table and a 32-byte output ring per lane in LDS, 8 waves per CU) and a general phase only stalled lanes enter:
 * 27.6 k incompressible / 38.8 k photograph-like tiles a second on 16 CUs; 0 = the kernel without LDS, decode tables in the scratch
 * buffer (global memory / L2): the fallback, 2-3 x slower (profiles/r05_inflate.txt).  Anything else: BQ_ERR_ARG. */
int bq_set_option(bq_ctx* ctx, const char* name, int value);

/* K0, optional front half: the `reinhard_fast` stain normaliser hp.py:19 selects, applied to the
 * uint8 tile before the standardisation exactly where results.py:251-252 calls
 * interface.wsi_normalizer.rgb_to_rgb(image).  uint8 NHWC [n,px,px,3] -> uint8 NHWC; d_out may equal
 * d_tiles.  target_means3 / target_stds3 are HOST pointers to the model's params.json `norm_fit`
 * (CIE-LAB L, a, b).  The algorithm lives in Slideflow, not in the reference: parity unpinned
 * (oracle/stain.py states the arithmetic both sides implement). */
int bq_stain_reinhard_fast(bq_ctx* ctx, const uint8_t* d_tiles_nhwc, int n, const float* target_means3,
                           const float* target_stds3, uint8_t* d_out_nhwc, bq_stream_t stream);

/* Per-tile CIE-LAB channel statistics [n][6] = mean L, a, b, population std L, a, b: what the
 * normaliser's fit() stores as norm_fit for a target image. */
int bq_stain_lab_stats(bq_ctx* ctx, const uint8_t* d_tiles_nhwc, int n, float* d_stats6, bq_stream_t stream);

/* Input side, PNG tiles: the reversal of the PNG scanline filters on the device.  d_rows:
 * [n][px][1 + 3*px] bytes -- per row the filter-type byte and the filtered RGB bytes, i.e. the inflated IDAT stream of an
 * 8-bit RGB non-interlaced PNG, as libbiscuit_io's bqio_decode_rows delivers it (include/biscuit_io.h).  d_out: uint8 NHWC
 * [n][px][px][3], the tiles bq_stage / bq_mc_infer take.  Bit-exact with a host PNG decoder (tests/test_png_unfilter.py). */
int bq_png_unfilter(bq_ctx* ctx, const uint8_t* d_rows, int n, int px, uint8_t* d_out_nhwc, bq_stream_t stream);

/* Input side, PNG tiles, the inflate itself on the device (round 5): n zlib streams -- the concatenated IDAT payloads of n 8-bit RGB
 * non-interlaced PNG tiles, as libbiscuit_io's bqio_extract_z packs them: stream i = d_z[d_off[i] .. d_off[i] + d_len[i]), every
 * d_off[i] a multiple of 16, 32 readable bytes behind every stream -- are inflated to their px rows of 1 + 3 px bytes at d_rows +
 * i * rows_stride (rows_stride a multiple of 4, >= px (1 + 3 px) + 4), one stream per lane.  d_status[i] = 0 iff stream i is a
 * well-formed zlib stream that inflates to exactly px (1 + 3 px) bytes with a matching Adler-32 (what zlib's uncompress() accepts)
 * and every row's filter-type byte is 0..4 (what a PNG decoder accepts);
 * any other value: the tile's rows are undefined and the caller decodes that record on the host.  d_scratch: table space,
 * bq_png_inflate_scratch_bytes(n).  Follow with bq_png_unfilter_strided.  No reference counterpart (tf.io.decode_png under tf.data). */
size_t bq_png_inflate_scratch_bytes(int n);
int bq_png_inflate(bq_ctx* ctx, const uint8_t* d_z, const uint32_t* d_off, const uint32_t* d_len, int n, int px, uint8_t* d_rows,
                   size_t rows_stride, void* d_scratch, size_t scratch_bytes, int32_t* d_status, bq_stream_t stream);
int bq_png_unfilter_strided(bq_ctx* ctx, const uint8_t* d_rows, size_t rows_stride, int n, int px, uint8_t* d_out_nhwc,
                            bq_stream_t stream);

/* Variant for callers that already hold standardised float32 NHWC tiles (the
 * UncertaintyInterface contract, results.py:256-257): converts to planar NCHW. */
int bq_stage_f32(bq_ctx* ctx, const float* d_tiles_nhwc_f32, int n, void* d_out_nchw,
                 bq_stream_t stream);

/* K1-K5: staged tiles -> [n,2048] fp32 global-average-pooled features. */
int bq_backbone(bq_ctx* ctx, const void* d_in_nchw, int n, float* d_feat2048, void* d_ws,
                size_t ws_bytes, bq_stream_t stream);

/* K0-K5 straight from the bytes: uint8 NHWC tiles -> [n,2048] features through the kernels bq_mc_infer runs (16-bit contexts:
 * standardisation + block1_conv1 + block1_conv2 fused into one launch; fp32 contexts: bq_stage + bq_backbone).  For callers
 * that need the features of a batch before they decide how to drive the head -- biscuit_amd.inference.evaluate on a batch whose
 * Philox tile indices are not one consecutive run: a tile's result is then the same whichever entry its batch took. */
int bq_backbone_u8(bq_ctx* ctx, const uint8_t* d_tiles_nhwc, int n, float* d_feat2048, void* d_ws,
                   size_t ws_bytes, bq_stream_t stream);

/* K6: mc_n stochastic passes of the dropout head over n feature rows, Welford-folded
 * on the device.  Dropout masks are Philox4x32-10 keyed by `seed` with counter
 * (unit/4, layer, pass, tile_idx0 + row): independent of batching.
 * pass0/init/finalize let a caller fold passes over several calls (BQ_MC_FULL):
 * init=1 zeroes the running state, finalize=1 writes mean/std.  d_state is
 * [n][5] fp32 (count, mean0, mean1, M2_0, M2_1), caller-owned. */
int bq_mc_head(bq_ctx* ctx, const float* d_feat, int n, int64_t tile_idx0, int mc_n,
               int pass0, uint64_t seed, int init, int finalize, float* d_state,
               float* d_mean2, float* d_std2, void* d_ws, size_t ws_bytes,
               bq_stream_t stream);

/* Optional device-side addend to tile_idx0 of bq_mc_head / bq_mc_infer (NULL: none; the pointer must stay valid while
 * launches that were enqueued with it are in flight).  The Philox tile counter of row i becomes
 * tile_idx0 + *d_tile_idx0 + i, read by the kernels when they RUN: a captured HIP graph of the path (the one-tile
 * loop of results.py:250-258) can then be replayed for tile after tile by updating 8 bytes of device memory. */
int bq_set_tile_index_ptr(bq_ctx* ctx, const int64_t* d_tile_idx0);

/* Optional per-tile Philox indices for the bq_mc_head / bq_mc_infer calls that follow (NULL: back to consecutive indices; the array
 * -- int64 [n of the call], device memory -- must stay valid while launches enqueued with it are in flight): the tile counter of
 * row i becomes tile_idx0 (+ *d_tile_idx0) + d_tile_idx[i] instead of ... + i.  A batch that holds the ends and beginnings of
 * several slides -- tiles whose global indices are not one consecutive run -- then takes ONE call instead of one head call per run
 * (biscuit_amd.inference.evaluate; results identical to the per-run calls, tests/test_gpu_parity.py). */
int bq_set_tile_index_array(bq_ctx* ctx, const int64_t* d_tile_idx);

/* Fused convenience: uint8 tiles -> (mean[n,2], std[n,2]).  mc_mode BQ_MC_HEAD runs
 * the backbone once and the head mc_n times; BQ_MC_FULL re-runs the whole network per
 * pass like the reference loop.  Results are bit-identical between the two. */
int bq_mc_infer(bq_ctx* ctx, const uint8_t* d_tiles_nhwc, int n, int64_t tile_idx0,
                int mc_n, uint64_t seed, int mc_mode, float* d_mean2, float* d_std2,
                void* d_ws, size_t ws_bytes, bq_stream_t stream);

/* K7: per-slide sums of y_pred (= mean of P(class 1)) and uncertainty (= std of
 * P(class 1)) plus tile counts, restricted to tiles with uncertainty < tile_uq when
 * tile_uq is a positive finite number (threshold.py:297-298: `if tile_uq:` and strict
 * `<`).  Accumulates into caller-zeroed 64-bit fixed-point buffers (order-independent,
 * bit-reproducible); call bq_slide_finish to convert to doubles. */
int bq_slide_reduce(bq_ctx* ctx, const float* d_mean2, const float* d_std2,
                    const int32_t* d_slide_idx, int n, int n_slides, float tile_uq,
                    int64_t* d_acc_pred, int64_t* d_acc_unc, int32_t* d_count,
                    bq_stream_t stream);
int bq_slide_finish(bq_ctx* ctx, const int64_t* d_acc_pred, const int64_t* d_acc_unc,
                    const int32_t* d_count, int n_slides, double* d_mean_pred,
                    double* d_mean_unc, bq_stream_t stream);

/* Threshold search of the consumer on the device: Youden's J over the ROC curve of (label, score), i.e.
 * `thresh[argmax(tpr - fpr)]` of sklearn.metrics.roc_curve as the reference uses it for the tile-level
 * prediction threshold (threshold.py:145-155), the tile-level uncertainty threshold over every tile of the
 * cohort (threshold.py:417-426) and the slide-level ones (threshold.py:212-218, 449-455): first maximum, curve
 * starting at (0, 0) with threshold +inf, one point per distinct score, rates in float64 from exact counts.
 * d_score [n] float64, d_label [n] uint8 (non-zero = positive); d_out6 = [threshold, J, tpr, fpr, n_pos,
 * n_neg] (float64, device).  With only one class present the rates are undefined (the reference raises):
 * callers check n_pos / n_neg.  Workspace: bq_roc_workspace_bytes(n). */
size_t bq_roc_workspace_bytes(int64_t n);
int bq_roc_youden(bq_ctx* ctx, const double* d_score, const uint8_t* d_label, int64_t n,
                  void* d_ws, size_t ws_bytes, double* d_out6, bq_stream_t stream);

/* Per-kernel timing with HIP events on the launch stream (bench.py roofline leg).
 * bq_profile_enable(1) brackets every subsequent launch with events;
 * bq_profile_read synchronises and returns, per kernel class, the number of launches,
 * total milliseconds, algorithmic FLOPs and algorithmic bytes since the last enable. */
enum { BQ_PROF_MAX = 64 };
typedef struct bq_prof_entry {
    char name[48];
    int64_t launches;
    double ms;
    double flops;
    double bytes;
} bq_prof_entry;
int bq_profile_enable(bq_ctx* ctx, int on);
int bq_profile_read(bq_ctx* ctx, bq_prof_entry* out, int max_entries);

/* Test hook: run the backbone on staged tiles up to and including the named layer
 * ("staged", "block1_conv1", "block1_conv2", "block{2,3,4}_{res,sepconv1,sepconv2,out}",
 * "block{5..12}_sepconv{1,2}", "block{5..13}_out", "block13_{res,sepconv1,sepconv2}", "block14_sepconv{1,2}") and copy that
 * activation -- as STORED: with activation exponents in the blob (weights.py: pack_blob(act_exp=...)) 2^-k times the network's
 * value -- as fp32 NHWC
 * [n,H,W,C] (true channel count, padding stripped) into d_out.  Returns the number of
 * elements written, or <0. */
int64_t bq_debug_activation(bq_ctx* ctx, const char* name, const void* d_in_nchw, int n,
                            void* d_ws, size_t ws_bytes, float* d_out, size_t out_elems,
                            bq_stream_t stream);

/* The same test hook on the path bq_mc_infer takes in a 16-bit context: from the uint8 tiles [n,299,299,3] through the fused
 * front kernel (standardise + block1_conv1 + block1_conv2 in one launch, csrc/kernels_front.hip), then as above.  "staged"
 * and "block1_conv1" are not materialised on this path and are refused; every name from "block1_conv2" on is available. */
int64_t bq_debug_activation_u8(bq_ctx* ctx, const char* name, const uint8_t* d_tiles, int n,
                               void* d_ws, size_t ws_bytes, float* d_out, size_t out_elems,
                               bq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BISCUIT_HIP_H */
