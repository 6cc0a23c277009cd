/*
 * sd_hip.h — C ABI of libsd_hip.so, the MI355X (gfx950) embedding hot path.
 *
 * The reference (hzane/speech-diarization) has no native code and no FFI: its
 * boundary for this path is a set of Python callables that bottom out in
 * third-party libraries.  Each entry point below replaces the arithmetic behind
 * one of those call sites; the Python modules of the same names as the
 * reference's (speech_encode / ecapa_annote / ...) bind them with ctypes.
 *
 *   sd_fbank_*            replaces  torchaudio MelSpectrogram + log + mean-norm
 *                                   [REF speech_encode.py:17-36]  (fbank_batch)
 *                         and       speechbrain Fbank + InputNormalization inside
 *                                   encode_batch [REF speech_encode.py:77]
 *   sd_ecapa_*            replaces  speechbrain ECAPA_TDNN forward inside
 *                                   EncoderClassifier.encode_batch
 *                                   [REF speech_encode.py:73-78] [REF ecapa_annote.py:22]
 *   sd_conv1d_cl_f32 / _f16, sd_res2net_chain_f16, sd_seg_mean_std_*, sd_se_scale_residual_*, sd_asp_pool_*,
 *   sd_asp_attend_pool_dt, sd_colstat_finish_dt
 *                         the layer operators sd_ecapa_forward is built from: speechbrain's Conv1d /
 *                         TDNNBlock / SEBlock / AttentiveStatisticsPooling as reached from the same
 *                         encode_batch call [REF speech_encode.py:77]
 *   sd_cosine_affinity_f32 replaces sklearn cosine_similarity(X)
 *                                   [REF anti_stick_diarize.py:177] [REF diar_diag.py:215,219,278,355]
 *   sd_l2norm_rows_f32    replaces  X / (||X|| + 1e-8) [REF anti_stick_diarize.py:176,430]
 *   sd_adjacent_cosine_f32 replaces the einsum pair cosine [REF anti_stick_diarize.py:102-104]
 *   sd_sim_argmax_f32     replaces  argmax(W @ C.T) [REF anti_stick_diarize.py:433-434]
 *   sd_topk_mean_std_f32, sd_asnorm_combine_f32   the top-k cohort statistics and combination of
 *                                   asnorm_scores [REF diar_diag.py:196-208] (its cosine products run on sd_conv1d_cl_f32)
 *   sd_viterbi_f32        replaces  viterbi_hmm [REF diar_diag.py:231-247]
 *
 * Conventions: every pointer named *_dev / documented "device" is a HIP device
 * pointer; all functions are asynchronous on `stream`, never allocate and never
 * synchronise in the launch path (plan/weights creation excepted).  Return value
 * 0 = ok, negative = error (message via sd_last_error(), thread-local).
 */
#ifndef SD_HIP_H
#define SD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sd_stream_t; /* hipStream_t */

#define SD_OK 0
#define SD_ERR_ARG (-1)
#define SD_ERR_UNSUPPORTED (-2)
#define SD_ERR_WORKSPACE (-3)
#define SD_ERR_HIP (-4)

#define SD_ABI_VERSION 10

int sd_abi_version(void);
/* sizeof of the structs below as this library was compiled (which: 0 sd_conv_args, 1 sd_layer, 2 sd_se_res2_block,
 * 3 sd_ecapa_weights; anything else: 0) — lets a binding in another language verify its own layout at load time */
size_t sd_sizeof(int which);
const char* sd_last_error(void);
/* number of HIP devices visible, or negative error */
int sd_device_count(void);

/* Per-kernel timing with HIP events recorded on the launch stream, for bench.py's
 * roofline figures.  While enabled every launch of the named kernel families is
 * bracketed by two events; sd_profile_read synchronises them and returns the summed
 * duration, the launch count and the summed algorithmic work (flops for
 * SD_PROF_CONV_GEMM / SD_PROF_CONV_WIDE / SD_PROF_SEG_SPLITK, bytes for SD_PROF_FBANK) since the last sd_profile_enable(1). */
#define SD_PROF_CONV_GEMM 0
#define SD_PROF_FBANK 1
#define SD_PROF_CONV_WIDE 2   /* the 256x256 ring kernels of the wide layers (cout >= 1024), f32 and f16 */
#define SD_PROF_SEG_SPLITK 3  /* per-segment layers that took the grid split-K pair of sd_seg_gemm_f32 (one record per layer) */
#define SD_PROF_KINDS 4
int sd_profile_enable(int on);
int sd_profile_read(int kind, double* ms, long long* launches, double* work);

/* ------------------------------------------------------------------ fbank */

/* pad_mode */
#define SD_PAD_ZERO 0    /* speechbrain STFT: center=True, pad_mode="constant" */
#define SD_PAD_REFLECT 1 /* torchaudio Spectrogram: center=True, pad_mode="reflect" */
/* log_mode */
#define SD_LOG_LN_EPS 0   /* ln(x + eps)                           [REF speech_encode.py:32] */
#define SD_LOG_DB_TOPDB 1 /* 10*log10(max(x, eps)), floor at utterance max - top_db */

typedef struct sd_fbank_plan sd_fbank_plan;

/* window: host [n_fft] (must satisfy w[k] == w[n_fft-k], true for periodic
 * Hann/Hamming); mel_fb: host [n_fft/2+1][n_mels] row-major, n_mels <= 80 (any filter
 * shapes: the mel product is dense on the matrix cores).  Only n_fft=400, hop=160 (25 ms
 * / 10 ms at 16 kHz, [REF speech_encode.py:14-15]) is implemented.  Allocates
 * the device-side basis / filter tables. */
sd_fbank_plan* sd_fbank_plan_create(const float* window, int n_fft, int hop,
                                    const float* mel_fb, int n_mels,
                                    int pad_mode, int log_mode, float log_eps, float top_db);
void sd_fbank_plan_destroy(sd_fbank_plan* plan);

/* frames per utterance of n samples: 1 + n / hop (center=True) */
int sd_fbank_num_frames(const sd_fbank_plan* plan, int n);
size_t sd_fbank_workspace_bytes(const sd_fbank_plan* plan, int B, int n);

/* wav_dev: device f32 [B][n]; out_dev: device f32 [B][T][n_mels]
 * (ld_out = row stride of out in floats, >= n_mels);
 * mean_norm != 0 subtracts each utterance's per-bin mean over T.
 * Input domain: samples with |x| <= 16 are processed exactly (audio is normalised to [-1, 1] everywhere on this
 * path); larger magnitudes are clipped to +-16 (the split-f16 DFT scales the folded sums by 2^10 and they must
 * stay inside the f16 range), so any finite input gives finite features; an infinite sample saturates like any
 * other magnitude above 16.  A NaN sample makes every feature of ITS row NaN (what the reference's utterance-level
 * floor and mean over T do with it; the row's embedding is then NaN as well); other rows are unaffected. */
int sd_fbank_f32(const sd_fbank_plan* plan, const float* wav_dev, int B, int n,
                 int mean_norm, float* out_dev, int ld_out,
                 void* ws_dev, size_t ws_bytes, sd_stream_t stream);

/* The same transform over B windows of ONE signal (the reference's callers embed overlapping windows of one
 * recording: SCD windows [REF anti_stick_diarize.py:82-100], reassignment windows [REF anti_stick_diarize.py:396-430]):
 * row b is wav_dev[starts_dev[b] .. starts_dev[b] + n), with zeros wherever that index falls outside [0, n_total)
 * (the reference zero-pads short tails, [REF anti_stick_diarize.py:163-168]).  Bitwise the result of sd_fbank_f32 on
 * the gathered [B][n] matrix, without materialising it: the signal is uploaded once, not once per overlapping window.
 * wav_dev: device f32 [n_total]; starts_dev: device int64 [B]. */
int sd_fbank_windows_f32(const sd_fbank_plan* plan, const float* wav_dev, long long n_total, const long long* starts_dev,
                         int B, int n, int mean_norm, float* out_dev, int ld_out,
                         void* ws_dev, size_t ws_bytes, sd_stream_t stream);

/* --------------------------------------------------------- layer operators */

#define SD_ACT_NONE 0
#define SD_ACT_RELU 1
#define SD_ACT_TANH 2
#define SD_ACT_SIGMOID 3

#define SD_DT_F32 0
#define SD_DT_F16 1
#define SD_DT_SPLIT16 2 /* f32 VALUES carried as two f16 halves (hi = f16(v), lo = f16(v - hi)), interleaved per 32 values:
                           value column c sits at halfs 64 (c / 32) + (c % 32) (hi) and + 32 (lo) of its row */

/* Channel-last 1-D convolution as an implicit GEMM on the matrix cores:
 *   y[m, n] = act2( affine( act( bias[n] + sum_{j<taps} sum_{c<cin}
 *                 x[rowmap(m, j), a_col0 + c] * w[n][j][c] ) ) )
 * with m = b*T + t and rowmap reflecting t + (j - taps/2)*dil into [0, T)
 * ("same" padding, reflect mode — speechbrain Conv1d).  x: [M][lda],
 * w: packed [cout][taps][cin_pad] (cin_pad = cin rounded up to the kernel's K step —
 * 32 for f32 weights, 64 for f16 weights — zero filled),
 * y: [M][ldo] written at column o_col0.  Optional "tee": for output columns in
 * [tee_lo, tee_hi) also store y (+ tee_add[m, ta_col0 + n - tee_lo]) to
 * tee[m, n - tee_lo]; this carries the Res2Net chain's  c_{j+1} + y_j  add.
 * Element types: sd_conv1d_cl_f32 is all-f32 (exact f32 MFMA).  sd_conv1d_cl_f16 takes
 * f16 weights, x of x_dtype (f32 is converted while staging), accumulates in f32 on
 * v_mfma_f32_32x32x16_f16 and writes y / tee (reads tee_add) as y_dtype. */
typedef struct {
  const void* x; int lda; int a_col0;
  const void* w;  int w_dtype;
  void* y; int ldo; int o_col0;
  int M; int T;
  int cin; int cin_pad; int cout; int taps; int dil;
  const float* bias; int bias_per_seg; /* bias[n], or bias[(m / T) * cout + n] */
  int act;
  const float* scale; const float* shift; /* per-channel affine after act (eval BatchNorm), may be NULL */
  int act2;
  void* tee; int ldt; int tee_lo; int tee_hi;
  const void* tee_add; int ld_ta; int ta_col0;
  int x_dtype; int y_dtype; /* SD_DT_*; 0 = f32 (the only choice for sd_conv1d_cl_f32) */
  /* Optional per-segment column statistics from the epilogue (the SE squeeze mean, the global mean / std
   * of attentive pooling), so that y is not read back for them.  colstat: [ceil(M / 128)][6][cout] floats;
   * for every 128-row tile, sums over its (existing) rows of (y - shift) and (y - shift)^2, split at the
   * segment boundaries inside the tile (a tile spans up to three segments):
   * [sum 1st | sum 2nd | sum 3rd | sumsq 1st | sumsq 2nd | sumsq 3rd].  sd_colstat_finish_dt turns them into
   * [mean | std] per segment.  Requirements (SD_ERR_UNSUPPORTED otherwise): T >= 64, cout a multiple of 256,
   * relu / identity activation, per-channel bias, 16-byte aligned slices, no tee. */
  float* colstat;
  /* sd_conv1d_cl_split16 with f32 x only: the accumulator is multiplied by this (2^-s for weights packed with the
   * scale 2^s; exact) before the bias; 0 means 1 */
  float w_scale_inv;
} sd_conv_args;

int sd_conv1d_cl_f32(const sd_conv_args* args, sd_stream_t stream);
/* The same operator with caller-provided scratch, for the per-segment layers of a small batch (T = 1, one tap, M <= 256 rows, cin_pad >= 512:
 * SE squeeze FC, global-context bias, final FC): K is split over the grid in chunks of 128 (cin_pad < 2048) or 256 values, partial 32x32
 * tiles go to `scratch`, a second launch adds them in split order and applies the epilogue (deterministic; within f32 rounding of
 * sd_conv1d_cl_f32).  `scratch` must hold sd_seg_gemm_scratch_bytes(M, cin_pad, cout) bytes, 16-byte aligned, and must not be used by another
 * stream meanwhile; any other shape, a NULL or short scratch: exactly sd_conv1d_cl_f32. */
int sd_seg_gemm_f32(const sd_conv_args* args, void* scratch, size_t scratch_bytes, sd_stream_t stream);
/* bytes of scratch with which sd_seg_gemm_f32 splits K over the grid for this shape; 0 = it would not */
size_t sd_seg_gemm_scratch_bytes(int M, int cin_pad, int cout);
int sd_conv1d_cl_f16(const sd_conv_args* args, sd_stream_t stream);
/* "f32-split16x3": the same operator at f32-level accuracy on the f16 matrix cores.  Every f32 operand value is split
 * v = hi + lo (two f16) and a product is hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_f16 with f32 accumulation: the
 * dropped lo.lo term and the representation error are 2^-22 relative per product (exact f32: 2^-24), three f16 MFMAs
 * instead of one f32 MFMA at 1/16 of their rate.  x: SD_DT_SPLIT16 rows [M][lda] (lda, a_col0, cin, cin_pad count VALUE
 * columns; lda, a_col0 and cin_pad multiples of 32; made by sd_split16_pack_f32); w: SD_DT_SPLIT16 [cout][taps][cin_pad],
 * optionally pre-scaled by a power of two 2^s to keep the low halves of small weights out of the f16 subnormals (the
 * caller then passes bias * 2^s and scale * 2^-s: exact); y: f32.  256x256 tiles (the wide layers: cout >= 256 pays);
 * epilogue as sd_conv1d_cl_f16's 256x256 kernel (tee without tee_add; colstat needs T >= 128).
 * x may also be plain f32 (x_dtype SD_DT_F32; lda, a_col0, cin multiples of 4): the narrow form — a 128x128 tile
 * kernel that splits the activations while it stages them (no pack pass, any row slice, the full tee / tee_add
 * epilogue, per-segment bias; no colstat) and multiplies the accumulators by w_scale_inv = 2^-s instead of folding
 * the weight scale into bias / scale: the Res2Net convs and the attention TDNN of the f32-split16x3 mode.
 * y may be SD_DT_SPLIT16 instead of f32 (y_dtype; ldo, o_col0 in VALUE columns, ldo % 32 == 0, aligned slices, no colstat): the
 * result leaves as split halves, bit for bit what sd_split16_pack_f32 would make of the f32 result, for a consumer that is another
 * split conv (the narrow form, and the wide form through its LDS-staged epilogue); the tee copy stays f32.
 * Domain: |x| <= 65504 (larger values are clamped when packed / staged). */
int sd_conv1d_cl_split16(const sd_conv_args* args, sd_stream_t stream);
/* f32 [M][ldx] columns [col0, col0 + C), each multiplied by `mul` (a power of two: exact; 1 for activations) ->
 * SD_DT_SPLIT16 rows out [M][ldo] (ldo value columns, a multiple of 32, >= C rounded up to 32; the padding columns
 * are zero filled); 4 bytes per value in, 4 out. */
int sd_split16_pack_f32(const float* x, int ldx, int col0, int M, int C, float mul, void* out, int ldo, sd_stream_t stream);
/* Kernel-selection knobs (process-wide; for tests and measurements, results stay within f32 rounding).
 * SD_TUNE_SKINNY_TILES: sd_conv1d_cl_f32 launches with fewer 128x128 tiles than `value` run the 32x32
 * split-K kernel (default 128; 0 = always the 128x128 kernel; negative = restore the default). */
#define SD_TUNE_SKINNY_TILES 1
/* SD_TUNE_WIDE_TILES: sd_conv1d_cl_f32 launches with cout >= 1024 and at least `value` 256x256 tiles run the 256x256 ring
 * kernel (default 1024 = four rounds over the CUs; 0 = whenever the layer qualifies; negative = restore the default). */
#define SD_TUNE_WIDE_TILES 2
/* SD_TUNE_F16_NARROW_TILES: launches of layers with cout == 1024 (the C-wide layers) of sd_conv1d_cl_f16, and of the
 * f32-split16x3 schedule, with at most `value` 256x256 tiles run the 128x128 kernel (two workgroups per CU fill the chip better than a
 * fraction of one round of big tiles: default 128 = up to 32 two-second segments; 0 = always the 256x256 kernel; negative = default). */
#define SD_TUNE_F16_NARROW_TILES 3
/* SD_TUNE_S64_TILES: sd_conv1d_cl_f32 launches with T > 1, at least 64 rows and fewer 128x128 tiles than `value` (and no column
 * statistics) run the 64x64 ring kernel (32x64 tiles below 128 workgroups): the time-axis convs of small batches (default 128; 0 = never;
 * negative = default). */
#define SD_TUNE_S64_TILES 4
/* SD_TUNE_HALF_TILES: 128x64 instead of 128x128 tiles in sd_conv1d_cl_f32: 0 = never, 1 = whenever the layer allows (column statistics
 * need T >= 128), negative = by the rule (when they lower the number of tile times of the busiest CU; the default). */
#define SD_TUNE_HALF_TILES 5
/* SD_TUNE_TILE_ROWS: tiles of 80 / 96 / 112 rows x 128 columns in sd_conv1d_cl_f32 (16-row MFMA granularity, for launches whose 128-row
 * tiles divide badly over the CUs): 0 = never, 80 / 96 / 112 = that height whenever the layer allows, negative = by the rule (the default). */
#define SD_TUNE_TILE_ROWS 6
/* SD_TUNE_T256_LOCKSTEP_TILES: launches of the 256x256 f16 / f32-split16x3 ring kernel (register epilogue, column tiles in fours) with at least
 * `value` tiles run as 256 persistent workgroups with a static schedule -- per pass an XCD's 32 workgroups take one super-tile of 8 activation
 * row panels x 4 weight panels and share them in its L2 -- instead of one workgroup per tile: 0 = always, a huge value = never, negative = the
 * default (1024 tiles).  Same bits either way. */
#define SD_TUNE_T256_LOCKSTEP_TILES 7
int sd_set_tuning(int key, long value);
/* floats needed for sd_conv_args.colstat */
size_t sd_colstat_floats(int M, int cout);
/* colstat -> out [B][C] = mean (want_std = 0) or
 * [B][2*C] = [mean | sqrt(clamp(var, eps))]; pivot = the conv's shift vector (NULL = 0); y [B*T][ldy]
 * of y_dtype at column y_col0 is the conv's output */
int sd_colstat_finish_dt(const float* colstat, const float* pivot, const void* y, int y_dtype, int ldy, int y_col0,
                         int B, int T, int C, int want_std, float eps, float* out, sd_stream_t stream);

/* mean over the T rows of every segment: x [B*T][ld] cols [col0, col0+C) -> mean [B][C] */
int sd_seg_mean_f32(const float* x, int ld, int col0, int B, int T, int C,
                    float* mean, sd_stream_t stream);
/* mean and sqrt(clamp(mean((x-mean)^2), eps)) over T -> stats [B][2*C] = [mean | std] */
int sd_seg_mean_std_f32(const float* x, int ld, int col0, int B, int T, int C,
                        float eps, float* stats, sd_stream_t stream);
/* y[m, y_col0 + c] = x[m, c] * gate[m / T, c] + res[m, r_col0 + c]   (SE scale + shortcut) */
int sd_se_scale_residual_f32(const float* x, int ldx, const float* gate,
                             const float* res, int ldr, int r_col0,
                             float* y, int ldy, int y_col0,
                             int B, int T, int C, sd_stream_t stream);
/* attentive statistics pooling: a = softmax_T(logit); mu = sum a*h;
 * sd = sqrt(clamp(sum a*(h-mu)^2, eps));  out [B][2*C] = [mu | sd] */
int sd_asp_pool_f32(const float* logit, int ldl, const float* h, int ldh,
                    int B, int T, int C, float eps, float* out, sd_stream_t stream);
/* the same three operators with f16 (SD_DT_F16) or f32 activations (logits and h share the dtype);
 * statistics and gates stay f32 */
int sd_seg_mean_std_dt(const void* x, int x_dtype, int ld, int col0, int B, int T, int C,
                       int want_std, float eps, float* out, sd_stream_t stream);
int sd_se_scale_residual_dt(const void* x, int ldx, const float* gate, const void* res, int ldr, int r_col0,
                            void* y, int ldy, int y_col0, int B, int T, int C, int dtype, sd_stream_t stream);
int sd_asp_pool_dt(const void* logit, int ldl, const void* h, int dtype, int ldh,
                   int B, int T, int C, float eps, float* out, sd_stream_t stream);
/* attention-logit conv + attentive statistics pooling in one kernel: logits = a1 * wc^T (the conv's bias
 * cannot change a softmax over T and is not an argument), a = softmax_T(logits), out [B][2*C] = [mu | sd].
 * a1 [B*T][att] contiguous, wc packed [C][1][att], h [B*T][ldh], all in `dtype`.  Replaces, for
 * speechbrain's AttentiveStatisticsPooling, `asp.conv` + the pooling that sd_asp_pool_dt does on stored
 * logits.  sd_asp_attend_pool_supported() says whether the geometry is covered (att = 128,
 * C % 256 == 0, T <= 256); the _dt entry returns SD_ERR_UNSUPPORTED (nothing launched) otherwise. */
int sd_asp_attend_pool_supported(int dtype, int T, int C, int att);
int sd_asp_attend_pool_dt(const void* a1, const void* wc, const void* h, int dtype, int ldh,
                          int B, int T, int C, int att, float eps, float* out, sd_stream_t stream);

/* ------------------------------------------------------------ ECAPA-TDNN */

typedef struct {
  const void* w;      /* packed [cout][taps][cin_pad] */
  const float* bias;  /* [cout] or NULL */
  const float* scale; /* [cout] eval-BN scale or NULL */
  const float* shift; /* [cout] eval-BN shift or NULL */
  int cin, cin_pad, cout, taps, dil;
  int w_dtype;        /* SD_DT_F32 or SD_DT_F16 packing of w */
  /* optional second packing of the same layer for sd_conv1d_cl_split16 (NULL: not packed): SD_DT_SPLIT16
   * [cout][taps][cin rounded up to 32], scaled by 2^s.  Wide layers (SD_DT_SPLIT16 activations): bias_split =
   * bias * 2^s and scale_split = scale * 2^-s carry the scale; narrow layers (f32 activations): bias_split /
   * scale_split NULL and split_scale_inv = 2^-s goes into sd_conv_args.w_scale_inv */
  const void* w_split; const float* bias_split; const float* scale_split; float split_scale_inv;
} sd_layer;

#define SD_MAX_RES2 15
#define SD_MAX_BLOCKS 8

typedef struct {
  sd_layer tdnn1;
  sd_layer res2[SD_MAX_RES2]; /* scale-1 used */
  sd_layer tdnn2;
  sd_layer se1, se2;
} sd_se_res2_block;

/* speechbrain ECAPA_TDNN geometry (Appendix A.3 of SURVEY.md) */
typedef struct {
  int w_dtype;        /* SD_DT_F32: everything f32.  SD_DT_F16: f16 weights + f16 activations for the
                         frame-level layers (the per-segment M = B layers stay f32) */
  int n_mels;         /* 80 */
  int channels;       /* C = 1024: width of blocks 0..n_blocks */
  int n_blocks;       /* 3 SE-Res2Net blocks */
  int res2_scale;     /* 8 */
  int mfa_channels;   /* n_blocks * C = 3072 */
  int att_channels;   /* 128 */
  int emb_dim;        /* 192 */
  float asp_eps;      /* 1e-12 */
  int split16;        /* with w_dtype SD_DT_F32.  1, the "f32-split16x3" mode: every frame-level layer that carries a w_split packing
                         (wide: stem, tdnn1, tdnn2, MFA; narrow: Res2Net convs, attention TDNN) runs on sd_conv1d_cl_split16 and the
                         fused pooling kernel forms its logits from split operands.  2: the narrow layers and the logits only — the
                         wide layers (86 % of the flops) stay on the exact-f32 kernel.  0: exact f32 everywhere */
  sd_layer block0;
  sd_se_res2_block blocks[SD_MAX_BLOCKS];
  sd_layer mfa;
  sd_layer asp_tdnn_h; /* attention TDNN, columns acting on h      (mfa -> att) */
  sd_layer asp_tdnn_g; /* attention TDNN, columns acting on [mu|sd] (2*mfa -> att), bias here */
  sd_layer asp_conv;   /* att -> mfa */
  sd_layer fc;         /* 2*mfa -> emb, asp_bn folded in */
} sd_ecapa_weights;

/* The Res2Net chain of one SE-Res2Net block as one kernel, in place on the tdnn1 output r [B*T][ld] (f16):
 *   y_1 = TDNN_1(c_1), y_j = TDNN_j(c_j + y_{j-1}) (j = 2..n), c_j = columns [128 j, 128 j + 128) of r, y_j written
 * over c_j; TDNN = BatchNorm(ReLU(conv_{k=3, dilation}(.) + bias)), "same" reflect padding inside each T-row segment.
 * layers: host array of n sd_layer (f16 packed weights, 128 -> 128, k = 3, one dilation).  Replaces, for speechbrain's
 * Res2NetBlock inside encode_batch [REF speech_encode.py:77], n launches of sd_conv1d_cl_f16 with the tee epilogue.
 * sd_res2net_chain_supported: chunk == 128, taps == 3, 1 <= n <= 7, dilation < T <= 212 (three [T][128] f16 buffers
 * in the 160 KB LDS); the entry returns SD_ERR_UNSUPPORTED (nothing launched) otherwise. */
int sd_res2net_chain_supported(int T, int chunk, int n, int taps, int dil);
size_t sd_res2net_chain_workspace_bytes(int n);     /* the n convs' weights in MFMA-fragment order (re-made by every call) */
int sd_res2net_chain_f16(void* r, int ld, int B, int T, const sd_layer* layers, int n, void* ws, size_t ws_bytes, sd_stream_t stream);

size_t sd_ecapa_workspace_bytes(const sd_ecapa_weights* w, int B, int T);

/* feats: device f32 [B][T][n_mels] (mean-normalised fbank); emb: device f32 [B][emb_dim] */
int sd_ecapa_forward_f32(const sd_ecapa_weights* w, const float* feats, int B, int T,
                         float* emb, void* ws_dev, size_t ws_bytes, sd_stream_t stream);
/* same schedule with f16 operands / f32 accumulation on the 1x1 and dilated convs and f16
 * activations in HBM (BASELINE.json configs[4]); feats and emb stay f32 */
int sd_ecapa_forward_f16(const sd_ecapa_weights* w, const float* feats, int B, int T,
                         float* emb, void* ws_dev, size_t ws_bytes, sd_stream_t stream);

/* ------------------------------------------------------ cosine / affinity */

/* xn[i] = x[i] / (||x[i]|| + eps_add), rows with ||x|| == 0 divided by 1 when
 * sklearn_zero_guard != 0 (sklearn.preprocessing.normalize semantics). */
int sd_l2norm_rows_f32(const float* x, int ldx, int N, int D, float eps_add, int sklearn_zero_guard,
                       float* xn, int ldo, sd_stream_t stream); /* columns [D, ldo) of xn are zero filled */
/* out [N][N] = normalize(X) @ normalize(X).T, sklearn cosine_similarity semantics.
 * ws_dev: N*D_pad floats (sd_cosine_workspace_bytes). */
size_t sd_cosine_workspace_bytes(int N, int D);
int sd_cosine_affinity_f32(const float* x, int N, int D, float* out, int ldo,
                           void* ws_dev, size_t ws_bytes, sd_stream_t stream);
/* rows [row_lo, row_hi) of the same matrix -> out [(row_hi-row_lo)][N]; the unit one rank
 * computes when the affinity is row-block sharded (same workspace size). */
int sd_cosine_affinity_rows_f32(const float* x, int N, int D, int row_lo, int row_hi, float* out, int ldo,
                                void* ws_dev, size_t ws_bytes, sd_stream_t stream);
/* The same rows with f16 matrix-core throughput and f32-level accuracy (|error| ~ 3e-7): the normalised rows
 * are split x = hi + lo (two f16 matrices) and hi.hi + hi.lo + lo.hi runs as one f16 GEMM with f32
 * accumulation.  For the large affinities of BASELINE configs[4]; same semantics as the _f32 entry. */
size_t sd_cosine_split16_workspace_bytes(int N, int D);
int sd_cosine_affinity_rows_split16(const float* x, int N, int D, int row_lo, int row_hi, float* out, int ldo,
                                    void* ws_dev, size_t ws_bytes, sd_stream_t stream);
/* sims[i] = <x[i], x[i+1]> / (||x[i]|| * ||x[i+1]|| + eps), i < N-1 */
int sd_adjacent_cosine_f32(const float* x, int ldx, int N, int D, float eps, float* sims, sd_stream_t stream);
/* best[i] = argmax_k <w[i], c[k]> (first max wins, numpy argmax), score[i] = max */
int sd_sim_argmax_f32(const float* w, int ldw, int N, int D, const float* c, int ldc, int K,
                      int32_t* best, float* score, sd_stream_t stream);

/* ------------------------------------------- score normalisation / smoothing */
/* Precondition of the three operators below: FINITE inputs.  A NaN is not treated as numpy treats it (the radix select
 * orders a positive NaN above every number, so it would enter the top-k and poison mean / std; the Viterbi recurrence
 * never selects a NaN candidate where np.argmax returns its index): "exact path equality" with the reference holds for
 * finite scores.  K == 1 in sd_viterbi_f32: pass log_move = 0 (the reference divides by K - 1 = 0 there). */

/* Per row of x [rows][ld] (n valid columns): mean and population std of its k largest values
 * (np.sort(x, axis=1)[:, -k:].mean / .std of `asnorm_scores` [REF diar_diag.py:201-204]; k is clipped to n).
 * out [rows][2] = [mean | std].  Exact selection (radix select), f32 arithmetic. */
int sd_topk_mean_std_f32(const float* x, int ld, int rows, int n, int k, float* out, sd_stream_t stream);
/* out[i][j] = 0.5 * ((raw[i][j] - qstat[i][0]) / (qstat[i][1] + 1e-6) + (raw[i][j] - rstat[j][0]) / (rstat[j][1] + 1e-6))
 * [REF diar_diag.py:205-208]; raw [nq][ld], out [nq][ldo]. */
int sd_asnorm_combine_f32(const float* raw, int ld, int nq, int nr, const float* qstat, const float* rstat,
                          float* out, int ldo, sd_stream_t stream);
/* Most likely state path through scores [T][ld] (K <= 64 states) under a transition matrix with log_stay on the
 * diagonal and log_move elsewhere, f32 arithmetic and first-maximum tie-breaking as `viterbi_hmm`
 * [REF diar_diag.py:231-247].  ws: sd_viterbi_workspace_bytes(T, K) (back pointers); path: int32 [T]. */
size_t sd_viterbi_workspace_bytes(int T, int K);
int sd_viterbi_f32(const float* scores, int ld, int T, int K, float log_stay, float log_move,
                   void* ws, size_t ws_bytes, int32_t* path, sd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SD_HIP_H */
