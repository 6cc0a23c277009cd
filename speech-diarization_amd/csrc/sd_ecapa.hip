// ECAPA-TDNN forward on one HIP stream: the layer schedule behind
// EncoderClassifier.encode_batch [REF speech_encode.py:73-78] / ECAPAEncoder.forward
// [REF ecapa_annote.py:13-22], geometry per SURVEY.md Appendix A.3.
//
// Activations are [B*T][C], channel contiguous, resident in a caller-provided workspace for
// the whole forward (no allocation, no synchronisation here, so the call can be captured in a
// hipGraph).  Two precisions share this schedule:
//   f32  — exact f32 MFMA everywhere (sd_ecapa_forward_f32);
//   f16  — frame-level layers (M = B*T rows) take f16 weights / f16 activations with f32
//          accumulation; the per-segment layers (M = B rows: SE gate, global-context bias,
//          final FC) and all statistics stay f32 (sd_ecapa_forward_f16).
// Data movement avoided by construction:
//   * block outputs are written straight into their slice of the [B*T][3C] buffer the
//     MFA conv reads (no torch.cat copy);
//   * tdnn1 writes into the Res2Net result buffer, so chunk 0 needs no copy;
//   * the Res2Net add  c_{j+1} + y_j  is produced by the epilogue of conv j ("tee");
//   * the attention TDNN's global-context columns (mean/std broadcast over T) collapse
//     to a per-segment bias computed by one tiny GEMM.
#include <cstdlib>

#include "sd_common.h"

namespace {

struct Carver {
  char* base; size_t off;
  void* take(size_t elems, size_t esz) {          // base == nullptr: a sizing pass (no pointer is formed from a null base)
    void* p = base ? base + off : nullptr;
    off += (elems * esz + 255) & ~(size_t)255;
    return p;
  }
};

struct Buffers {
  void *x0, *r, *t2, *xcat, *h, *s0, *s1, *a1, *e;   // activation dtype (e = attention logits)
  void* xs;                                            // split16 mode: the SD_DT_SPLIT16 copy of a wide conv's input (pack scratch)
  void* xcs;                                           // split16 mode: SD_DT_SPLIT16 twin of xcat, written by the SE scale + residual kernel
  void* x0s;                                           // split16 mode: the stem's output as SD_DT_SPLIT16 (block 1's tdnn1 input and shortcut)
  void* rs;                                            // split16 mode: the Res2Net output r as SD_DT_SPLIT16 (tdnn2's input), written by the narrow convs
  void* wpk; size_t wpk_bytes;                        // Res2Net chain weights in fragment order (f16 path)
  float *semean, *seh, *gate, *stats, *gbias, *pooled;
  void* skp; size_t skp_bytes;                          // partial sums of the per-segment layers' grid split-K (sd_seg_gemm_f32)
  size_t bytes;
};

int max_i(int a, int b) { return a > b ? a : b; }

Buffers carve(const sd_ecapa_weights* w, int B, int T, void* ws, int act_dtype) {
  const size_t M = (size_t)B * T;
  const size_t es = act_dtype == SD_DT_F16 ? 2 : 4;
  const int C = w->channels, Cm = w->mfa_channels, chunk = C / w->res2_scale;
  int se = 0;
  for (int i = 0; i < w->n_blocks; ++i) se = max_i(se, w->blocks[i].se1.cout);
  Carver c{static_cast<char*>(ws), 0};
  Buffers b;
  b.x0 = c.take(M * max_i(C, w->att_channels), es);
  b.r = c.take(M * C, es);
  b.t2 = c.take(M * C, es);
  // the concatenated block outputs are dead after the MFA conv; the attention logits reuse the space
  b.xcat = c.take(M * Cm, es);
  b.h = c.take(M * Cm, es);
  b.s0 = c.take(M * chunk, es);
  b.s1 = c.take(M * chunk, es);
  b.semean = static_cast<float*>(c.take((size_t)B * C, 4));
  b.seh = static_cast<float*>(c.take((size_t)B * se, 4));
  b.gate = static_cast<float*>(c.take((size_t)B * C, 4));
  b.stats = static_cast<float*>(c.take((size_t)B * 2 * Cm, 4));
  b.gbias = static_cast<float*>(c.take((size_t)B * w->att_channels, 4));
  b.pooled = static_cast<float*>(c.take((size_t)B * 2 * Cm, 4));
  // scratch of the grid split-K (M <= 256 rows): the largest of the three per-segment layers that use it (final FC at B = 256: 4.7 MB)
  b.skp_bytes = 0;
  if (B <= 256) {
    size_t need = sd_seg_gemm_scratch_bytes(B, w->fc.cin_pad, w->fc.cout);
    const size_t g = sd_seg_gemm_scratch_bytes(B, w->asp_tdnn_g.cin_pad, w->asp_tdnn_g.cout);
    need = g > need ? g : need;
    for (int i = 0; i < w->n_blocks; ++i) {
      const size_t s1 = sd_seg_gemm_scratch_bytes(B, w->blocks[i].se1.cin_pad, w->blocks[i].se1.cout);
      need = s1 > need ? s1 : need;
    }
    b.skp_bytes = (need + 255) & ~(size_t)255;
  }
  b.skp = c.take(b.skp_bytes, 1);
  b.wpk_bytes = act_dtype == SD_DT_F16 ? sd_res2net_chain_workspace_bytes(w->res2_scale - 1) : 0;
  b.wpk = c.take(b.wpk_bytes, 1);
  b.xs = (w->split16 == 1 && act_dtype == SD_DT_F32) ? c.take(M * (size_t)((Cm + 31) / 32 * 32), 4) : nullptr;
  b.xcs = (w->split16 == 1 && act_dtype == SD_DT_F32 && Cm % 32 == 0 && C % 32 == 0) ? c.take(M * (size_t)Cm, 4) : nullptr;
  b.x0s = (w->split16 == 1 && act_dtype == SD_DT_F32 && C % 32 == 0) ? c.take(M * (size_t)C, 4) : nullptr;
  b.rs = (w->split16 == 1 && act_dtype == SD_DT_F32 && C % 32 == 0 && (C / w->res2_scale) % 32 == 0) ? c.take(M * (size_t)C, 4) : nullptr;
  b.a1 = b.x0;    // block-0 output is dead once block 1 has consumed it
  b.e = b.xcat;
  b.bytes = c.off;
  return b;
}

int check_layer(const char* name, const sd_layer& l, int cin, int cout, int taps, int dtype) {
  SD_CHECK_ARG(l.w != nullptr, "sd_ecapa: layer %s has no weights", name);
  SD_CHECK_ARG(l.cin == cin && l.cout == cout && l.taps == taps,
               "sd_ecapa: layer %s is %d->%d k=%d, geometry wants %d->%d k=%d", name, l.cin, l.cout, l.taps, cin, cout, taps);
  SD_CHECK_ARG(l.w_dtype == dtype, "sd_ecapa: layer %s weights are dtype %d, schedule wants %d", name, l.w_dtype, dtype);
  return SD_OK;
}

int check_weights(const sd_ecapa_weights* w, int act_dtype) {
  SD_CHECK_ARG(w != nullptr, "sd_ecapa: null weights");
  SD_CHECK_ARG(w->w_dtype == act_dtype, "sd_ecapa: weights were packed for dtype %d, forward is dtype %d", w->w_dtype, act_dtype);
  SD_CHECK_ARG(w->n_blocks >= 1 && w->n_blocks <= SD_MAX_BLOCKS, "sd_ecapa: n_blocks=%d", w->n_blocks);
  SD_CHECK_ARG(w->res2_scale >= 2 && w->res2_scale - 1 <= SD_MAX_RES2, "sd_ecapa: res2_scale=%d", w->res2_scale);
  const int gran = act_dtype == SD_DT_F16 ? 8 : 4;
  SD_CHECK_ARG(w->channels % (gran * w->res2_scale) == 0, "sd_ecapa: channels=%d must be a multiple of %d*res2_scale", w->channels, gran);
  SD_CHECK_ARG(w->mfa_channels == w->n_blocks * w->channels, "sd_ecapa: mfa_channels=%d != n_blocks*channels", w->mfa_channels);
  SD_CHECK_ARG(w->n_mels % gran == 0 && w->att_channels % gran == 0, "sd_ecapa: n_mels / att_channels must be multiples of %d", gran);
  const int C = w->channels, Cm = w->mfa_channels, chunk = C / w->res2_scale;
  const int fd = act_dtype;       // frame-level layers
  const int sd = SD_DT_F32;       // per-segment layers
  if (int e = check_layer("block0", w->block0, w->n_mels, C, w->block0.taps, fd)) return e;
  for (int i = 0; i < w->n_blocks; ++i) {
    const sd_se_res2_block& b = w->blocks[i];
    if (int e = check_layer("tdnn1", b.tdnn1, C, C, 1, fd)) return e;
    for (int j = 0; j < w->res2_scale - 1; ++j)
      if (int e = check_layer("res2net", b.res2[j], chunk, chunk, b.res2[j].taps, fd)) return e;
    if (int e = check_layer("tdnn2", b.tdnn2, C, C, 1, fd)) return e;
    if (int e = check_layer("se1", b.se1, C, b.se1.cout, 1, sd)) return e;
    if (int e = check_layer("se2", b.se2, b.se1.cout, C, 1, sd)) return e;
    SD_CHECK_ARG(b.se1.cout % 4 == 0, "sd_ecapa: se_channels must be a multiple of 4");
  }
  if (int e = check_layer("mfa", w->mfa, Cm, Cm, 1, fd)) return e;
  if (int e = check_layer("asp_tdnn_h", w->asp_tdnn_h, Cm, w->att_channels, 1, fd)) return e;
  if (int e = check_layer("asp_tdnn_g", w->asp_tdnn_g, 2 * Cm, w->att_channels, 1, sd)) return e;
  if (int e = check_layer("asp_conv", w->asp_conv, w->att_channels, Cm, 1, fd)) return e;
  if (int e = check_layer("fc", w->fc, 2 * Cm, w->emb_dim, 1, sd)) return e;
  return SD_OK;
}

sd_conv_args conv_of(const sd_layer& l, const void* x, int x_dtype, int lda, int a_col0, void* y, int y_dtype, int ldo, int o_col0,
                     int M, int T, int act) {
  sd_conv_args a = {};
  a.x = x; a.lda = lda; a.a_col0 = a_col0; a.x_dtype = x_dtype;
  a.w = l.w; a.w_dtype = l.w_dtype;
  a.y = y; a.ldo = ldo; a.o_col0 = o_col0; a.y_dtype = y_dtype;
  a.M = M; a.T = T;
  a.cin = l.cin; a.cin_pad = l.cin_pad; a.cout = l.cout; a.taps = l.taps; a.dil = l.dil;
  a.bias = l.bias; a.bias_per_seg = 0;
  a.act = act; a.scale = l.scale; a.shift = l.shift; a.act2 = SD_ACT_NONE;
  return a;
}

// stat_rows (optional): receives the row unit of the column statistics the launch wrote (128 unless the exact-f32 operator picked one of
// its 80 / 96 / 112-row tiles: small launches)
int run_conv(const sd_conv_args& a, sd_stream_t stream, int* stat_rows = nullptr) {
  if (stat_rows) *stat_rows = 128;
  if (a.w_dtype == SD_DT_F16) return sd_conv1d_cl_f16(&a, stream);
  return stat_rows ? sd_conv1d_cl_f32_rows(&a, stream, stat_rows) : sd_conv1d_cl_f32(&a, stream);
}

// A wide layer of the f32 schedule: in split16 mode (and when the layer carries the second packing) its f32 input is
// re-written as SD_DT_SPLIT16 rows (4 bytes per value in, 4 out) and the conv runs on the f16 matrix cores with three
// products per value pair; otherwise the exact-f32 kernel.
// A narrow layer of the f32 schedule (Res2Net convs, attention TDNN): in split16 mode, when it carries the second packing, the
// 128x128 split kernel (f32 activations split while staged; tee / tee_add / per-segment bias as in the exact kernel).
// Small launches (at most 56 tiles of 128x128: up to ~35 two-second segments) stay on the exact-f32 operator, whose ring kernel spreads
// them over 64x64 / 32x64 tiles: measured per launch at 16 / 32 segments, Res2Net conv 9.5 / 14.3 us exact against 17.7 / 19.0 split (26 / 51
// workgroups of 128x128), attention TDNN 49 / 76 against 105 / 110; at 64 segments the split kernel is ahead again.
int run_narrow(const sd_layer& l, sd_conv_args a, bool split, sd_stream_t stream) {
  const long tiles128 = (long)((a.M + 127) / 128) * ((a.cout + 127) / 128);
  const bool small = a.T > 1 && tiles128 <= 56 && a.y_dtype != SD_DT_SPLIT16 && l.w != nullptr && l.w_dtype == SD_DT_F32;
  if (small || !(split && l.w_split && !l.bias_split && a.x_dtype == SD_DT_F32 && !a.colstat)) {
    if (a.y_dtype == SD_DT_SPLIT16) return sd_set_error(SD_ERR_UNSUPPORTED, "sd_ecapa_forward: a split output needs the split narrow kernel");
    return run_conv(a, stream);
  }
  a.w = l.w_split; a.w_dtype = SD_DT_SPLIT16; a.cin_pad = (l.cin + 31) / 32 * 32;
  a.w_scale_inv = l.split_scale_inv;
  return sd_conv1d_cl_split16(&a, stream);
}

// Small launches of the C-wide layers (measured, tools/probe_split16.py: 1024 -> 1024 at 32 segments 0.063 ms on the 128x128 split
// kernel against 0.085 ms for pack + 256x256 kernel, 0.041 against 0.079 at 16; from 64 segments up, and for 3C -> 3C always, the
// 256x256 kernel wins): at most 128 tiles of 256x256 and cout <= 1024 -> the narrow kernel, which has no column statistics.
// `narrow_tiles`: SD_TUNE_F16_NARROW_TILES (default 128), read ONCE per forward: the schedule decides up front which tensors exist
// only as SD_DT_SPLIT16, so every evaluation inside one forward must see the same value whatever sd_set_tuning() does meanwhile.
bool wide_goes_narrow(const sd_layer& l, int M, long narrow_tiles) {
  const long tiles = (long)((M + 255) / 256) * ((l.cout + 255) / 256);
  return l.cout <= 1024 && tiles <= narrow_tiles;
}

// The wide kernel's packing: weights scaled by 2^s with the scale folded into bias_split / scale_split.  A wide-ROLE layer (stem,
// tdnn1, tdnn2, MFA) of a small geometry (cout <= 256) may carry the NARROW packing instead (w_split + split_scale_inv, no folded
// vectors): it then runs on the 128x128 split kernel with its own bias / scale, or on the exact kernel.
bool wide_packed(const sd_layer& l) { return l.w_split && l.bias_split && l.scale_split; }

// twin / twin_ld: an SD_DT_SPLIT16 copy of a.x that already exists (same value columns a.a_col0 .. of rows of twin_ld value columns)
int run_wide(const sd_layer& l, sd_conv_args a, bool split, void* xs, long narrow_tiles, sd_stream_t stream, const void* twin = nullptr, int twin_ld = 0,
             int* stat_rows = nullptr) {
  const int cp = (l.cin + 31) / 32 * 32;
  if (stat_rows) *stat_rows = 128;
  if (split && l.w_split && !wide_packed(l)) {
    // narrow packing on a wide-role layer: the weights are scaled by 2^s and NOTHING else carries the scale, so the only split
    // kernel that may take them is the 128x128 one with w_scale_inv (the layer's own bias / scale); otherwise exact f32
    if (a.y_dtype == SD_DT_SPLIT16) return sd_set_error(SD_ERR_UNSUPPORTED, "sd_ecapa_forward: a split output needs the wide packing");
    if (l.split_scale_inv > 0.f && a.x_dtype == SD_DT_F32 && a.y_dtype == SD_DT_F32 && !a.colstat && l.cin % 4 == 0 && a.lda % 4 == 0 &&
        a.a_col0 % 4 == 0) {
      a.w = l.w_split; a.w_dtype = SD_DT_SPLIT16; a.cin_pad = cp; a.w_scale_inv = l.split_scale_inv;
      return sd_conv1d_cl_split16(&a, stream);
    }
    return run_conv(a, stream, stat_rows);
  }
  if (!(split && wide_packed(l) && xs && a.x_dtype == SD_DT_F32 && (a.y_dtype == SD_DT_F32 || a.y_dtype == SD_DT_SPLIT16) && !(a.tee && a.tee_add) &&
        !(a.colstat && a.T < 128))) {
    if (a.y_dtype == SD_DT_SPLIT16) return sd_set_error(SD_ERR_UNSUPPORTED, "sd_ecapa_forward: a split output needs the split wide kernel");
    return run_conv(a, stream, stat_rows);
  }
  if (a.y_dtype != SD_DT_SPLIT16 && wide_goes_narrow(l, a.M, narrow_tiles) && !a.colstat && l.cin % 4 == 0 && a.lda % 4 == 0 && a.a_col0 % 4 == 0) {
    a.w = l.w_split; a.w_dtype = SD_DT_SPLIT16; a.cin_pad = cp;           // f32 x stays: split while staged; the folded 2^s
    a.bias = l.bias_split; a.scale = l.scale_split; a.w_scale_inv = 0.f;    // form of bias / scale serves this kernel too
    return sd_conv1d_cl_split16(&a, stream);
  }
  if (twin && l.cin % 32 == 0 && a.a_col0 % 32 == 0) {
    a.x = twin; a.lda = twin_ld; a.x_dtype = SD_DT_SPLIT16;            // (a_col0 stays: the twin has the layout of the f32 buffer)
  } else {
    if (int e = sd_split16_pack_f32(static_cast<const float*>(a.x), a.lda, a.a_col0, a.M, l.cin, 1.f, xs, cp, stream)) return e;
    a.x = xs; a.lda = cp; a.a_col0 = 0; a.x_dtype = SD_DT_SPLIT16;
  }
  a.w = l.w_split; a.w_dtype = SD_DT_SPLIT16; a.cin_pad = cp;
  a.bias = l.bias_split; a.scale = l.scale_split;
  return sd_conv1d_cl_split16(&a, stream);
}

#define SD_TRY(expr)            \
  do {                          \
    int e_ = (expr);            \
    if (e_ != SD_OK) return e_; \
  } while (0)

int forward(const sd_ecapa_weights* w, const float* feats, int B, int T, float* emb, void* ws_dev, size_t ws_bytes,
            sd_stream_t stream, int dt) {
  SD_TRY(check_weights(w, dt));
  SD_CHECK_ARG(B >= 0 && T > 0, "sd_ecapa_forward: B=%d T=%d", B, T);
  if (B == 0) return SD_OK;
  SD_CHECK_ARG(feats && emb && ws_dev, "sd_ecapa_forward: null feats/emb/workspace");
  SD_CHECK_ARG((long)B * T < (1L << 31), "sd_ecapa_forward: B*T overflows int");
  SD_CHECK_ARG(sd_aligned16(ws_dev), "sd_ecapa_forward: workspace must be 16-byte aligned");
  const Buffers b = carve(w, B, T, ws_dev, dt);
  if (ws_bytes < b.bytes) return sd_set_error(SD_ERR_WORKSPACE, "sd_ecapa_forward: workspace %zu < %zu bytes", ws_bytes, b.bytes);

  const int M = B * T;
  const int C = w->channels, Cm = w->mfa_channels, chunk = C / w->res2_scale;
  const int F32 = SD_DT_F32;
  const size_t es = dt == SD_DT_F16 ? 2 : 4;
  // split16 = 1, "f32-split16x3": every frame-level contraction on the f16 matrix cores, three products per value pair, f32-level accuracy;
  // split16 = 2: only the narrow layers (Res2Net convs, attention TDNN, the attention logits of the pooling kernel); the wide layers,
  // 86 % of the flops, stay on the exact-f32 kernel
  const bool split = w->split16 != 0 && dt == SD_DT_F32;
  const bool wsplit = w->split16 == 1 && dt == SD_DT_F32;
  const long nt = sd_f16_narrow_tiles().load(std::memory_order_relaxed);      // one snapshot per forward
  // rows per unit of the column statistics, for sizing them: the exact-f32 operator may pick tiles of 80 / 96 / 112 rows (small launches);
  // the f16 and split kernels always write units of 128
  // the statistics buffers are sized for the smallest row unit a launch of this forward can write: an f32 forward can reach the exact
  // operator's 80 / 96 / 112-row tiles (sd_conv1d_cl_f32_rows) also WITH split weights, whenever run_wide falls through to run_conv
  const size_t stat_unit = dt == SD_DT_F32 ? 80 : 128;

  static const bool colstat_ok = [] {     // SD_COLSTAT=0: A/B switch for measurements
    const char* e = sd_experiment_env("SD_COLSTAT");
    return !(e && e[0] == '0');
  }();
  bool x0_split = false;                                 // the stem's output exists as SD_DT_SPLIT16 only (b.x0s)
  // block 0: TDNNBlock(n_mels -> C, k=5) on the f32 features.  f16: the features are rounded to f16 once (the
  // operand precision of that path anyway; t2 is free here), which lets the stem run on the LDS-DMA kernel of the
  // wide layers (-0.75 ms per 5000 segments: 106.6 -> 108.3 k segments/s).  SD_STEM_CAST=0: A/B switch.
  {
    static const bool cast_ok = [] { const char* e = sd_experiment_env("SD_STEM_CAST"); return !(e && e[0] == '0'); }();
    const void* x = feats;
    int xdt = F32;
    if (dt == SD_DT_F16 && cast_ok && ((long)M * w->n_mels) % 8 == 0 && sd_aligned16(feats)) {
      SD_TRY(sd_cast_f32_f16(feats, (long)M * w->n_mels, b.t2, stream));
      x = b.t2; xdt = SD_DT_F16;
    }
    sd_conv_args a = conv_of(w->block0, x, xdt, w->n_mels, 0, b.x0, dt, C, 0, M, T, SD_ACT_RELU);
    // f32-split16x3, wide layers split: both readers of the stem's output (block 1's tdnn1 and its shortcut) take SD_DT_SPLIT16, so the
    // stem writes that form and nothing else (no f32 tensor, no pack pass) -- unless a small launch sends tdnn1 to the narrow kernel
    x0_split = wsplit && b.x0s && b.xcs && wide_packed(w->block0) && wide_packed(w->blocks[0].tdnn1) && !wide_goes_narrow(w->block0, M, nt) &&
               !wide_goes_narrow(w->blocks[0].tdnn1, M, nt);
    if (x0_split) { a.y = b.x0s; a.y_dtype = SD_DT_SPLIT16; }
    SD_TRY(run_wide(w->block0, a, wsplit, b.xs, nt, stream));
  }
  const void* xin = b.x0; int ldin = C, colin = 0;
  // where the current block input exists as SD_DT_SPLIT16 (null: it does not), and whether its f32 form was skipped
  const void* in_sp = x0_split ? b.x0s : nullptr;
  int in_sp_ld = C, in_sp_col = 0;
  bool res_is_twin = x0_split;
  for (int i = 0; i < w->n_blocks; ++i) {
    const sd_se_res2_block& blk = w->blocks[i];
    // Res2Net chain: y_j = TDNN_j(c_j + y_{j-1}), written over chunk j of r.  f16: one kernel per block keeps the
    // segment's chain state in LDS (sd_res2net_f16.hip); otherwise (f32, or a segment too long for LDS) one conv per
    // step, chunk 1 teed to s0 by tdnn1 and the adds c_{j+1} + y_j produced by each conv's epilogue.
    static const bool chain_ok = [] {     // SD_RES2_FUSED=0: A/B switch for measurements
      const char* e = sd_experiment_env("SD_RES2_FUSED");
      return !(e && e[0] == '0');
    }();
    const sd_layer& r2 = blk.res2[0];
    bool r_split = false;                                // this block's Res2Net output exists as SD_DT_SPLIT16 (b.rs) instead of f32 chunks in b.r
    const bool chain = chain_ok && dt == SD_DT_F16 && w->res2_scale >= 2 &&
                       sd_res2net_chain_supported(T, chunk, w->res2_scale - 1, r2.taps, r2.dil);
    {
      sd_conv_args a = conv_of(blk.tdnn1, xin, dt, ldin, colin, b.r, dt, C, 0, M, T, SD_ACT_RELU);
      if (!chain) { a.tee = b.s0; a.ldt = chunk; a.tee_lo = chunk; a.tee_hi = 2 * chunk; }
      // (blocks 2..: the input is a slice of xcat, whose split twin the previous block's SE kernel has written)
      SD_TRY(run_wide(blk.tdnn1, a, wsplit, b.xs, nt, stream, in_sp, in_sp_ld));
    }
    if (chain) {
      SD_TRY(sd_res2net_chain_f16(b.r, C, B, T, blk.res2, w->res2_scale - 1, b.wpk, b.wpk_bytes, stream));
    } else {
      // f32-split16x3 with the wide layers split too: tdnn2 reads r as SD_DT_SPLIT16, so the narrow convs write their chunk in that
      // form directly (same bytes as the f32 chunk, no pack pass: 8 of the pass's 8 bytes per value go) and only chunk 0, which
      // tdnn1's output passes through unchanged, is packed; r itself keeps tdnn1's output (the tee_add source of every conv)
      r_split = wsplit && b.rs && wide_packed(blk.tdnn2) && !wide_goes_narrow(blk.tdnn2, M, nt);
      for (int j = 1; j < w->res2_scale && r_split; ++j) r_split = blk.res2[j - 1].w_split != nullptr && !blk.res2[j - 1].bias_split;
      if (r_split) SD_TRY(sd_split16_pack_f32(static_cast<const float*>(b.r), C, 0, M, chunk, 1.f, b.rs, C, stream));
      for (int j = 1; j < w->res2_scale; ++j) {
        void* src = (j & 1) ? b.s0 : b.s1;
        void* dst = (j & 1) ? b.s1 : b.s0;
        sd_conv_args a = conv_of(blk.res2[j - 1], src, dt, chunk, 0, b.r, dt, C, j * chunk, M, T, SD_ACT_RELU);
        if (r_split) { a.y = b.rs; a.y_dtype = SD_DT_SPLIT16; }
        if (j + 1 < w->res2_scale) {
          a.tee = dst; a.ldt = chunk; a.tee_lo = 0; a.tee_hi = chunk;
          a.tee_add = b.r; a.ld_ta = C; a.ta_col0 = (j + 1) * chunk;
        }
        SD_TRY(run_narrow(blk.res2[j - 1], a, split, stream));
      }
    }
    // tdnn2; the SE squeeze (mean over T) comes out of its epilogue as per-tile column sums where the
    // geometry allows (the Res2Net scratch s0 is dead and holds them), else from a pass over t2
    {
      sd_conv_args a = conv_of(blk.tdnn2, b.r, dt, C, 0, b.t2, dt, C, 0, M, T, SD_ACT_RELU);
      const bool stat = colstat_ok && T >= (wsplit ? 128 : 64) && C % 256 == 0 && !(wsplit && blk.tdnn2.w_split && wide_goes_narrow(blk.tdnn2, M, nt)) &&
                        (size_t)((M + stat_unit - 1) / stat_unit) * 6 * C * sizeof(float) <= (size_t)M * chunk * es;
      if (stat) a.colstat = static_cast<float*>(b.s0);
      int rows = 128;
      SD_TRY(run_wide(blk.tdnn2, a, wsplit, b.xs, nt, stream, r_split ? b.rs : nullptr, C, stat ? &rows : nullptr));
      if (stat) SD_TRY(sd_colstat_finish_rows(a.colstat, a.shift, b.t2, dt, C, 0, B, T, C, 0, 0.f, b.semean, rows, stream));
      else SD_TRY(sd_seg_mean_std_dt(b.t2, dt, C, 0, B, T, C, 0, 0.f, b.semean, stream));
    }
    // squeeze-excitation gate (per-segment, f32)
    {
      sd_conv_args a = conv_of(blk.se1, b.semean, F32, C, 0, b.seh, F32, blk.se1.cout, 0, B, 1, SD_ACT_RELU);
      SD_TRY(sd_seg_gemm_f32(&a, b.skp_bytes ? b.skp : nullptr, b.skp_bytes, stream));
      sd_conv_args a2 = conv_of(blk.se2, b.seh, F32, blk.se1.cout, 0, b.gate, F32, C, 0, B, 1, SD_ACT_SIGMOID);
      SD_TRY(run_conv(a2, stream));
    }
    // gate * t2 + shortcut -> slice i of the MFA input
    {
      // f32-split16x3 with the wide layers split: every reader of this slice of the MFA input takes its SD_DT_SPLIT16 copy (the next
      // block's tdnn1, the MFA conv, and the next block's shortcut below), so the f32 slice is not written and the shortcut is read
      // from the copy the previous block wrote -- unless a small launch routes the next tdnn1 to the narrow kernel, which stages f32
      const bool twin = wsplit && b.xcs != nullptr;
      const bool next_reads_f32 = i + 1 < w->n_blocks && wide_goes_narrow(w->blocks[i + 1].tdnn1, M, nt);
      const bool skip_f32 = twin && wide_packed(w->mfa) && !wide_goes_narrow(w->mfa, M, nt) && !next_reads_f32 &&
                            (i + 1 >= w->n_blocks || wide_packed(w->blocks[i + 1].tdnn1));
      const bool res_twin = twin && in_sp != nullptr && res_is_twin;
      SD_TRY(sd_se_scale_residual_split(b.t2, C, b.gate, xin, ldin, colin, b.xcat, Cm, i * C, B, T, C, dt,
                                        twin ? b.xcs : nullptr, Cm, i * C, stream,
                                        res_twin ? in_sp : nullptr, in_sp_ld, in_sp_col, skip_f32 ? 0 : 1));
      res_is_twin = skip_f32;                            // the next block's shortcut exists only as the split copy
      in_sp = twin ? b.xcs : nullptr; in_sp_ld = Cm; in_sp_col = i * C;
    }
    xin = b.xcat; ldin = Cm; colin = i * C;
  }
  // multi-layer feature aggregation; the global mean / std of attentive pooling likewise from the
  // epilogue (column sums in r, dead since the last block's tdnn2)
  {
    sd_conv_args a = conv_of(w->mfa, b.xcat, dt, Cm, 0, b.h, dt, Cm, 0, M, T, SD_ACT_RELU);
    const bool stat = colstat_ok && T >= (wsplit ? 128 : 64) && Cm % 256 == 0 && !(wsplit && w->mfa.w_split && wide_goes_narrow(w->mfa, M, nt)) &&
                      (size_t)((M + stat_unit - 1) / stat_unit) * 6 * Cm * sizeof(float) <= (size_t)M * C * es;
    if (stat) a.colstat = static_cast<float*>(b.r);
    int rows = 128;
    SD_TRY(run_wide(w->mfa, a, wsplit, b.xs, nt, stream, b.xcs, Cm, stat ? &rows : nullptr));
    if (stat) SD_TRY(sd_colstat_finish_rows(a.colstat, a.shift, b.h, dt, Cm, 0, B, T, Cm, 1, w->asp_eps, b.stats, rows, stream));
    else SD_TRY(sd_seg_mean_std_dt(b.h, dt, Cm, 0, B, T, Cm, 1, w->asp_eps, b.stats, stream));
  }
  // attentive statistics pooling with global context
  {
    sd_conv_args g = conv_of(w->asp_tdnn_g, b.stats, F32, 2 * Cm, 0, b.gbias, F32, w->att_channels, 0, B, 1, SD_ACT_NONE);
    g.scale = nullptr; g.shift = nullptr;
    SD_TRY(sd_seg_gemm_f32(&g, b.skp_bytes ? b.skp : nullptr, b.skp_bytes, stream));
    sd_conv_args a = conv_of(w->asp_tdnn_h, b.h, dt, Cm, 0, b.a1, dt, w->att_channels, 0, M, T, SD_ACT_RELU);
    a.bias = b.gbias; a.bias_per_seg = 1; a.act2 = SD_ACT_TANH;
    SD_TRY(run_narrow(w->asp_tdnn_h, a, split, stream));
    // asp.conv + softmax over T + weighted statistics: one kernel where the geometry allows (the
    // [M][3C] logits are then never stored), else the conv followed by the pooling kernel
    static const bool fuse_ok = [] {     // SD_ASP_FUSED=0: A/B switch for measurements
      const char* e = sd_experiment_env("SD_ASP_FUSED");
      return !(e && e[0] == '0');
    }();
    const bool fused = fuse_ok && w->asp_conv.taps == 1 && w->asp_conv.cin_pad == w->att_channels &&
                       sd_asp_attend_pool_supported(dt, T, Cm, w->att_channels);
    if (fused) {
      // (split16 mode: the same f32 tensors, the logits product on the f16 matrix cores with split operands)
      // (the weights' 2^s: asp_conv.split_scale_inv = 2^-s from the host, data dependent like every other split weight's)
      const float ws = w->asp_conv.split_scale_inv > 0.f ? 1.f / w->asp_conv.split_scale_inv : 256.f;
      SD_TRY(sd_asp_attend_pool_scaled(b.a1, w->asp_conv.w, b.h, split ? SD_DT_SPLIT16 : dt, Cm, B, T, Cm, w->att_channels, w->asp_eps, ws, b.pooled, stream));
    } else {
      sd_conv_args c = conv_of(w->asp_conv, b.a1, dt, w->att_channels, 0, b.e, dt, Cm, 0, M, T, SD_ACT_NONE);
      SD_TRY(run_conv(c, stream));
      SD_TRY(sd_asp_pool_dt(b.e, Cm, b.h, dt, Cm, B, T, Cm, w->asp_eps, b.pooled, stream));
    }
  }
  // asp_bn (folded into the weights by the host) + fc
  {
    sd_conv_args a = conv_of(w->fc, b.pooled, F32, 2 * Cm, 0, emb, F32, w->emb_dim, 0, B, 1, SD_ACT_NONE);
    SD_TRY(sd_seg_gemm_f32(&a, b.skp_bytes ? b.skp : nullptr, b.skp_bytes, stream));
  }
  return SD_OK;
}

}  // namespace

extern "C" size_t sd_ecapa_workspace_bytes(const sd_ecapa_weights* w, int B, int T) {
  if (!w || B <= 0 || T <= 0 || w->res2_scale <= 0) return 0;
  return carve(w, B, T, nullptr, w->w_dtype).bytes;
}

extern "C" int sd_ecapa_forward_f32(const sd_ecapa_weights* w, const float* feats, int B, int T, float* emb,
                                    void* ws_dev, size_t ws_bytes, sd_stream_t stream) {
  return forward(w, feats, B, T, emb, ws_dev, ws_bytes, stream, SD_DT_F32);
}

extern "C" int sd_ecapa_forward_f16(const sd_ecapa_weights* w, const float* feats, int B, int T, float* emb,
                                    void* ws_dev, size_t ws_bytes, sd_stream_t stream) {
  return forward(w, feats, B, T, emb, ws_dev, ws_bytes, stream, SD_DT_F16);
}
