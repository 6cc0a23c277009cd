/* Host-side exercise of the C ABI for the ASan + UBSan build (tools/asan_host.sh).
 * Plain C: also proves that include/sd_hip.h is a valid C header.  Runs in the BUILD container (no GPU): every call
 * either validates its arguments and refuses, answers a size query, or reaches HIP and comes back with an error
 * code -- none may touch memory it does not own.  Never run on the GPU box (sanitizer runs are refused there). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sd_hip.h"

static int failures = 0;
#define EXPECT(cond, what) do { if (!(cond)) { ++failures; fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, what); } } while (0)

static void fill_layer(sd_layer* l, int cin, int cout, int taps, int dil, int dtype, float* w, float* v) {
  memset(l, 0, sizeof(*l));
  const int step = dtype == SD_DT_F16 ? 64 : 32;
  l->w = w; l->bias = v; l->scale = v; l->shift = v;
  l->cin = cin; l->cin_pad = (cin + step - 1) / step * step; l->cout = cout; l->taps = taps; l->dil = dil; l->w_dtype = dtype;
}

static void fill_weights(sd_ecapa_weights* w, int dtype, int C, float* big, float* vec) {
  memset(w, 0, sizeof(*w));
  w->w_dtype = dtype; w->n_mels = 80; w->channels = C; w->n_blocks = 3; w->res2_scale = 8; w->mfa_channels = 3 * C;
  w->att_channels = 128; w->emb_dim = 192; w->asp_eps = 1e-12f;
  fill_layer(&w->block0, 80, C, 5, 1, dtype, big, vec);
  for (int i = 0; i < 3; ++i) {
    sd_se_res2_block* b = &w->blocks[i];
    fill_layer(&b->tdnn1, C, C, 1, 1, dtype, big, vec);
    for (int j = 0; j < 7; ++j) fill_layer(&b->res2[j], C / 8, C / 8, 3, i + 2, dtype, big, vec);
    fill_layer(&b->tdnn2, C, C, 1, 1, dtype, big, vec);
    fill_layer(&b->se1, C, 128, 1, 1, SD_DT_F32, big, vec);
    fill_layer(&b->se2, 128, C, 1, 1, SD_DT_F32, big, vec);
  }
  fill_layer(&w->mfa, 3 * C, 3 * C, 1, 1, dtype, big, vec);
  fill_layer(&w->asp_tdnn_h, 3 * C, 128, 1, 1, dtype, big, vec);
  fill_layer(&w->asp_tdnn_g, 6 * C, 128, 1, 1, SD_DT_F32, big, vec);
  fill_layer(&w->asp_conv, 128, 3 * C, 1, 1, dtype, big, vec);
  fill_layer(&w->fc, 6 * C, 192, 1, 1, SD_DT_F32, big, vec);
}

int main(void) {
  EXPECT(sd_abi_version() == SD_ABI_VERSION, "abi version");
  EXPECT(sd_sizeof(0) == sizeof(sd_conv_args) && sd_sizeof(1) == sizeof(sd_layer) && sd_sizeof(2) == sizeof(sd_se_res2_block) &&
         sd_sizeof(3) == sizeof(sd_ecapa_weights) && sd_sizeof(-1) == 0 && sd_sizeof(1 << 30) == 0, "sd_sizeof");
  EXPECT(sd_last_error() != NULL, "last error string");
  (void)sd_device_count();

  /* profiling: bad kinds, null outputs, enable / disable with nothing recorded */
  double ms = 0, work = 0; long long n = 0;
  EXPECT(sd_profile_read(-1, &ms, &n, &work) < 0 && sd_profile_read(SD_PROF_KINDS, &ms, &n, &work) < 0, "profile kind range");
  EXPECT(sd_profile_read(0, NULL, &n, &work) < 0 && sd_profile_read(0, &ms, NULL, &work) < 0 && sd_profile_read(0, &ms, &n, NULL) < 0, "profile nulls");
  EXPECT(sd_profile_enable(1) == SD_OK && sd_profile_read(SD_PROF_SEG_SPLITK, &ms, &n, &work) == SD_OK && n == 0 && sd_profile_enable(0) == SD_OK, "profile empty");
  for (int key = -2; key < 12; ++key) (void)sd_set_tuning(key, -1);
  for (int key = -2; key < 12; ++key) (void)sd_set_tuning(key, 0);

  /* fbank plan: every refusal path of the host-side table builder */
  float* win = (float*)malloc(400 * sizeof(float));
  float* mel = (float*)calloc(201 * 80, sizeof(float));
  for (int i = 0; i < 400; ++i) win[i] = 0.54f - 0.46f * cosf(6.283185307f * (float)i / 400.f);
  for (int k = 0; k < 201; ++k) mel[k * 80 + (k * 80) / 201] = 1.0f;
  EXPECT(sd_fbank_plan_create(NULL, 400, 160, mel, 80, 0, 0, 1e-6f, -1.f) == NULL, "null window");
  EXPECT(sd_fbank_plan_create(win, 400, 160, NULL, 80, 0, 0, 1e-6f, -1.f) == NULL, "null mel");
  EXPECT(sd_fbank_plan_create(win, 4, 2, mel, 80, 0, 0, 1e-6f, -1.f) == NULL, "n_fft too small");
  EXPECT(sd_fbank_plan_create(win, 16384, 160, mel, 80, 0, 0, 1e-6f, -1.f) == NULL, "n_fft too large");
  EXPECT(sd_fbank_plan_create(win, 400, 0, mel, 80, 0, 0, 1e-6f, -1.f) == NULL, "hop 0");
  EXPECT(sd_fbank_plan_create(win, 400, 401, mel, 80, 0, 0, 1e-6f, -1.f) == NULL, "hop > n_fft");
  EXPECT(sd_fbank_plan_create(win, 400, 160, mel, 0, 0, 0, 1e-6f, -1.f) == NULL, "n_mels 0");
  EXPECT(sd_fbank_plan_create(win, 400, 160, mel, 4096, 0, 0, 1e-6f, -1.f) == NULL, "n_mels large");
  {                                          /* the generic framing (any sr): host tables for 200 / 80 are built, then the device is needed */
    sd_fbank_plan* g = sd_fbank_plan_create(win, 200, 80, mel, 40, 1, 0, 1e-6f, -1.f);
    if (g) { EXPECT(sd_fbank_num_frames(g, 8000) == 101, "generic frames"); (void)sd_fbank_workspace_bytes(g, 7, 8000); sd_fbank_plan_destroy(g); }
  }
  EXPECT(sd_fbank_plan_create(win, 400, 160, mel, 80, 7, 0, 1e-6f, -1.f) == NULL, "pad mode");
  EXPECT(sd_fbank_plan_create(win, 400, 160, mel, 80, 0, 7, 1e-6f, -1.f) == NULL, "log mode");
  win[3] += 0.25f;
  EXPECT(sd_fbank_plan_create(win, 400, 160, mel, 80, 0, 0, 1e-6f, -1.f) == NULL, "asymmetric window");
  win[3] -= 0.25f;
  /* a valid request: builds all host tables, then needs the device for their copies -> NULL here, a plan on a GPU box */
  sd_fbank_plan* plan = sd_fbank_plan_create(win, 400, 160, mel, 80, 1, 1, 1e-10f, 80.f);
  if (plan) {
    EXPECT(sd_fbank_num_frames(plan, 32000) == 201, "frames");
    (void)sd_fbank_workspace_bytes(plan, 10000, 32000);
    sd_fbank_plan_destroy(plan);
  }
  sd_fbank_plan_destroy(NULL);
  EXPECT(sd_fbank_num_frames(NULL, 32000) < 0 || sd_fbank_num_frames(NULL, 32000) == 201 || 1, "frames null plan");
  (void)sd_fbank_workspace_bytes(NULL, 1, 32000);
  EXPECT(sd_fbank_f32(NULL, NULL, 1, 32000, 1, NULL, 80, NULL, 0, NULL) < 0, "fbank nulls");
  EXPECT(sd_fbank_windows_f32(NULL, NULL, 0, NULL, 1, 32000, 1, NULL, 80, NULL, 0, NULL) < 0, "fbank windows nulls");

  /* size queries over ranges incl. zero, negative and overflow-sized arguments */
  const int ms_[] = {-5, 0, 1, 31, 32, 33, 224, 225, 256, 257, 100000};
  for (unsigned i = 0; i < sizeof(ms_) / sizeof(ms_[0]); ++i) {
    const int M = ms_[i];
    (void)sd_seg_gemm_scratch_bytes(M, 6144, 192); (void)sd_seg_gemm_scratch_bytes(M, 1024, 128); (void)sd_seg_gemm_scratch_bytes(M, 512, 0);
    (void)sd_seg_gemm_scratch_bytes(M, -32, 192); (void)sd_seg_gemm_scratch_bytes(M, 2147483616, 2147483647);
    (void)sd_colstat_floats(M, 1024); (void)sd_colstat_floats(M, -1); (void)sd_colstat_floats(2147483647, 2147483647);
    (void)sd_cosine_workspace_bytes(M, 192); (void)sd_cosine_split16_workspace_bytes(M, 192); (void)sd_cosine_workspace_bytes(2147483647, 2147483647);
    (void)sd_viterbi_workspace_bytes(M, 8); (void)sd_viterbi_workspace_bytes(2147483647, 64);
    (void)sd_res2net_chain_workspace_bytes(M % 9);
  }
  EXPECT(sd_seg_gemm_scratch_bytes(256, 6144, 192) > (size_t)4 << 20, "final FC at 256 rows needs more than 4 MB (ADVICE r4)");
  EXPECT(sd_seg_gemm_scratch_bytes(257, 6144, 192) == 0, "no grid split-K past 256 rows");
  EXPECT(sd_res2net_chain_supported(201, 128, 7, 3, 2) == 1 && sd_res2net_chain_supported(213, 128, 7, 3, 2) == 0 &&
         sd_res2net_chain_supported(201, 64, 7, 3, 2) == 0 && sd_res2net_chain_supported(201, 128, 8, 3, 2) == 0, "chain geometry");
  EXPECT(sd_asp_attend_pool_supported(SD_DT_F32, 201, 3072, 128) == 1 && sd_asp_attend_pool_supported(SD_DT_F32, 257, 3072, 128) == 0 &&
         sd_asp_attend_pool_supported(7, 201, 3072, 128) == 0, "fused pooling geometry");

  /* the conv operator family: zeroed args, then one field wrong at a time around a well-formed launch (which then fails in HIP: no device) */
  float* big = (float*)calloc((size_t)1 << 20, sizeof(float));
  float* vec = (float*)calloc(16384, sizeof(float));
  sd_conv_args a; memset(&a, 0, sizeof(a));
  EXPECT(sd_conv1d_cl_f32(NULL, NULL) < 0 && sd_conv1d_cl_f16(NULL, NULL) < 0 && sd_conv1d_cl_split16(NULL, NULL) < 0 && sd_seg_gemm_f32(NULL, NULL, 0, NULL) < 0, "null args");
  EXPECT(sd_conv1d_cl_f32(&a, NULL) < 0 && sd_conv1d_cl_f16(&a, NULL) < 0 && sd_conv1d_cl_split16(&a, NULL) < 0, "zeroed args");
  a.x = big; a.lda = 128; a.w = big; a.w_dtype = SD_DT_F32; a.y = big; a.ldo = 128; a.M = 64; a.T = 32; a.cin = 128; a.cin_pad = 128; a.cout = 128;
  a.taps = 3; a.dil = 2; a.bias = vec; a.act = SD_ACT_RELU; a.scale = vec; a.shift = vec;
  for (int field = 0; field < 14; ++field) {
    sd_conv_args b = a;
    switch (field) {
      case 0: b.M = -1; break;            case 1: b.T = 0; break;              case 2: b.cin = 0; break;
      case 3: b.cin_pad = 100; break;     case 4: b.cout = -4; break;          case 5: b.taps = 0; break;
      case 6: b.dil = 40; break;          case 7: b.lda = 64; break;           case 8: b.ldo = 64; break;
      case 9: b.a_col0 = 64; break;       case 10: b.o_col0 = -4; break;       case 11: b.M = 63; break;     /* M % T != 0 */
      case 12: b.act = 99; break;         case 13: b.colstat = vec; break;     /* colstat needs T >= 64, cout % 256 == 0 */
    }
    EXPECT(sd_conv1d_cl_f32(&b, NULL) < 0, "conv f32 refuses a malformed launch");
    b.w_dtype = SD_DT_F16; b.x_dtype = SD_DT_F16; b.y_dtype = SD_DT_F16;
    EXPECT(sd_conv1d_cl_f16(&b, NULL) < 0, "conv f16 refuses a malformed launch");
    b.w_dtype = SD_DT_SPLIT16; b.x_dtype = SD_DT_F32; b.y_dtype = SD_DT_F32;
    EXPECT(sd_conv1d_cl_split16(&b, NULL) < 0, "conv split16 refuses a malformed launch");
  }
  {
    sd_conv_args b = a;
    (void)sd_conv1d_cl_f32(&b, NULL);         /* well-formed: decides a kernel, tries to launch, reports HIP's error (no device here) */
    b.T = 1; b.M = 32; b.taps = 1; b.dil = 1; b.cin = 6144; b.cin_pad = 6144; b.lda = 6144; b.cout = 192; b.ldo = 192;
    (void)sd_seg_gemm_f32(&b, big, 64, NULL);                 /* scratch too small -> the plain operator */
    (void)sd_seg_gemm_f32(&b, big, (size_t)4 << 20, NULL);
    (void)sd_seg_gemm_f32(&b, (char*)big + 4, (size_t)4 << 20, NULL);     /* misaligned scratch */
  }
  EXPECT(sd_split16_pack_f32(NULL, 0, 0, 0, 0, 1.f, NULL, 0, NULL) < 0 || 1, "pack nulls");
  (void)sd_split16_pack_f32(big, 100, 3, 10, 33, 1.f, big, 64, NULL);

  /* pooling / SE / statistics / scores: nulls and impossible shapes */
  EXPECT(sd_seg_mean_f32(NULL, 0, 0, 0, 0, 0, NULL, NULL) < 0 || 1, "seg mean");
  (void)sd_seg_mean_std_f32(NULL, 0, 0, -1, -1, -1, 0.f, NULL, NULL);
  (void)sd_seg_mean_std_dt(NULL, 9, 0, 0, 1, 1, 4, 1, 0.f, NULL, NULL);
  (void)sd_se_scale_residual_f32(NULL, 0, NULL, NULL, 0, 0, NULL, 0, 0, 0, 0, 0, NULL);
  (void)sd_asp_pool_f32(NULL, 0, NULL, 0, 0, 0, 0, 0.f, NULL, NULL);
  (void)sd_asp_attend_pool_dt(NULL, NULL, NULL, SD_DT_F32, 0, 1, 300, 3072, 128, 1e-12f, NULL, NULL);
  (void)sd_colstat_finish_dt(NULL, NULL, NULL, 0, 0, 0, 0, 0, 0, 0, 0.f, NULL, NULL);
  (void)sd_l2norm_rows_f32(NULL, 0, -1, 0, 0.f, 1, NULL, 0, NULL);
  (void)sd_cosine_affinity_f32(NULL, 10, 192, NULL, 10, NULL, 0, NULL);
  (void)sd_cosine_affinity_f32(big, 10, 192, big, 4, big, 1 << 20, NULL);                 /* ldo < N */
  (void)sd_cosine_affinity_rows_f32(big, 10, 192, 7, 3, big, 10, big, 1 << 20, NULL);     /* row_lo > row_hi */
  (void)sd_cosine_affinity_rows_f32(big, 10, 192, 0, 11, big, 10, big, 1 << 20, NULL);    /* row_hi > N */
  (void)sd_cosine_affinity_rows_f32(big, 10, 192, 0, 10, big, 10, big, 16, NULL);         /* workspace too small */
  (void)sd_cosine_affinity_rows_split16(big, 10, 192, 0, 10, big, 10, big, 16, NULL);
  (void)sd_adjacent_cosine_f32(NULL, 0, 1, 192, 1e-8f, NULL, NULL);
  (void)sd_sim_argmax_f32(NULL, 0, 0, 0, NULL, 0, 0, NULL, NULL, NULL);
  (void)sd_topk_mean_std_f32(NULL, 0, 0, 0, 0, NULL, NULL);
  (void)sd_topk_mean_std_f32(big, 10, 4, 10, 11, big, NULL);                              /* k > n */
  (void)sd_asnorm_combine_f32(NULL, 0, 0, 0, NULL, NULL, NULL, 0, NULL);
  (void)sd_viterbi_f32(NULL, 0, 0, 0, 0.f, 0.f, NULL, 0, NULL, NULL);
  (void)sd_viterbi_f32(big, 8, 100, 65, -0.1f, -3.f, big, 1 << 20, (int32_t*)big, NULL);      /* K > 64 */

  /* ECAPA schedule: geometry checks, workspace sizes, too-small workspace, then a well-formed forward (HIP error without a device) */
  sd_ecapa_weights* w = (sd_ecapa_weights*)malloc(sizeof(sd_ecapa_weights));
  for (int dtype = 0; dtype < 2; ++dtype) {
    fill_weights(w, dtype, 1024, big, vec);
    size_t prev = 0;
    const int bs[] = {1, 16, 32, 224, 256, 257, 10000};
    for (unsigned i = 0; i < sizeof(bs) / sizeof(bs[0]); ++i) {
      const size_t need = sd_ecapa_workspace_bytes(w, bs[i], 201);
      /* grows with the batch, except across 256 -> 257 segments, where the grid split-K scratch of the per-segment layers (4.7 MB) goes away */
      EXPECT(need > prev || bs[i] == 257, "workspace grows with the batch");
      prev = need;
    }
    EXPECT(sd_ecapa_workspace_bytes(w, 0, 201) == 0 || 1, "empty batch");
    EXPECT(sd_ecapa_workspace_bytes(NULL, 1, 201) == 0, "null weights");
    int (*fwd)(const sd_ecapa_weights*, const float*, int, int, float*, void*, size_t, sd_stream_t) = dtype ? sd_ecapa_forward_f16 : sd_ecapa_forward_f32;
    EXPECT(fwd(NULL, big, 1, 201, big, big, 1 << 20, NULL) < 0, "forward: null weights");
    EXPECT(fwd(w, NULL, 1, 201, big, big, 1 << 20, NULL) < 0, "forward: null features");
    EXPECT(fwd(w, big, 1, 201, big, big, 1024, NULL) < 0, "forward: workspace too small");
    EXPECT(fwd(w, big, 0, 201, big, big, 1 << 20, NULL) <= 0, "forward: empty batch");
    EXPECT(fwd(w, big, 1, 2, big, big, 1 << 20, NULL) < 0, "forward: T shorter than the dilations");
    {                                         /* well-formed: the schedule carves its workspace and starts launching; HIP refuses without a device */
      const size_t need = sd_ecapa_workspace_bytes(w, 2, 201);
      void* ws = malloc(need);
      float* emb = (float*)malloc(2 * 192 * sizeof(float));
      float* feats = (float*)calloc((size_t)2 * 201 * 80, sizeof(float));
      const int rc = fwd(w, feats, 2, 201, emb, ws, need, NULL);
      EXPECT(rc != SD_OK || sd_device_count() > 0, "forward cannot succeed without a device");
      free(feats); free(emb); free(ws);
    }
    sd_ecapa_weights* bad = (sd_ecapa_weights*)malloc(sizeof(sd_ecapa_weights));
    for (int k = 0; k < 9; ++k) {
      memcpy(bad, w, sizeof(*w));
      switch (k) {
        case 0: bad->n_blocks = 0; break;            case 1: bad->n_blocks = SD_MAX_BLOCKS + 1; break;     case 2: bad->res2_scale = 1; break;
        case 3: bad->res2_scale = 17; break;         case 4: bad->channels = 1000; break;                  case 5: bad->mfa_channels = 1024; break;
        case 6: bad->block0.cout = 512; break;       case 7: bad->fc.w = NULL; break;                      case 8: bad->w_dtype = dtype ^ 1; break;
      }
      void* ws = malloc(64);
      EXPECT(fwd(bad, big, 1, 201, big, ws, (size_t)1 << 40, NULL) < 0, "forward refuses inconsistent weights");
      free(ws);
    }
    free(bad);
  }
  (void)sd_res2net_chain_f16(NULL, 0, 0, 0, NULL, 0, NULL, 0, NULL);
  (void)sd_res2net_chain_f16(big, 1024, 1, 300, w->blocks[0].res2, 7, big, 16, NULL);

  free(w); free(big); free(vec); free(win); free(mel);
  if (failures) { fprintf(stderr, "%d expectation(s) failed; last error: %s\n", failures, sd_last_error()); return 1; }
  printf("asan_host_driver: all host-side checks passed (last error string: \"%s\")\n", sd_last_error());
  return 0;
}
