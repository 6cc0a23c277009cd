// Error reporting, library identity and the N x N cosine-affinity entry point.
#include "sd_common.h"

namespace {
thread_local char g_err[512] = "";
}

int sd_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* sd_last_error(void) { return g_err; }
extern "C" int sd_abi_version(void) { return SD_ABI_VERSION; }

extern "C" int sd_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return sd_set_error(SD_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  return n;
}

static int pad32(int d) { return (d + 31) & ~31; }

extern "C" size_t sd_cosine_workspace_bytes(int N, int D) {
  if (N <= 0 || D <= 0) return 0;
  return ((size_t)N * pad32(D) * sizeof(float) + 255) & ~(size_t)255;
}

// sklearn.metrics.pairwise.cosine_similarity(X): K = normalize(X) @ normalize(X).T with
// zero-norm rows left as zeros, dtype preserved [REF anti_stick_diarize.py:177]
// [REF diar_diag.py:215,219,278,355].  Rows are normalised once into the workspace
// (zero padded to a multiple of 32 columns) and the product runs on the f32 matrix
// cores through the same implicit-GEMM operator as the pointwise convs.
extern "C" int sd_cosine_affinity_f32(const float* x, int N, int D, float* out, int ldo,
                                      void* ws_dev, size_t ws_bytes, sd_stream_t stream) {
  SD_CHECK_ARG(N >= 0 && D > 0, "sd_cosine_affinity_f32: N=%d D=%d", N, D);
  if (N == 0) return SD_OK;
  SD_CHECK_ARG(x && out && ws_dev, "sd_cosine_affinity_f32: null pointer");
  SD_CHECK_ARG(ldo >= N, "sd_cosine_affinity_f32: ldo=%d < N=%d", ldo, N);
  if (ws_bytes < sd_cosine_workspace_bytes(N, D))
    return sd_set_error(SD_ERR_WORKSPACE, "sd_cosine_affinity_f32: workspace %zu < %zu bytes", ws_bytes,
                        sd_cosine_workspace_bytes(N, D));
  const int Dp = pad32(D);
  float* xn = static_cast<float*>(ws_dev);
  int e = sd_l2norm_rows_f32(x, D, N, D, 0.f, 1, xn, Dp, stream);
  if (e != SD_OK) return e;
  sd_conv_args a = {};
  a.x = xn; a.lda = Dp; a.a_col0 = 0;
  a.w = xn; a.w_dtype = SD_DT_F32;
  a.y = out; a.ldo = ldo; a.o_col0 = 0;
  a.M = N; a.T = 1;
  a.cin = Dp; a.cin_pad = Dp; a.cout = N; a.taps = 1; a.dil = 1;
  a.act = SD_ACT_NONE; a.act2 = SD_ACT_NONE;
  return sd_conv1d_cl_f32(&a, stream);
}
