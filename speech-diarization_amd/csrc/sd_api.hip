// Error reporting, library identity and the N x N cosine-affinity entry point.
#include <cstdlib>

#include "sd_common.h"
#include <atomic>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

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

extern "C" size_t sd_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(sd_conv_args);
    case 1: return sizeof(sd_layer);
    case 2: return sizeof(sd_se_res2_block);
    case 3: return sizeof(sd_ecapa_weights);
    default: return 0;
  }
}

extern "C" int sd_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return sd_set_error(SD_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  return n;
}

hipError_t sd_func_max_lds(const void* func, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(mu);
  const auto key = std::make_pair(dev, func);
  if (done.count(key)) return hipSuccess;
  e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.insert(key);
  return e;
}

// ---------------------------------------------------------------- event profiling
namespace {
struct ProfRec { hipEvent_t a, b; int kind; double work; };
std::atomic<bool> g_prof_on{false};
std::vector<ProfRec> g_prof;       // records in use
std::vector<hipEvent_t> g_pool;    // recycled events
std::mutex g_prof_mu;

hipEvent_t prof_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

SdProfScope::SdProfScope(int kind, hipStream_t s, double work) : slot(-1), stream(s) {
  if (!g_prof_on) return;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r{prof_event(), prof_event(), kind, work};
  if (!r.a || !r.b) return;
  (void)hipEventRecord(r.a, s);
  g_prof.push_back(r);
  slot = (int)g_prof.size() - 1;
}

SdProfScope::~SdProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (slot < (int)g_prof.size()) (void)hipEventRecord(g_prof[slot].b, stream);
}

extern "C" int sd_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (ProfRec& r : g_prof) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
  g_prof.clear();
  g_prof_on = on != 0;
  return SD_OK;
}

extern "C" int sd_profile_read(int kind, double* ms, long long* launches, double* work) {
  SD_CHECK_ARG(kind >= 0 && kind < SD_PROF_KINDS && ms && launches && work, "sd_profile_read: bad arguments");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double t = 0.0, w = 0.0;
  long long n = 0;
  for (ProfRec& r : g_prof) {
    if (r.kind != kind) continue;
    SD_CHECK_HIP(hipEventSynchronize(r.b));
    float dt = 0.f;
    SD_CHECK_HIP(hipEventElapsedTime(&dt, r.a, r.b));
    t += dt; w += r.work; ++n;
  }
  *ms = t; *launches = n; *work = w;
  return SD_OK;
}

static int pad32(int d) { return (int)(((long)d + 31) & ~31L); }     // (callers bound d: a row of more than 2^31 - 32 floats is refused)

extern "C" size_t sd_cosine_workspace_bytes(int N, int D) {
  if (N <= 0 || D <= 0) return 0;
  return ((size_t)N * (((size_t)D + 31) & ~(size_t)31) * sizeof(float) + 255) & ~(size_t)255;
}

// sklearn.metrics.pairwise.cosine_similarity(X): K = normalize(X) @ normalize(X).T with
// zero-norm rows left as zeros, dtype preserved [REF anti_stick_diarize.py:177]
// [REF diar_diag.py:215,219,278,355].  Rows are normalised once into the workspace
// (zero padded to a multiple of 32 columns); the whole matrix runs on sd_affinity.hip's
// triangle + mirror kernel (exact f32 MFMA), a row block (multi-GPU) or an unaligned output
// through the same implicit-GEMM operator as the pointwise convs.
extern "C" int sd_cosine_affinity_f32(const float* x, int N, int D, float* out, int ldo,
                                      void* ws_dev, size_t ws_bytes, sd_stream_t stream) {
  return sd_cosine_affinity_rows_f32(x, N, D, 0, N, out, ldo, ws_dev, ws_bytes, stream);
}

// Rows [row_lo, row_hi) of the same matrix: out [(row_hi-row_lo)][N].  This is the unit a
// rank computes when the affinity is row-block sharded across GPUs (the N x N result is
// never moved over xGMI, only the N x D embeddings are).
extern "C" int sd_cosine_affinity_rows_f32(const float* x, int N, int D, int row_lo, int row_hi, float* out, int ldo,
                                           void* ws_dev, size_t ws_bytes, sd_stream_t stream) {
  SD_CHECK_ARG(N >= 0 && D > 0 && D <= (1 << 24), "sd_cosine_affinity_f32: N=%d D=%d", N, D);
  SD_CHECK_ARG(row_lo >= 0 && row_lo <= row_hi && row_hi <= N, "sd_cosine_affinity_rows_f32: bad row block [%d,%d) of %d", row_lo, row_hi, N);
  if (N == 0 || row_lo == row_hi) return SD_OK;
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
  a.x = xn + (size_t)row_lo * Dp; a.lda = Dp; a.a_col0 = 0;
  a.w = xn; a.w_dtype = SD_DT_F32;
  a.y = out; a.ldo = ldo; a.o_col0 = 0;
  a.M = row_hi - row_lo; a.T = 1;
  a.cin = Dp; a.cin_pad = Dp; a.cout = N; a.taps = 1; a.dil = 1;
  a.act = SD_ACT_NONE; a.act2 = SD_ACT_NONE;
  // the whole matrix: tiles on and above the diagonal, each stored as is and transposed (half the MFMA work;
  // K[i][j] and K[j][i] are then the same bits).  SD_AFFINITY_SYM=0 (diagnostic) computes every tile.
  static const bool sym = [] { const char* e = sd_experiment_env("SD_AFFINITY_SYM"); return !e || atoi(e) != 0; }();
  static const bool own = [] { const char* e = sd_experiment_env("SD_AFFINITY_CONV"); return !(e && atoi(e) != 0); }();   // SD_AFFINITY_CONV=1: round 2's conv-kernel form
  if (sym && own && row_lo == 0 && row_hi == N && N % 4 == 0 && ldo % 4 == 0 && sd_aligned16(out))
    return sd_affinity_sym_f32(xn, Dp, N, Dp / 32, out, ldo, stream);       // sd_affinity.hip: 16x16x4 f32 MFMA, two workgroups per CU, 256-byte stores
  if (sym && row_lo == 0 && row_hi == N) return sd_conv1d_cl_f32_symmetric(&a, stream);
  return sd_conv1d_cl_f32(&a, stream);
}

// The same matrix at the f16 matrix-core rate with f32-level accuracy (BASELINE configs[4]: the 50k x 50k affinity): the
// normalised rows, scaled by 2^4 (low halves of typical entries ~0.07 stay clear of the f16 subnormals), are packed once as
// SD_DT_SPLIT16 (hi = f16(v), lo = f16(v - hi) per 32 values) and the product is the three-MFMA form hi.hi + hi.lo + lo.hi of
// the SAME packed matrix on both sides; the epilogue takes the 2^-8 back.  The whole matrix is computed on and above the diagonal only
// (sd_affinity.hip: 128 x 128 tiles, two workgroups per CU) and every off-diagonal tile is stored twice from the same accumulators, both
// copies as 256-byte runs; what is left is writing 4 N^2 bytes.
static int pad32s(int d) { return pad32(d); }

extern "C" size_t sd_cosine_split16_workspace_bytes(int N, int D) {
  if (N <= 0 || D <= 0) return 0;
  const size_t f32 = ((size_t)N * pad32(D) * sizeof(float) + 255) & ~(size_t)255;
  const size_t h = ((size_t)N * 2 * pad32s(D) * 2 + 255) & ~(size_t)255;
  const size_t sc = ((size_t)N * sizeof(float) + 255) & ~(size_t)255;
  return f32 + h + sc;
}

namespace {
__global__ void fill_f32_kernel(float* p, int n, float v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
}  // namespace

extern "C" int sd_cosine_affinity_rows_split16(const float* x, int N, int D, int row_lo, int row_hi, float* out, int ldo,
                                               void* ws_dev, size_t ws_bytes, sd_stream_t stream) {
  SD_CHECK_ARG(N >= 0 && D > 0 && D <= (1 << 24), "sd_cosine_affinity_rows_split16: N=%d D=%d", N, D);
  SD_CHECK_ARG(row_lo >= 0 && row_lo <= row_hi && row_hi <= N, "sd_cosine_affinity_rows_split16: bad row block [%d,%d) of %d", row_lo, row_hi, N);
  if (N == 0 || row_lo == row_hi) return SD_OK;
  SD_CHECK_ARG(x && out && ws_dev, "sd_cosine_affinity_rows_split16: null pointer");
  SD_CHECK_ARG(ldo >= N, "sd_cosine_affinity_rows_split16: ldo=%d < N=%d", ldo, N);
  if (ws_bytes < sd_cosine_split16_workspace_bytes(N, D))
    return sd_set_error(SD_ERR_WORKSPACE, "sd_cosine_affinity_rows_split16: workspace %zu < %zu bytes", ws_bytes,
                        sd_cosine_split16_workspace_bytes(N, D));
  const int Dp = pad32(D), Ds = pad32s(D);
  char* ws = static_cast<char*>(ws_dev);
  float* xn = reinterpret_cast<float*>(ws);
  const size_t f32 = ((size_t)N * Dp * sizeof(float) + 255) & ~(size_t)255;
  const size_t h = ((size_t)N * 2 * Ds * 2 + 255) & ~(size_t)255;
  char* xs = ws + f32;                                          // SD_DT_SPLIT16 [N][Ds]
  float* sc = reinterpret_cast<float*>(ws + f32 + h);           // [N] = 2^-8
  int e = sd_l2norm_rows_f32(x, D, N, D, 0.f, 1, xn, Dp, stream);
  if (e != SD_OK) return e;
  e = sd_split16_pack_f32(xn, Dp, 0, N, D, 16.f, xs, Ds, stream);
  if (e != SD_OK) return e;
  static const bool sym = [] { const char* e = sd_experiment_env("SD_AFFINITY_SYM"); return !e || atoi(e) != 0; }();
  if (sym && N % 4 == 0 && ldo % 4 == 0 && sd_aligned16(out) && row_lo == 0 && row_hi == N)
    return sd_affinity_sym_split16(xs, 2 * Ds, N, Ds / 32, out, ldo, 1.0f / 256.0f, stream);       // sd_affinity.hip: upper triangle + mirror
  // a row block, or rows that cannot take 16-byte stores: every tile of the block through the conv kernel, the 2^-8 as its per-column scale
  hipLaunchKernelGGL(fill_f32_kernel, dim3((N + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), sc, N, 1.0f / 256.0f);
  SD_CHECK_LAUNCH("fill_f32_kernel");
  sd_conv_args a = {};
  a.x = xs + (size_t)row_lo * 2 * Ds * 2; a.lda = Ds; a.a_col0 = 0; a.x_dtype = SD_DT_SPLIT16;
  a.w = xs; a.w_dtype = SD_DT_SPLIT16;
  a.y = out; a.ldo = ldo; a.o_col0 = 0; a.y_dtype = SD_DT_F32;
  a.M = row_hi - row_lo; a.T = 1;
  a.cin = D; a.cin_pad = Ds; a.cout = N; a.taps = 1; a.dil = 1;
  a.act = SD_ACT_NONE; a.act2 = SD_ACT_NONE;
  a.scale = sc;
  return sd_conv1d_cl_split16(&a, stream);
}
