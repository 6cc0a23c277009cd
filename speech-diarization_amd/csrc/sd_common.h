// Internal helpers shared by the HIP translation units of libsd_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include "sd_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

int sd_set_error(int code, const char* fmt, ...);
// SD_TUNE_F16_NARROW_TILES (sd_set_tuning): C-wide f16 / split16 layers with at most this many 256x256 tiles take the 128x128 kernel
#include <atomic>
std::atomic<long>& sd_f16_narrow_tiles();
std::atomic<long>& sd_t256_lockstep_tiles();     // SD_TUNE_T256_LOCKSTEP_TILES (negative: 4 x the CU count)
// library-internal entry points
int sd_conv1d_cl_f32_symmetric(const sd_conv_args* a, sd_stream_t stream);   // sd_conv_gemm.hip: x == w, upper triangle + mirror
int sd_conv1d_cl_f32_rows(const sd_conv_args* a, sd_stream_t stream, int* stat_rows);   // sd_conv_gemm.hip: sd_conv1d_cl_f32 that may write colstat in units of *stat_rows rows
int sd_colstat_finish_rows(const float* colstat, const float* pivot, const void* y, int y_dtype, int ldy, int y_col0, int B, int T, int C, int want_std,
                           float eps, float* out, int unit_rows, sd_stream_t stream);              // sd_pool.hip: sd_colstat_finish_dt for such units
int sd_affinity_sym_f32(const float* xn, int ldx, int N, int groups, float* out, long ldo, sd_stream_t stream);                          // sd_affinity.hip
int sd_affinity_sym_split16(const void* xs, int ldx, int N, int groups, float* out, long ldo, float alpha, sd_stream_t stream);   // sd_affinity.hip
int sd_cast_f32_f16(const float* x, long n, void* y, sd_stream_t stream);           // sd_pool.hip
int sd_asp_attend_pool_scaled(const void* a1, const void* wc, const void* h, int dtype, int ldh, int B, int T, int C, int att, float eps,
                              float w_scale, float* out, sd_stream_t stream);         // sd_asp_fused.hip: sd_asp_attend_pool_dt with the split weights' 2^s
int sd_se_scale_residual_split(const void* x, int ldx, const float* gate, const void* res, int ldr, int r_col0, void* y, int ldy, int y_col0,
                               int B, int T, int C, int dtype, void* ys, int lds, int s_col0, sd_stream_t stream,
                               const void* res_split, int ld_rs, int rs_col0, int write_y);   // sd_pool.hip

#define SD_CHECK_ARG(cond, ...)                            \
  do {                                                     \
    if (!(cond)) return sd_set_error(SD_ERR_ARG, __VA_ARGS__); \
  } while (0)

#define SD_CHECK_HIP(expr)                                                              \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return sd_set_error(SD_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));   \
  } while (0)

// Kernel launches report configuration errors through hipGetLastError.
#define SD_CHECK_LAUNCH(name)                                                           \
  do {                                                                                  \
    hipError_t e_ = hipGetLastError();                                                  \
    if (e_ != hipSuccess)                                                               \
      return sd_set_error(SD_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
  } while (0)

// ---- optional per-kernel timing with HIP events on the launch stream (sd_profile_* in sd_hip.h)
struct SdProfScope {
  SdProfScope(int kind, hipStream_t stream, double work);
  ~SdProfScope();
  int slot;
  hipStream_t stream;
};

// hipFuncSetAttribute(func, MaxDynamicSharedMemorySize, bytes) once per (device, kernel): thread-safe, keyed by the
// calling thread's current device, so a second GPU in the same process or two threads racing the first call
// (the reference's web UI calls the pipeline from a worker thread) both get the attribute set before the launch.
hipError_t sd_func_max_lds(const void* func, int bytes);

// A/B switches (SD_F32_WIDE, SD_RES2_FUSED, ...) re-route kernels and exist for measurements only: they are read ONLY when the
// process also sets SD_EXPERIMENT=1, so a stray SD_* variable cannot change which kernels a product run uses.
static inline const char* sd_experiment_env(const char* name) {
  static const bool on = [] { const char* e = getenv("SD_EXPERIMENT"); return e && e[0] == '1' && e[1] == 0; }();
  return on ? getenv(name) : nullptr;
}

static inline bool sd_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// The value an f32 activation / operand takes on its way into an SD_DT_SPLIT16 pair (hi = f16(w), lo = f16(w - hi)): clamped to the
// f16 range, NaN kept (v_med3_f32 alone returns min3 of its operands for a NaN, i.e. -65504: a NaN would be laundered into a
// finite value).  Every split site uses this, so a producer that writes split halves gives the bits sd_split16_pack_f32 would.
__device__ __forceinline__ float sd_split16_clamp(float v) {
  const float c = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
  return v != v ? v : c;
}

// max(x, lo) that KEEPS a NaN x (ReLU: lo = 0; identity: lo = -inf).  v_max_f32 returns the other operand for a NaN, so a NaN activation
// (say the features sd_fbank writes for an utterance with a NaN sample) would become 0 in the first ReLU and the segment would leave the
// network with a plausible finite embedding; torch's relu / clamp propagate it [REF speech_encode.py:77 via speechbrain's TDNNBlock].
// One v_cmp + v_cndmask instead of one v_max per output element (-DSD_RELU_VMAX: the old form, for A/B timing only).
__device__ __forceinline__ float sd_max_keep_nan(float x, float lo) {
#ifdef SD_RELU_VMAX
  return fmaxf(x, lo);
#else
  return x < lo ? lo : x;
#endif
}

__device__ __forceinline__ float sd_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float sd_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
