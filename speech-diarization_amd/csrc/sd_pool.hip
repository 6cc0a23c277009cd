// Per-segment reductions and elementwise stages of the ECAPA-TDNN forward that are
// not contractions: squeeze-excitation mean, SE gate * x + shortcut, global-context
// mean/std, attentive statistics pooling (softmax over T, weighted mean/std), plus
// the small cosine helpers of the callers (the call sites
// are [REF anti_stick_diarize.py:102-104,176,430,433-434]).
//
// Layout everywhere: activations are [B*T][ld] f32, channel contiguous, so a thread
// owns 4 consecutive channels (one 16-byte load per row) and walks time; a
// workgroup is 64 channel groups x 4 row phases, combined through LDS in a fixed
// order (results are run-to-run identical).
#include "sd_common.h"

namespace {

constexpr int CG = 64;  // channel groups (of 4) per workgroup
constexpr int RP = 4;   // row phases per workgroup

typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// 4 consecutive channels as f32, from f32 or f16 storage
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const _Float16* p) {
  const h4 h = *reinterpret_cast<const h4*>(p);
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = (float)h[e];
  return r;
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(_Float16* p, f32x4 v) {
  h4 h;
#pragma unroll
  for (int e = 0; e < 4; ++e) h[e] = (_Float16)v[e];
  *reinterpret_cast<h4*>(p) = h;
}

// part[RP][CG] reduction helper: returns the fixed-order sum over row phases (valid for rp == 0)
template <int CGv = CG, int RPv = RP>
__device__ __forceinline__ f32x4 combine_sum(f32x4* part, int rp, int cg, f32x4 v) {
  part[rp * CGv + cg] = v;
  __syncthreads();
  f32x4 s = part[cg];
#pragma unroll
  for (int k = 1; k < RPv; ++k) s += part[k * CGv + cg];
  __syncthreads();
  return s;
}
__device__ __forceinline__ f32x4 combine_max(f32x4* part, int rp, int cg, f32x4 v) {
  part[rp * CG + cg] = v;
  __syncthreads();
  f32x4 s = part[cg];
#pragma unroll
  for (int k = 1; k < RP; ++k) {
    const f32x4 o = part[k * CG + cg];
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = fmaxf(s[e], o[e]);
  }
  __syncthreads();
  return s;
}

// CGv channel groups (of 4) x RPv row phases = 256 threads: 64 x 4 (a thread walks a quarter of the segment's rows) for launches that fill the
// chip, 16 x 16 for the few segments of a small batch (16 segments x 1024 channels: 64 workgroups of 50 dependent steps each took 15-17 us)
template <typename T, int CGv = CG, int RPv = RP>
__global__ __launch_bounds__(256) void seg_mean_std_kernel(const T* x, int ld, int col0, int Tn, int C,
                                                           int want_std, float eps, float* out) {
  static_assert(CGv * RPv == 256, "one workgroup");
  __shared__ f32x4 part[RPv * CGv];
  const int b = blockIdx.y;
  const int cg = threadIdx.x % CGv, rp = threadIdx.x / CGv;
  const int c = (blockIdx.x * CGv + cg) * 4;
  const bool ok = c < C;
  const T* base = x + (size_t)b * Tn * ld + col0 + (ok ? c : 0);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (ok)
    for (int t = rp; t < Tn; t += RPv) s += ld4(base + (size_t)t * ld);
  s = combine_sum<CGv, RPv>(part, rp, cg, s);
  const float invT = 1.0f / (float)Tn;
  const f32x4 mean = s * invT;
  const int ostride = want_std ? 2 * C : C;
  if (ok && rp == 0) st4(out + (size_t)b * ostride + c, mean);
  if (!want_std) return;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (ok)
    for (int t = rp; t < Tn; t += RPv) {
      const f32x4 d = ld4(base + (size_t)t * ld) - mean;
      v += d * d;
    }
  v = combine_sum<CGv, RPv>(part, rp, cg, v);
  if (ok && rp == 0) {
    f32x4 sd;
#pragma unroll
    for (int e = 0; e < 4; ++e) sd[e] = sqrtf(fmaxf(v[e] * invT, eps));
    st4(out + (size_t)b * ostride + C + c, sd);
  }
}

// y = x * gate[segment] + res, 16 bytes per lane, no 64-bit index arithmetic: a workgroup walks whole
// rows (grid-stride over row groups), a thread keeps its channel group for all of them, the
// segment index costs one 32-bit division per row.
// ys (f32 only, may be null): a second, SD_DT_SPLIT16 copy of the result (hi = f16(v), lo = f16(v - hi), interleaved per 32
// channels) at value column s_col0 of rows of lds value columns: what the split16x3 wide convs read, written here so that
// they need no separate pack pass over the MFA input.
template <typename T>
__global__ __launch_bounds__(256) void se_scale_residual_kernel(const T* x, int ldx, const float* gate,
                                                                const T* res, int ldr, int r_col0,
                                                                T* y, int ldy, int y_col0,
                                                                int M, int Tn, int C, _Float16* ys = nullptr, int lds = 0, int s_col0 = 0,
                                                                const _Float16* rsp = nullptr, int ld_rsp = 0, int rsp_col0 = 0, int write_y = 1) {
  constexpr int VEC = 16 / sizeof(T);                 // 4 f32 or 8 f16 channels per lane
  typedef T vec_t __attribute__((ext_vector_type(16 / sizeof(T))));
  const int groups = C / VEC;                          // channel groups per row
  const int gpr = groups < 256 ? groups : 256;         // threads used per row
  const int rows_per_pass = 256 / gpr;
  const int tr = threadIdx.x / gpr, tg = threadIdx.x % gpr;
  if (tr >= rows_per_pass) return;
  for (int m = blockIdx.x * rows_per_pass + tr; m < M; m += gridDim.x * rows_per_pass) {
    const float* g = gate + (size_t)(m / Tn) * C;
    const T* xr = x + (size_t)m * ldx;
    const T* rr = res + (size_t)m * ldr + r_col0;
    T* yr = y + (size_t)m * ldy + y_col0;
    for (int gq = tg; gq < groups; gq += gpr) {
      const int c = gq * VEC;
      const vec_t xv = *reinterpret_cast<const vec_t*>(xr + c);
      vec_t rv;
      if constexpr (sizeof(T) == 4) {
        if (rsp) {      // the shortcut as the SD_DT_SPLIT16 copy a previous call wrote: hi + lo (the value the split convs see)
          typedef _Float16 h4r __attribute__((ext_vector_type(4)));
          const int rc = rsp_col0 + c;
          const _Float16* d = rsp + (size_t)m * 2 * ld_rsp + 64 * (rc / 32) + (rc % 32);
          const h4r hi = *reinterpret_cast<const h4r*>(d), lo = *reinterpret_cast<const h4r*>(d + 32);
#pragma unroll
          for (int e = 0; e < 4; ++e) rv[e] = (float)hi[e] + (float)lo[e];
        } else {
          rv = *reinterpret_cast<const vec_t*>(rr + c);
        }
      } else {
        rv = *reinterpret_cast<const vec_t*>(rr + c);
      }
      vec_t o;
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = (T)((float)xv[e] * g[c + e] + (float)rv[e]);
      if (write_y) *reinterpret_cast<vec_t*>(yr + c) = o;
      if constexpr (sizeof(T) == 4) {
        if (ys) {
          typedef _Float16 h4v __attribute__((ext_vector_type(4)));
          h4v hi, lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float w = sd_split16_clamp((float)o[e]);
            hi[e] = (_Float16)w;
            lo[e] = (_Float16)(w - (float)hi[e]);
          }
          const int sc = s_col0 + c;
          _Float16* d = ys + (size_t)m * 2 * lds + 64 * (sc / 32) + (sc % 32);
          *reinterpret_cast<h4v*>(d) = hi;
          *reinterpret_cast<h4v*>(d + 32) = lo;
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void asp_pool_kernel(const T* logit, int ldl, const T* h, int ldh,
                                                       int Tn, int C, float eps, float* out) {
  __shared__ f32x4 part[RP * CG];
  const int b = blockIdx.y;
  const int cg = threadIdx.x & (CG - 1), rp = threadIdx.x >> 6;
  const int c = (blockIdx.x * CG + cg) * 4;
  const bool ok = c < C;
  const T* lb = logit + (size_t)b * Tn * ldl + (ok ? c : 0);
  const T* hb = h + (size_t)b * Tn * ldh + (ok ? c : 0);
  const float ninf = -INFINITY;
  f32x4 mx = {ninf, ninf, ninf, ninf};
  if (ok)
    for (int t = rp; t < Tn; t += RP) {
      const f32x4 l = ld4(lb + (size_t)t * ldl);
#pragma unroll
      for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], l[e]);
    }
  mx = combine_max(part, rp, cg, mx);
  f32x4 den = {0.f, 0.f, 0.f, 0.f}, num = {0.f, 0.f, 0.f, 0.f};
  if (ok)
    for (int t = rp; t < Tn; t += RP) {
      const f32x4 l = ld4(lb + (size_t)t * ldl);
      const f32x4 hv = ld4(hb + (size_t)t * ldh);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float w = expf(l[e] - mx[e]);
        den[e] += w;
        num[e] += w * hv[e];
      }
    }
  den = combine_sum(part, rp, cg, den);
  num = combine_sum(part, rp, cg, num);
  const f32x4 mu = num / den;
  f32x4 var = {0.f, 0.f, 0.f, 0.f};
  if (ok)
    for (int t = rp; t < Tn; t += RP) {
      const f32x4 l = ld4(lb + (size_t)t * ldl);
      const f32x4 hv = ld4(hb + (size_t)t * ldh);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float w = expf(l[e] - mx[e]);
        const float d = hv[e] - mu[e];
        var[e] += w * d * d;
      }
    }
  var = combine_sum(part, rp, cg, var);
  if (ok && rp == 0) {
    f32x4 sd;
#pragma unroll
    for (int e = 0; e < 4; ++e) sd[e] = sqrtf(fmaxf(var[e] / den[e], eps));
    st4(out + (size_t)b * 2 * C + c, mu);
    st4(out + (size_t)b * 2 * C + C + c, sd);
  }
}


// Attentive statistics pooling with the (segment, channel-tile) block of logits and h resident in
// LDS: one pass over HBM instead of three (the streaming kernel above measured 12.6 GB of traffic
// per 1024-segment launch against 5 GB algorithmic).  The tile is 128 bytes of channels wide (32
// f32 or 64 f16 channels: whole cache lines per row) and kept in the storage type; 256 threads =
// APC channels x (256 / APC) row phases; the three reductions (max, sum / weighted sum, weighted
// variance) are combined through LDS in a fixed order.  Longer segments (T * 256 bytes above the
// LDS budget) take the streaming kernel.
template <typename T> struct AspTile { static constexpr int APC = 32; };
template <> struct AspTile<_Float16> { static constexpr int APC = 64; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(_Float16 v) { return (float)v; }

template <typename T>
__global__ __launch_bounds__(256) void asp_pool_lds_kernel(const T* logit, int ldl, const T* h, int ldh,
                                                           int Tn, int C, float eps, float* out) {
  constexpr int APC = AspTile<T>::APC;
  constexpr int APR = 256 / APC;
  constexpr int VEC = 16 / sizeof(T);          // elements per 16-byte piece
  constexpr int TPR = APC / VEC;               // threads per row while staging (8)
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];
  T* sl = reinterpret_cast<T*>(sm_raw);                        // [Tn][APC] logits
  T* sh = sl + (size_t)Tn * APC;                               // [Tn][APC] h
  float* red = reinterpret_cast<float*>(sm_raw + (((size_t)2 * Tn * APC * sizeof(T) + 15) & ~(size_t)15));  // [2][APR][APC]
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * APC;
  const int tid = threadIdx.x;
  {
    typedef T vec_t __attribute__((ext_vector_type(16 / sizeof(T))));
    const int q = (tid % TPR) * VEC, r = tid / TPR;
    const T* lb = logit + (size_t)b * Tn * ldl + c0 + q;
    const T* hb = h + (size_t)b * Tn * ldh + c0 + q;
    for (int t = r; t < Tn; t += 256 / TPR) {
      *reinterpret_cast<vec_t*>(sl + t * APC + q) = *reinterpret_cast<const vec_t*>(lb + (size_t)t * ldl);
      *reinterpret_cast<vec_t*>(sh + t * APC + q) = *reinterpret_cast<const vec_t*>(hb + (size_t)t * ldh);
    }
  }
  __syncthreads();
  const int c = tid % APC, rp = tid / APC;
  auto combine = [&](float v, float* buf, bool is_max) {
    buf[rp * APC + c] = v;
    __syncthreads();
    float s = buf[c];
#pragma unroll
    for (int k = 1; k < APR; ++k) s = is_max ? fmaxf(s, buf[k * APC + c]) : s + buf[k * APC + c];
    __syncthreads();
    return s;
  };
  float mx = -INFINITY;
  for (int t = rp; t < Tn; t += APR) mx = fmaxf(mx, to_f32(sl[t * APC + c]));
  mx = combine(mx, red, true);
  float den = 0.f, num = 0.f;
  for (int t = rp; t < Tn; t += APR) {
    const float w = expf(to_f32(sl[t * APC + c]) - mx);
    den += w;
    num += w * to_f32(sh[t * APC + c]);
  }
  den = combine(den, red, false);
  num = combine(num, red + APR * APC, false);
  const float mu = num / den;
  float var = 0.f;
  for (int t = rp; t < Tn; t += APR) {
    const float w = expf(to_f32(sl[t * APC + c]) - mx);
    const float d = to_f32(sh[t * APC + c]) - mu;
    var += w * d * d;
  }
  var = combine(var, red, false);
  if (rp == 0) {
    out[(size_t)b * 2 * C + c0 + c] = mu;
    out[(size_t)b * 2 * C + C + c0 + c] = sqrtf(fmaxf(var / den, eps));
  }
}

// one wave per row
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* x, int ldx, int N, int D, float eps_add,
                                                          int zero_guard, float* xn, int ldo) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= N) return;
  const float* xr = x + (size_t)row * ldx;
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) ss += xr[d] * xr[d];
  ss = sd_wave_sum(ss);
  float nrm = sqrtf(ss);
  if (zero_guard && nrm == 0.f) nrm = 1.f;
  nrm += eps_add;
  float* o = xn + (size_t)row * ldo;
  for (int d = lane; d < ldo; d += 64) o[d] = d < D ? xr[d] / nrm : 0.f;
}

__global__ __launch_bounds__(256) void adjacent_cosine_kernel(const float* x, int ldx, int N, int D, float eps, float* sims) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= N - 1) return;
  const float* a = x + (size_t)i * ldx;
  const float* b = a + ldx;
  float dot = 0.f, na = 0.f, nb = 0.f;
  for (int d = lane; d < D; d += 64) {
    dot += a[d] * b[d];
    na += a[d] * a[d];
    nb += b[d] * b[d];
  }
  dot = sd_wave_sum(dot); na = sd_wave_sum(na); nb = sd_wave_sum(nb);
  if (lane == 0) sims[i] = dot / (sqrtf(na) * sqrtf(nb) + eps);
}

__global__ __launch_bounds__(256) void sim_argmax_kernel(const float* w, int ldw, int N, int D, const float* c, int ldc,
                                                         int K, int32_t* best, float* score) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= N) return;
  const float* wr = w + (size_t)i * ldw;
  float bs = -INFINITY;
  int bk = 0;
  for (int k = 0; k < K; ++k) {
    const float* cr = c + (size_t)k * ldc;
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) dot += wr[d] * cr[d];
    dot = sd_wave_sum(dot);
    if (dot > bs) { bs = dot; bk = k; }
  }
  if (lane == 0) {
    best[i] = bk;
    if (score) score[i] = bs;
  }
}

int check_cl_dt(const char* fn, const void* x, int dtype, int ld, int col0, int C) {
  SD_CHECK_ARG(x != nullptr, "%s: null input", fn);
  SD_CHECK_ARG(dtype == SD_DT_F32 || dtype == SD_DT_F16, "%s: bad dtype %d", fn, dtype);
  SD_CHECK_ARG(C > 0 && C % 4 == 0 && ld % 4 == 0 && col0 % 4 == 0 && col0 >= 0 && col0 + C <= ld,
               "%s: C=%d ld=%d col0=%d must be multiples of 4 with the slice inside the row", fn, C, ld, col0);
  SD_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & (dtype == SD_DT_F16 ? 7u : 15u)) == 0, "%s: input is not aligned for 4-channel accesses", fn);
  return SD_OK;
}
// Column statistics left by the conv epilogue (sd_conv_args.colstat) -> per-segment mean (and std).
// colstat [units][6][C]: sums of (y - pivot) and (y - pivot)^2 over each tile of U rows (128; 80 / 96 / 112 from the variable-height
// conv kernel), split at the segment boundaries inside the tile (up to three segments: T >= U / 2) (a trailing partial tile counts its
// existing rows).  Thread = (segment, channel); the tiles of a segment are added in ascending order.
template <typename T>
__global__ __launch_bounds__(256) void colstat_finish_kernel(const float* __restrict__ cs, const float* __restrict__ pivot,
                                                             const T* __restrict__ y, int ldy, int Tn, int C, int M,
                                                             int want_std, float eps, float* __restrict__ out, int U) {
  const int cblocks = (C + 255) / 256;
  const int c = (blockIdx.x % cblocks) * 256 + threadIdx.x;
  const int b = blockIdx.x / cblocks;
  if (c >= C) return;
  const int r0 = b * Tn, r1 = r0 + Tn;            // rows of this segment
  const float pv = pivot ? pivot[c] : 0.f;
  float s = 0.f, q = 0.f;
  for (int u = r0 / U; u * U < r1; ++u) {
    const int part = b - (u * U) / Tn;              // this segment is the tile's first, second or third (T >= U / 2)
    const float* t = cs + (size_t)u * 6 * C + c;
    s += t[(size_t)part * C];
    q += t[(size_t)(3 + part) * C];
  }
  const float inv = 1.f / (float)Tn;
  const float m1 = s * inv;
  out[(size_t)b * (want_std ? 2 : 1) * C + c] = pv + m1;
  if (want_std) out[(size_t)b * 2 * C + C + c] = sqrtf(fmaxf(q * inv - m1 * m1, eps));
}

}  // namespace

// f32 -> f16 (round to nearest even), n % 8 == 0 elements, both 16-byte aligned: the features in front of the f16 stem
__global__ __launch_bounds__(256) void cast_f32_f16_kernel(const float* __restrict__ x, long n8, _Float16* __restrict__ y) {
  typedef _Float16 h8v __attribute__((ext_vector_type(8)));
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
    h8v o;
#pragma unroll
    for (int k = 0; k < 4; ++k) { o[k] = (_Float16)a[k]; o[4 + k] = (_Float16)b[k]; }
    reinterpret_cast<h8v*>(y)[i] = o;
  }
}

int sd_cast_f32_f16(const float* x, long n, void* y, sd_stream_t stream) {
  SD_CHECK_ARG(n >= 0 && n % 8 == 0 && sd_aligned16(x) && sd_aligned16(y), "sd_cast_f32_f16: n=%ld must be a multiple of 8, pointers 16-byte aligned", n);
  if (n == 0) return SD_OK;
  const long n8 = n / 8;
  const long blocks = (n8 + 255) / 256;
  hipLaunchKernelGGL(cast_f32_f16_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, static_cast<hipStream_t>(stream), x, n8,
                     static_cast<_Float16*>(y));
  SD_CHECK_LAUNCH("cast_f32_f16_kernel");
  return SD_OK;
}

extern "C" int sd_seg_mean_std_dt(const void* x, int x_dtype, int ld, int col0, int B, int T, int C, int want_std, float eps,
                                  float* out, sd_stream_t stream) {
  if (int e = check_cl_dt("sd_seg_mean_std_dt", x, x_dtype, ld, col0, C)) return e;
  SD_CHECK_ARG(B > 0 && T > 0 && out && sd_aligned16(out), "sd_seg_mean_std_dt: B=%d T=%d / null or unaligned output", B, T);
  dim3 grid((C / 4 + CG - 1) / CG, B);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if ((long)grid.x * B < 256) {           // a small batch: 16 channel groups x 16 row phases per workgroup
    dim3 g16((C / 4 + 15) / 16, B);
    if (x_dtype == SD_DT_F16)
      hipLaunchKernelGGL((seg_mean_std_kernel<_Float16, 16, 16>), g16, dim3(256), 0, s, static_cast<const _Float16*>(x), ld, col0, T, C, want_std, eps, out);
    else
      hipLaunchKernelGGL((seg_mean_std_kernel<float, 16, 16>), g16, dim3(256), 0, s, static_cast<const float*>(x), ld, col0, T, C, want_std, eps, out);
  } else if (x_dtype == SD_DT_F16)
    hipLaunchKernelGGL(seg_mean_std_kernel<_Float16>, grid, dim3(256), 0, s, static_cast<const _Float16*>(x), ld, col0, T, C, want_std, eps, out);
  else
    hipLaunchKernelGGL(seg_mean_std_kernel<float>, grid, dim3(256), 0, s, static_cast<const float*>(x), ld, col0, T, C, want_std, eps, out);
  SD_CHECK_LAUNCH("seg_mean_std_kernel");
  return SD_OK;
}

extern "C" int sd_seg_mean_f32(const float* x, int ld, int col0, int B, int T, int C, float* mean, sd_stream_t stream) {
  return sd_seg_mean_std_dt(x, SD_DT_F32, ld, col0, B, T, C, 0, 0.f, mean, stream);
}

extern "C" int sd_seg_mean_std_f32(const float* x, int ld, int col0, int B, int T, int C, float eps, float* stats,
                                   sd_stream_t stream) {
  return sd_seg_mean_std_dt(x, SD_DT_F32, ld, col0, B, T, C, 1, eps, stats, stream);
}

extern "C" int sd_se_scale_residual_dt(const void* x, int ldx, const float* gate, const void* res, int ldr, int r_col0,
                                       void* y, int ldy, int y_col0, int B, int T, int C, int dtype, sd_stream_t stream) {
  return sd_se_scale_residual_split(x, ldx, gate, res, ldr, r_col0, y, ldy, y_col0, B, T, C, dtype, nullptr, 0, 0, stream, nullptr, 0, 0, 1);
}

// library-internal: the same, plus (f32 only) an SD_DT_SPLIT16 copy of the result at value column s_col0 of ys [B*T][lds]
int sd_se_scale_residual_split(const void* x, int ldx, const float* gate, const void* res, int ldr, int r_col0,
                               void* y, int ldy, int y_col0, int B, int T, int C, int dtype, void* ys, int lds, int s_col0,
                               sd_stream_t stream, const void* res_split, int ld_rs, int rs_col0, int write_y) {
  if (res_split || !write_y)      // (f32-split16x3 schedule: the shortcut read from the split copy, the f32 result not written)
    SD_CHECK_ARG(dtype == SD_DT_F32 && ys && (!res_split || (ld_rs % 32 == 0 && rs_col0 % 4 == 0 && rs_col0 + C <= ld_rs && sd_aligned16(res_split))),
                 "sd_se_scale_residual: a split shortcut / no f32 result need f32 activations and the split copy of the result");
  if (ys) SD_CHECK_ARG(dtype == SD_DT_F32 && lds % 32 == 0 && s_col0 % 4 == 0 && s_col0 + C <= lds && sd_aligned16(ys),
                       "sd_se_scale_residual: the split copy needs f32 activations, lds %% 32 == 0, an aligned slice inside the row");
  if (int e = check_cl_dt("sd_se_scale_residual(x)", x, dtype, ldx, 0, C)) return e;
  if (int e = check_cl_dt("sd_se_scale_residual(res)", res, dtype, ldr, r_col0, C)) return e;
  if (int e = check_cl_dt("sd_se_scale_residual(y)", y, dtype, ldy, y_col0, C)) return e;
  SD_CHECK_ARG(gate && sd_aligned16(gate) && B > 0 && T > 0, "sd_se_scale_residual: bad gate / B / T");
  const int M = B * T;
  const int vec = dtype == SD_DT_F16 ? 8 : 4;
  SD_CHECK_ARG(C % vec == 0 && ldx % vec == 0 && ldr % vec == 0 && ldy % vec == 0 && r_col0 % vec == 0 && y_col0 % vec == 0,
               "sd_se_scale_residual: C / strides / column offsets must be multiples of %d", vec);
  SD_CHECK_ARG(sd_aligned16(x) && sd_aligned16(res) && sd_aligned16(y), "sd_se_scale_residual: x / res / y must be 16-byte aligned");
  const int groups = C / vec;
  const int rpp = 256 / (groups < 256 ? groups : 256);
  long blocks = ((long)M + rpp - 1) / rpp;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == SD_DT_F16)
    hipLaunchKernelGGL(se_scale_residual_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const _Float16*>(x), ldx, gate,
                       static_cast<const _Float16*>(res), ldr, r_col0, static_cast<_Float16*>(y), ldy, y_col0, M, T, C);
  else
    hipLaunchKernelGGL(se_scale_residual_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const float*>(x), ldx, gate,
                       static_cast<const float*>(res), ldr, r_col0, static_cast<float*>(y), ldy, y_col0, M, T, C,
                       static_cast<_Float16*>(ys), lds, s_col0, static_cast<const _Float16*>(res_split), ld_rs, rs_col0, write_y);
  SD_CHECK_LAUNCH("se_scale_residual_kernel");
  return SD_OK;
}

extern "C" int sd_se_scale_residual_f32(const float* x, int ldx, const float* gate, const float* res, int ldr, int r_col0,
                                        float* y, int ldy, int y_col0, int B, int T, int C, sd_stream_t stream) {
  return sd_se_scale_residual_dt(x, ldx, gate, res, ldr, r_col0, y, ldy, y_col0, B, T, C, SD_DT_F32, stream);
}

extern "C" size_t sd_colstat_floats(int M, int cout) {
  if (M <= 0 || cout <= 0) return 0;
  return (((size_t)M + 127) / 128) * 6 * (size_t)cout;
}

extern "C" int sd_colstat_finish_dt(const float* colstat, const float* pivot, const void* y, int y_dtype, int ldy, int y_col0,
                                    int B, int T, int C, int want_std, float eps, float* out, sd_stream_t stream) {
  return sd_colstat_finish_rows(colstat, pivot, y, y_dtype, ldy, y_col0, B, T, C, want_std, eps, out, 128, stream);
}

int sd_colstat_finish_rows(const float* colstat, const float* pivot, const void* y, int y_dtype, int ldy, int y_col0,
                           int B, int T, int C, int want_std, float eps, float* out, int unit_rows, sd_stream_t stream) {
  SD_CHECK_ARG(colstat && y && out, "sd_colstat_finish_dt: null pointer");
  // (units of 80 rows carry two parts per tile, the others three: sd_conv_gemm.hip)
  SD_CHECK_ARG(unit_rows > 0 && unit_rows <= 128 && (unit_rows == 80 ? T >= 80 : 2 * T >= unit_rows), "sd_colstat_finish: unit of %d rows with T=%d", unit_rows, T);
  SD_CHECK_ARG(y_dtype == SD_DT_F32 || y_dtype == SD_DT_F16, "sd_colstat_finish_dt: y_dtype=%d", y_dtype);
  SD_CHECK_ARG(B >= 0 && T >= 64 && C > 0, "sd_colstat_finish_dt: B=%d T=%d (>= 64) C=%d", B, T, C);
  SD_CHECK_ARG((long)B * T < (1L << 31) && y_col0 >= 0 && y_col0 + C <= ldy, "sd_colstat_finish_dt: bad shape");
  if (B == 0) return SD_OK;
  SD_CHECK_ARG((long)B * ((C + 255) / 256) < (1L << 31), "sd_colstat_finish_dt: grid too large");
  const dim3 grid((unsigned)((long)B * ((C + 255) / 256)));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (y_dtype == SD_DT_F16)
    hipLaunchKernelGGL(colstat_finish_kernel<_Float16>, grid, dim3(256), 0, s, colstat, pivot, static_cast<const _Float16*>(y) + y_col0, ldy, T, C,
                       B * T, want_std, eps, out, unit_rows);
  else
    hipLaunchKernelGGL(colstat_finish_kernel<float>, grid, dim3(256), 0, s, colstat, pivot, static_cast<const float*>(y) + y_col0, ldy, T, C,
                       B * T, want_std, eps, out, unit_rows);
  SD_CHECK_LAUNCH("colstat_finish_kernel");
  return SD_OK;
}

extern "C" int sd_asp_pool_dt(const void* logit, int ldl, const void* h, int dtype, int ldh, int B, int T, int C, float eps,
                              float* out, sd_stream_t stream) {
  if (int e = check_cl_dt("sd_asp_pool(logit)", logit, dtype, ldl, 0, C)) return e;
  if (int e = check_cl_dt("sd_asp_pool(h)", h, dtype, ldh, 0, C)) return e;
  SD_CHECK_ARG(B > 0 && T > 0 && out && sd_aligned16(out), "sd_asp_pool: B=%d T=%d / null or unaligned output", B, T);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool half = dtype == SD_DT_F16;
  const int apc = half ? AspTile<_Float16>::APC : AspTile<float>::APC;
  const size_t lds = (((size_t)2 * T * 128 + 15) & ~(size_t)15) + (size_t)2 * 256 * sizeof(float);   // 128-byte tile rows
  const _Float16* lh = static_cast<const _Float16*>(logit); const _Float16* hh = static_cast<const _Float16*>(h);
  const float* lf = static_cast<const float*>(logit); const float* hf = static_cast<const float*>(h);
  if (C % apc == 0 && lds <= 64 * 1024) {
    dim3 g2(C / apc, B);
    if (half) hipLaunchKernelGGL(asp_pool_lds_kernel<_Float16>, g2, dim3(256), lds, s, lh, ldl, hh, ldh, T, C, eps, out);
    else hipLaunchKernelGGL(asp_pool_lds_kernel<float>, g2, dim3(256), lds, s, lf, ldl, hf, ldh, T, C, eps, out);
    SD_CHECK_LAUNCH("asp_pool_lds_kernel");
    return SD_OK;
  }
  dim3 grid((C / 4 + CG - 1) / CG, B);
  if (half) hipLaunchKernelGGL(asp_pool_kernel<_Float16>, grid, dim3(256), 0, s, lh, ldl, hh, ldh, T, C, eps, out);
  else hipLaunchKernelGGL(asp_pool_kernel<float>, grid, dim3(256), 0, s, lf, ldl, hf, ldh, T, C, eps, out);
  SD_CHECK_LAUNCH("asp_pool_kernel");
  return SD_OK;
}

extern "C" int sd_asp_pool_f32(const float* logit, int ldl, const float* h, int ldh, int B, int T, int C, float eps,
                               float* out, sd_stream_t stream) {
  return sd_asp_pool_dt(logit, ldl, h, SD_DT_F32, ldh, B, T, C, eps, out, stream);
}

extern "C" int sd_l2norm_rows_f32(const float* x, int ldx, int N, int D, float eps_add, int sklearn_zero_guard,
                                  float* xn, int ldo, sd_stream_t stream) {
  SD_CHECK_ARG(N >= 0 && D > 0 && ldx >= D && ldo >= D, "sd_l2norm_rows_f32: N=%d D=%d ldx=%d ldo=%d", N, D, ldx, ldo);
  if (N == 0) return SD_OK;
  SD_CHECK_ARG(x && xn, "sd_l2norm_rows_f32: null pointer");
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3((N + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                     x, ldx, N, D, eps_add, sklearn_zero_guard, xn, ldo);
  SD_CHECK_LAUNCH("l2norm_rows_kernel");
  return SD_OK;
}

extern "C" int sd_adjacent_cosine_f32(const float* x, int ldx, int N, int D, float eps, float* sims, sd_stream_t stream) {
  SD_CHECK_ARG(N >= 0 && D > 0 && ldx >= D, "sd_adjacent_cosine_f32: N=%d D=%d ldx=%d", N, D, ldx);
  if (N < 2) return SD_OK;
  SD_CHECK_ARG(x && sims, "sd_adjacent_cosine_f32: null pointer");
  hipLaunchKernelGGL(adjacent_cosine_kernel, dim3((N - 1 + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                     x, ldx, N, D, eps, sims);
  SD_CHECK_LAUNCH("adjacent_cosine_kernel");
  return SD_OK;
}

extern "C" int sd_sim_argmax_f32(const float* w, int ldw, int N, int D, const float* c, int ldc, int K,
                                 int32_t* best, float* score, sd_stream_t stream) {
  SD_CHECK_ARG(N >= 0 && D > 0 && K > 0 && ldw >= D && ldc >= D, "sd_sim_argmax_f32: N=%d D=%d K=%d", N, D, K);
  if (N == 0) return SD_OK;
  SD_CHECK_ARG(w && c && best, "sd_sim_argmax_f32: null pointer");
  hipLaunchKernelGGL(sim_argmax_kernel, dim3((N + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                     w, ldw, N, D, c, ldc, K, best, score);
  SD_CHECK_LAUNCH("sim_argmax_kernel");
  return SD_OK;
}
