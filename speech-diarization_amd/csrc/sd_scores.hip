// Score normalisation and smoothing on the device: the consumers of the cosine products in the reference's
// `asnorm_scores` [REF diar_diag.py:196-208] and `viterbi_hmm` [REF diar_diag.py:231-247].
//
//   AS-norm: the three cosine products (queries x centres, queries x cohort, centres x cohort) run on the f32 matrix
//   cores through sd_conv1d_cl_f32 (T = 1, the cohort as the weight matrix); here: the per-row statistics of the
//   TOP-K cohort scores (np.sort(...)[:, -k:].mean / .std) without sorting, and the final combination.
//   Top-k statistics: one workgroup per row; an exact radix select over the order-preserving integer image of the
//   floats (4 passes of 8 bits, LDS histogram) finds the k-th largest value and how many values are strictly
//   greater; the top-k multiset is then {x > kth} plus copies of kth.  Mean, then sum of squared deviations (two
//   passes: no E[x^2] - mean^2 cancellation).  A row is read from L2 six times; at the reference's density
//   (~36 k windows x 10 k cohort) that is 1.4 GB x 6 of L2-resident traffic, where np.sort moves the same bytes
//   ~log2(n) = 13 times through host memory.
//
//   Viterbi: K <= 64 states, one wave, lane j = state j.  dp and the transition matrix's two distinct values stay in
//   registers; per step  cand_i = dp_i + logA[i][j]  (f32, like the reference), first maximum wins (np.argmax).
//   Scores and back pointers move through LDS in chunks of 128 steps so that no step waits on global memory; the
//   backtrack walks the back pointers chunk by chunk from LDS.  Sequential by nature: ~T x (K + 8) wave instructions.
#include "sd_common.h"

namespace {

__device__ __forceinline__ unsigned f32_ord(float v) {          // order-preserving map float -> unsigned
  const unsigned b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

constexpr int TK_THREADS = 256;

__global__ __launch_bounds__(TK_THREADS) void topk_mean_std_kernel(const float* __restrict__ x, int ld, int n, int k, float* __restrict__ out) {
  __shared__ unsigned hist[256];
  __shared__ unsigned sel_prefix, sel_remaining;
  __shared__ float red[TK_THREADS / 64];
  __shared__ float bcast[2];
  const int tid = threadIdx.x;
  const float* row = x + (size_t)blockIdx.x * ld;
  if (k > n) k = n;
  // ---- exact k-th largest by radix select on the ordered keys, most significant byte first
  unsigned prefix = 0, remaining = (unsigned)k;      // keys with the chosen high bytes; how many of the top k are still among them
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    const unsigned mask_hi = pass == 0 ? 0u : 0xFFFFFFFFu << (shift + 8);
    for (int i = tid; i < n; i += TK_THREADS) {
      const unsigned key = f32_ord(row[i]);
      if ((key & mask_hi) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned rem = remaining;
      int d = 255;
      for (; d > 0; --d) {                            // walk down from the largest digit
        if (hist[d] >= rem) break;
        rem -= hist[d];
      }
      sel_prefix = prefix | ((unsigned)d << shift);
      sel_remaining = rem;                            // copies still to take among keys with this digit
    }
    __syncthreads();
    prefix = sel_prefix;
    remaining = sel_remaining;
    __syncthreads();
  }
  const unsigned kth = prefix;                        // key of the k-th largest value; `remaining` copies of it belong to the top k
  // ---- mean of the top k
  auto block_sum = [&](float v) {
    v = sd_wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < TK_THREADS / 64; ++w) t += red[w];
    __syncthreads();
    return t;
  };
  float kth_val = 0.f;
  float s = 0.f;
  for (int i = tid; i < n; i += TK_THREADS) {
    const float v = row[i];
    const unsigned key = f32_ord(v);
    if (key > kth) s += v;
    if (key == kth) kth_val = v;
  }
  // every thread that saw the k-th value holds the same float; broadcast it
  if (tid == 0) bcast[0] = 0.f;
  __syncthreads();
  if (f32_ord(kth_val) == kth) bcast[0] = kth_val;     // benign race: identical values
  __syncthreads();
  kth_val = bcast[0];
  const float total = block_sum(s) + (float)remaining * kth_val;
  const float mean = total / (float)k;
  float q = 0.f;
  for (int i = tid; i < n; i += TK_THREADS) {
    const float v = row[i];
    if (f32_ord(v) > kth) q += (v - mean) * (v - mean);
  }
  const float ss = block_sum(q) + (float)remaining * (kth_val - mean) * (kth_val - mean);
  if (tid == 0) {
    out[(size_t)blockIdx.x * 2] = mean;
    out[(size_t)blockIdx.x * 2 + 1] = sqrtf(ss / (float)k);     // population std (numpy default ddof = 0)
  }
}

__global__ void asnorm_combine_kernel(const float* __restrict__ raw, int ld, int nq, int nr, const float* __restrict__ qstat,
                                      const float* __restrict__ rstat, float* __restrict__ out, int ldo) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)nq * nr) return;
  const int i = (int)(idx / nr), j = (int)(idx - (long)i * nr);
  const float v = raw[(size_t)i * ld + j];
  const float zq = (v - qstat[2 * i]) / (qstat[2 * i + 1] + 1e-6f);
  const float zr = (v - rstat[2 * j]) / (rstat[2 * j + 1] + 1e-6f);
  out[(size_t)i * ldo + j] = 0.5f * (zq + zr);
}

constexpr int VT_CHUNK = 128;

__global__ __launch_bounds__(64) void viterbi_kernel(const float* __restrict__ scores, int ld, int T, int K, float log_stay, float log_move,
                                                     unsigned char* __restrict__ back, int* __restrict__ path) {
  __shared__ float s_sc[VT_CHUNK * 64];
  __shared__ unsigned char s_bk[VT_CHUNK * 64];
  const int lane = threadIdx.x;
  const bool act = lane < K;
  float dp = act ? scores[lane] : -INFINITY;           // dp[0] = scores[0]
  for (int t0 = 1; t0 < T; t0 += VT_CHUNK) {
    const int steps = T - t0 < VT_CHUNK ? T - t0 : VT_CHUNK;
    for (int e = lane; e < steps * K; e += 64) {       // stage the chunk's scores [steps][K]
      const int tt = e / K, j = e - tt * K;
      s_sc[tt * K + j] = scores[(size_t)(t0 + tt) * ld + j];
    }
    __syncthreads();
    for (int tt = 0; tt < steps; ++tt) {
      float best = -INFINITY;
      int arg = 0;
      for (int i = 0; i < K; ++i) {
        const float cand = __shfl(dp, i, 64) + (i == lane ? log_stay : log_move);     // f32 add, as dp[t-1][:, None] + logA
        if (cand > best) { best = cand; arg = i; }                                    // first maximum wins (np.argmax)
      }
      if (act) {
        dp = best + s_sc[tt * K + lane];
        s_bk[tt * K + lane] = (unsigned char)arg;
      }
    }
    __syncthreads();
    for (int e = lane; e < steps * K; e += 64) back[(size_t)t0 * K + e] = s_bk[e];
    __syncthreads();
  }
  // ---- last state: first maximum of dp
  float bv = act ? dp : -INFINITY;
  int bi = act ? lane : 64;
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  int cur = bi;
  if (lane == 0) path[T - 1] = cur;
  // ---- backtrack, chunk by chunk from the end: path[t] = back[t + 1][path[t + 1]]
  __threadfence();                                    // the wave re-reads the back pointers it stored through global memory
  for (int hi = T - 1; hi >= 1; hi -= VT_CHUNK) {      // back rows (lo, hi] give path[lo .. hi - 1]
    const int lo = hi - VT_CHUNK > 0 ? hi - VT_CHUNK : 0;
    const int rows = hi - lo;
    __syncthreads();
    for (int e = lane; e < rows * K; e += 64) s_bk[e] = back[(size_t)(lo + 1) * K + e];
    __syncthreads();
    if (lane == 0) {
      for (int t = hi; t > lo; --t) {
        cur = s_bk[(t - lo - 1) * K + cur];
        path[t - 1] = cur;
      }
    }
    cur = __shfl(cur, 0, 64);
  }
}

}  // namespace

extern "C" int sd_topk_mean_std_f32(const float* x, int ld, int rows, int n, int k, float* out, sd_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SD_CHECK_ARG(rows >= 0 && n >= 1 && k >= 1 && ld >= n, "sd_topk_mean_std_f32: rows=%d n=%d k=%d ld=%d", rows, n, k, ld);
  if (rows == 0) return SD_OK;
  SD_CHECK_ARG(x && out, "sd_topk_mean_std_f32: null pointer");
  hipLaunchKernelGGL(topk_mean_std_kernel, dim3((unsigned)rows), dim3(TK_THREADS), 0, stream, x, ld, n, k, out);
  SD_CHECK_LAUNCH("topk_mean_std_kernel");
  return SD_OK;
}

extern "C" int sd_asnorm_combine_f32(const float* raw, int ld, int nq, int nr, const float* qstat, const float* rstat,
                                     float* out, int ldo, sd_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SD_CHECK_ARG(nq >= 0 && nr >= 0 && ld >= nr && ldo >= nr, "sd_asnorm_combine_f32: nq=%d nr=%d ld=%d ldo=%d", nq, nr, ld, ldo);
  if (nq == 0 || nr == 0) return SD_OK;
  SD_CHECK_ARG(raw && qstat && rstat && out, "sd_asnorm_combine_f32: null pointer");
  const long total = (long)nq * nr;
  hipLaunchKernelGGL(asnorm_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, raw, ld, nq, nr, qstat, rstat, out, ldo);
  SD_CHECK_LAUNCH("asnorm_combine_kernel");
  return SD_OK;
}

extern "C" size_t sd_viterbi_workspace_bytes(int T, int K) {
  if (T <= 0 || K <= 0) return 0;
  return ((size_t)T * K + 255) & ~(size_t)255;
}

extern "C" int sd_viterbi_f32(const float* scores, int ld, int T, int K, float log_stay, float log_move,
                              void* ws, size_t ws_bytes, int32_t* path, sd_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SD_CHECK_ARG(T >= 0 && K >= 1 && K <= 64 && ld >= K, "sd_viterbi_f32: T=%d K=%d (1..64) ld=%d", T, K, ld);
  if (T == 0) return SD_OK;
  SD_CHECK_ARG(scores && ws && path, "sd_viterbi_f32: null pointer");
  SD_CHECK_ARG(ws_bytes >= sd_viterbi_workspace_bytes(T, K), "sd_viterbi_f32: workspace %zu < %zu bytes", ws_bytes, sd_viterbi_workspace_bytes(T, K));
  hipLaunchKernelGGL(viterbi_kernel, dim3(1), dim3(64), 0, stream, scores, ld, T, K, log_stay, log_move, static_cast<unsigned char*>(ws), path);
  SD_CHECK_LAUNCH("viterbi_kernel");
  return SD_OK;
}
