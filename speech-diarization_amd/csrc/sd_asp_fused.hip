// Attention logits + attentive statistics pooling in one kernel (f16 variant first, f32 variant below).
//
// speechbrain's AttentiveStatisticsPooling ends with  Conv1d(att -> C, k=1)  -> softmax over time
// -> weighted mean / std of h (reached from [REF speech_encode.py:77]).
// Run as separate operators that costs a [B*T][C] logits tensor written and read back (12 GB per
// 5000 segments at C = 3072) around a K = 128 GEMM whose tiles are all epilogue.
//
// Here a workgroup owns (segment, 256 channels) and splits TIME over its 4 waves, flash-attention
// style.  Each wave keeps the MFMA B fragments of its <= 64 frames of the attention activations
// a1 [T][128] in registers for the whole kernel (loaded once, straight from global memory), and
// walks the 256 channels in groups of 16: logits tile = W[16 ch][128] x a1^T on
// v_mfma_f32_16x16x32_f16 with CHANNELS as MFMA rows and TIME as MFMA columns, so a lane holds 4
// consecutive channels of one frame per tile -- exactly one 8-byte load of h.  Logits never leave
// the accumulators.  Per channel the wave reduces (max, sum w, sum w h, sum w h^2) over its frames
// with 16-lane DPP reductions; the four waves' partial results are merged through LDS with the
// usual running-max rescaling, and only [mu | sd] (2 C floats per segment) is written.
//
// The conv's bias is not needed: softmax over time is invariant to a per-channel shift.
// var = E_w[h^2] - mu^2 in f32 (h has f16 precision here, 11 bits; the f32 cancellation error is
// 2^-24 relative to E[h^2]).
#include "sd_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int AK = 128;       // attention channels = K of the logits GEMM
constexpr int WLD = AK + 8;   // LDS row stride of the weight tile in halves (272 B: conflict-free ds_read_b128)
constexpr int CPB = 256;      // channels per workgroup
constexpr int NG = CPB / 16;  // channel groups of 16 (one MFMA row tile)

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}

// reduction over the 16 lanes of a DPP row; every lane of the row ends up with the result
template <bool MAX>
__device__ __forceinline__ float red16(float v) {
#define SD_RED_OP(a, b) (MAX ? fmaxf((a), (b)) : (a) + (b))
  v = SD_RED_OP(v, dpp_mov<0xB1>(v));    // quad_perm [1,0,3,2]
  v = SD_RED_OP(v, dpp_mov<0x4E>(v));    // quad_perm [2,3,0,1]
  v = SD_RED_OP(v, dpp_mov<0x141>(v));   // row_half_mirror
  v = SD_RED_OP(v, dpp_mov<0x140>(v));   // row_mirror
#undef SD_RED_OP
  return v;
}

template <int TPW>   // 16-frame tiles per wave: T <= 64 * TPW
__global__ __launch_bounds__(256, 3) void asp_attend_pool_f16_kernel(const _Float16* __restrict__ a1, const _Float16* __restrict__ wc,
                                                                     const _Float16* __restrict__ h, int ldh, int Tn, int C,
                                                                     float eps, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) _Float16 sw[];   // [CPB / 2][WLD] weights; reused for the partial statistics
  const int cblocks = C / CPB;
  const int b = blockIdx.x / cblocks, cblk = blockIdx.x % cblocks;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 15, quad = lane >> 4;      // MFMA column (frame) / k group and output row group

  // this wave's frames: B fragments for the whole kernel.  lane (col, quad) holds a1[t0 + 16 j + col][32 ks + 8 quad .. +7]
  const int t0 = wid * TPW * 16;
  h8 af[TPW][AK / 32];
  {
    const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int t = t0 + j * 16 + col;
      const _Float16* ar = a1 + ((size_t)b * Tn + (t < Tn ? t : 0)) * AK + quad * 8;
#pragma unroll
      for (int ks = 0; ks < AK / 32; ++ks) af[j][ks] = t < Tn ? *reinterpret_cast<const h8*>(ar + ks * 32) : z;
    }
  }
  // weight tile -> LDS, one half (128 channels, 34 KB) at a time so that three workgroups fit a CU
  auto stage_w = [&](int half) {
    const _Float16* wb = wc + ((size_t)cblk * CPB + half * (CPB / 2)) * AK;
#pragma unroll 4
    for (int p = tid; p < (CPB / 2) * (AK / 8); p += 256) {
      const int row = p / (AK / 8), q = (p % (AK / 8)) * 8;
      *reinterpret_cast<h8*>(sw + row * WLD + q) = *reinterpret_cast<const h8*>(wb + (size_t)row * AK + q);
    }
  };
  stage_w(0);
  __syncthreads();

  // h pointer of this lane: frame t0 + col (+16 j), channels cblk*256 + 16 g + 4 quad .. +3
  const _Float16* hl = h + ((size_t)b * Tn + t0 + col) * ldh + (size_t)cblk * CPB + 4 * quad;
  bool live[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) live[j] = t0 + j * 16 + col < Tn;
  const h4 hz = {0, 0, 0, 0};
  auto load_h = [&](int g, h4* dst) {
#pragma unroll
    for (int j = 0; j < TPW; ++j) dst[j] = live[j] ? *reinterpret_cast<const h4*>(hl + (size_t)j * 16 * ldh + g * 16) : hz;
  };

  // per-wave partial statistics of channel 16 g + 4 quad + r: (max, sum w, sum w h, sum w h^2), kept by lane col == g % 16
  float pm[4], pd[4], pn[4], pq[4];
  h4 hv[TPW], hn[TPW];
  load_h(0, hv);
#pragma unroll 1
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) load_h(g + 1, hn);
    if (g == NG / 2) {
      __syncthreads();              // every wave is done with the first half of the weights
      stage_w(1);
      __syncthreads();
    }
    // A fragment: lane (col, quad) holds W[16 g + col][32 ks + 8 quad .. +7]
    const _Float16* wr = sw + ((g % (NG / 2)) * 16 + col) * WLD + quad * 8;
    f32x4 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < AK / 32; ++ks) {
      const h8 wf = *reinterpret_cast<const h8*>(wr + ks * 32);
#pragma unroll
      for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, af[j][ks], acc[j], 0, 0, 0);
    }
    float m[4], d[4], n[4], q[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        acc[j][r] = live[j] ? acc[j][r] : -INFINITY;
        mx = fmaxf(mx, acc[j][r]);
      }
      mx = red16<true>(mx);
      const float base = mx == -INFINITY ? 0.f : mx;     // a wave whose frames are all past T: every w = exp(-inf) = 0
      float dd = 0.f, nn = 0.f, qq = 0.f;
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const float w = __expf(acc[j][r] - base);
        const float hh = (float)hv[j][r];
        const float wh = w * hh;
        dd += w;
        nn += wh;
        qq += wh * hh;
      }
      m[r] = mx; d[r] = red16<false>(dd); n[r] = red16<false>(nn); q[r] = red16<false>(qq);
    }
    if (col == (g & 15)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { pm[r] = m[r]; pd[r] = d[r]; pn[r] = n[r]; pq[r] = q[r]; }
    }
#pragma unroll
    for (int j = 0; j < TPW; ++j) hv[j] = hn[j];
  }

  // merge the four waves: stats[wave][channel][4] in LDS (the weight tile is dead)
  __syncthreads();
  float* st = reinterpret_cast<float*>(sw);
  {
    // lane (col, quad) kept group g = col: channels 16 col + 4 quad + r
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<f32x4*>(st + ((size_t)wid * CPB + 16 * col + 4 * quad + r) * 4) = f32x4{pm[r], pd[r], pn[r], pq[r]};
  }
  __syncthreads();
  {
    const int c = tid;   // 256 threads = 256 channels
    f32x4 s[4];
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      s[w] = *reinterpret_cast<const f32x4*>(st + ((size_t)w * CPB + c) * 4);
      M = fmaxf(M, s[w][0]);
    }
    float den = 0.f, num = 0.f, sq = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float sc = s[w][0] == -INFINITY ? 0.f : __expf(s[w][0] - M);
      den += sc * s[w][1];
      num += sc * s[w][2];
      sq += sc * s[w][3];
    }
    const float mu = num / den;
    const float var = sq / den - mu * mu;
    float* o = out + (size_t)b * 2 * C + (size_t)cblk * CPB + c;
    o[0] = mu;
    o[C] = sqrtf(fmaxf(var, eps));
  }
}

template <int TPW>
int launch_f16(const void* a1, const void* wc, const void* h, int ldh, int B, int T, int C, float eps, float* out, hipStream_t s) {
  auto kern = asp_attend_pool_f16_kernel<TPW>;
  const size_t lds = (size_t)(CPB / 2) * WLD * sizeof(_Float16);
  static_assert((size_t)(CPB / 2) * WLD * sizeof(_Float16) >= (size_t)4 * CPB * 4 * sizeof(float), "statistics must fit in the weight tile");
  SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(kern), (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)((long)B * (C / CPB))), dim3(256), lds, s, static_cast<const _Float16*>(a1),
                     static_cast<const _Float16*>(wc), static_cast<const _Float16*>(h), ldh, T, C, eps, out);
  SD_CHECK_LAUNCH("asp_attend_pool_f16_kernel");
  return SD_OK;
}

// ------------------------------------------------------------------------------------------
// f32 activations: exact-f32 v_mfma_f32_16x16x4_f32 is 16x slower per MAC than the f16 MFMA, so
// this variant is MFMA bound (as the asp.conv it replaces was) and organised the other way
// round: the segment's a1 [T][128] sits in LDS, a workgroup of 8 waves owns (segment, 256 or 512
// channels), and each WAVE owns whole channel groups of 16 with ALL frames of the segment, so
// softmax and the (two-pass) weighted variance stay inside the wave and only 16-lane DPP
// reductions are needed.  Weight fragments come straight from global memory (no reuse between
// waves), fetched for the next group while the softmax of this one runs; h for the group is
// fetched while its MFMAs run.  Same k-permutation
// as the conv kernel: a lane reads 4 consecutive k with one 16-byte access and feeds 4 MFMAs.
// Measured at 5000 segments: 8.05 ms against 14.2 ms for the conv + pooling pair; matrix pipe 66 % busy.
// In-kernel cycle counters, per group and wave: 5.7 k cycles issuing the scattered h / weight loads (16
// lines per instruction), 14.5 k in the MFMA loop (13.3 k of MFMA), 12 k in softmax + store: a wave is in
// its MFMA loop 45 % of the time and two per SIMD do not interleave perfectly.  Tried and dropped: delaying
// one partner by 4-10 k cycles (no change), s_setprio around the MFMA loop (-0.7 %), three waves per SIMD
// (12-wave workgroups; 168 registers mean spills and no h prefetch: 8.5 ms).
constexpr int FLD = AK + 4;   // LDS row stride of the f32 a1 tile in floats (528 B: conflict-free ds_read_b128)

template <int NT>   // 16-frame tiles: T <= 16 * NT
__global__ __launch_bounds__(512, 2) void asp_attend_pool_f32_kernel(const float* __restrict__ a1, const float* __restrict__ wc,
                                                                     const float* __restrict__ h, int ldh, int Tn, int C, int cpb,
                                                                     float eps, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sf[];   // [NT * 16][FLD]
  const int cblocks = C / cpb;
  const int b = blockIdx.x / cblocks, cblk = blockIdx.x % cblocks;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 15, quad = lane >> 4;
  {
    const float* ab = a1 + (size_t)b * Tn * AK;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int p = tid; p < NT * 16 * (AK / 4); p += 512) {
      const int row = p / (AK / 4), q = (p % (AK / 4)) * 4;
      *reinterpret_cast<f32x4*>(sf + row * FLD + q) = row < Tn ? *reinterpret_cast<const f32x4*>(ab + (size_t)row * AK + q) : z;
    }
  }
  __syncthreads();

  const int ngroups = cpb / 16;                       // groups of this workgroup; wave w takes w, w + 8, ...
  const float* hl = h + ((size_t)b * Tn + col) * ldh + (size_t)cblk * cpb + 4 * quad;
  const float* wl = wc + ((size_t)cblk * cpb + col) * AK + 4 * quad;
  const float* al = sf + col * FLD + 4 * quad;
  auto load_w = [&](int g, f32x4* dst) {
#pragma unroll
    for (int ks = 0; ks < AK / 16; ++ks) dst[ks] = *reinterpret_cast<const f32x4*>(wl + (size_t)g * 16 * AK + ks * 16);
  };
  f32x4 wf[AK / 16];
  if (wid < ngroups) load_w(wid, wf);
#pragma unroll 1
  for (int g = wid; g < ngroups; g += 8) {
    // h of this group in the accumulator layout (consumed after the MFMAs): frame 16 j + col, channels 16 g + 4 quad .. +3
    f32x4 hv[NT];
    {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NT; ++j) hv[j] = j * 16 + col < Tn ? *reinterpret_cast<const f32x4*>(hl + (size_t)j * 16 * ldh + g * 16) : z;
    }
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Two frame tiles per step (their MFMAs alternate, so no MFMA waits on its predecessor's
    // 40-cycle result latency) and the next step's two B fragments in flight; the scheduling
    // barriers keep the compiler from hoisting all NT * 8 LDS reads (416 registers) above the MFMAs.
    constexpr int NU = (NT + 1) / 2;
    auto frag = [&](int u, int which) {
      const int ks = u / NU, j = 2 * (u % NU) + which;
      return *reinterpret_cast<const f32x4*>(al + (j < NT ? j : NT - 1) * 16 * FLD + ks * 16);
    };
    f32x4 b0 = frag(0, 0), b1 = frag(0, 1);
#pragma unroll
    for (int u = 0; u < NU * (AK / 16); ++u) {
      const int ks = u / NU, j = 2 * (u % NU);
      f32x4 n0 = b0, n1 = b1;
      if (u + 1 < NU * (AK / 16)) { n0 = frag(u + 1, 0); n1 = frag(u + 1, 1); }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ks][e], b0[e], acc[j], 0, 0, 0);
        if (j + 1 < NT) acc[j + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ks][e], b1[e], acc[j + 1], 0, 0, 0);
      }
      b0 = n0; b1 = n1;
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // the two LDS reads of the next step first ...
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   // ... then this step's MFMAs
      __builtin_amdgcn_sched_barrier(0);
    }
    // the next group's weight fragments load under the softmax phase (wf is dead from here on)
    if (g + 8 < ngroups) load_w(g + 8, wf);
    f32x4 mu4, sd4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[j][r] = j * 16 + col < Tn ? acc[j][r] : -INFINITY;
        mx = fmaxf(mx, acc[j][r]);
      }
      mx = red16<true>(mx);
      float dd = 0.f, nn = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float w = __expf(acc[j][r] - mx);
        acc[j][r] = w;
        dd += w;
        nn += w * hv[j][r];
      }
      dd = red16<false>(dd);
      const float mu = red16<false>(nn) / dd;
      float vv = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float d = hv[j][r] - mu;
        vv += acc[j][r] * d * d;
      }
      vv = red16<false>(vv);
      mu4[r] = mu;
      sd4[r] = sqrtf(fmaxf(vv / dd, eps));
    }
    if (col == 0) {
      float* o = out + (size_t)b * 2 * C + (size_t)cblk * cpb + g * 16 + 4 * quad;
      *reinterpret_cast<f32x4*>(o) = mu4;
      *reinterpret_cast<f32x4*>(o + C) = sd4;
    }
  }
}

template <int NT>
int launch_f32(const void* a1, const void* wc, const void* h, int ldh, int B, int T, int C, float eps, float* out, hipStream_t s) {
  auto kern = asp_attend_pool_f32_kernel<NT>;
  const size_t lds = (size_t)NT * 16 * FLD * sizeof(float);
  const int cpb = C % 512 == 0 ? 512 : 256;
  SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(kern), (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)((long)B * (C / cpb))), dim3(512), lds, s, static_cast<const float*>(a1),
                     static_cast<const float*>(wc), static_cast<const float*>(h), ldh, T, C, cpb, eps, out);
  SD_CHECK_LAUNCH("asp_attend_pool_f32_kernel");
  return SD_OK;
}

}  // namespace

extern "C" int sd_asp_attend_pool_supported(int dtype, int T, int C, int att) {
  return (dtype == SD_DT_F16 || dtype == SD_DT_F32) && att == AK && C > 0 && C % CPB == 0 && T > 0 && T <= 256;
}

extern "C" int sd_asp_attend_pool_dt(const void* a1, const void* wc, const void* h, int dtype, int ldh, int B, int T, int C,
                                     int att, float eps, float* out, sd_stream_t stream) {
  SD_CHECK_ARG(a1 && wc && h && out, "sd_asp_attend_pool_dt: null pointer");
  SD_CHECK_ARG(B >= 0 && (long)B * (C > 0 ? C : 1) < (1L << 31), "sd_asp_attend_pool_dt: B=%d", B);
  if (!sd_asp_attend_pool_supported(dtype, T, C, att))
    return sd_set_error(SD_ERR_UNSUPPORTED, "sd_asp_attend_pool_dt: dtype=%d T=%d C=%d att=%d not covered (att=128, C%%256==0, T<=256)",
                        dtype, T, C, att);
  SD_CHECK_ARG(ldh >= C && ldh % 4 == 0, "sd_asp_attend_pool_dt: ldh=%d (need >= C and a multiple of 4)", ldh);
  SD_CHECK_ARG(sd_aligned16(a1) && sd_aligned16(wc) && sd_aligned16(h) && sd_aligned16(out), "sd_asp_attend_pool_dt: pointers must be 16-byte aligned");
  if (B == 0) return SD_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == SD_DT_F32) {
    if (T <= 64) return launch_f32<4>(a1, wc, h, ldh, B, T, C, eps, out, s);
    if (T <= 128) return launch_f32<8>(a1, wc, h, ldh, B, T, C, eps, out, s);
    if (T <= 208) return launch_f32<13>(a1, wc, h, ldh, B, T, C, eps, out, s);
    return launch_f32<16>(a1, wc, h, ldh, B, T, C, eps, out, s);
  }
  if (T <= 64) return launch_f16<1>(a1, wc, h, ldh, B, T, C, eps, out, s);
  if (T <= 128) return launch_f16<2>(a1, wc, h, ldh, B, T, C, eps, out, s);
  return launch_f16<4>(a1, wc, h, ldh, B, T, C, eps, out, s);
}
