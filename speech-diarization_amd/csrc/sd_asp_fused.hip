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
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int AK = 128;       // attention channels = K of the logits GEMM
constexpr int WLD = AK + 8;   // LDS row stride of the weight tile in halves (272 B: conflict-free ds_read_b128)
constexpr int CPB = 256;      // channels per workgroup
constexpr int NG = CPB / 16;  // channel groups of 16 (one MFMA row tile)

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
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

// 16-lane (DPP row) reduction of FOUR registers at once, one instruction per register and step: dependent steps are
// three instructions apart (a DPP read needs two wait states behind the VALU write of its source); every lane of a
// row ends up with the row's result.  (Written out because hipcc pairs up the builtin form into v_mov_b32_dpp x 2 +
// v_pk_add_f32, and canonicalises the operands of a max: 1.5 and 4 instructions per step instead of 1.)
#define SD_DPP_STEP4(op, ctrl)                                              \
  op " %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                  \
  op " %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                  \
  op " %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                  \
  op " %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf\n\t"
#define SD_DPP_RED4(op, a, b, c, d)                                                                              \
  asm volatile("s_nop 1\n\t" SD_DPP_STEP4(op, "quad_perm:[1,0,3,2]") SD_DPP_STEP4(op, "quad_perm:[2,3,0,1]")     \
               SD_DPP_STEP4(op, "row_half_mirror") SD_DPP_STEP4(op, "row_mirror") "s_nop 1"                        \
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

template <int TPW>   // 16-frame tiles per wave: T <= 64 * TPW
__global__ __launch_bounds__(256, 3) void asp_attend_pool_f16_kernel(const _Float16* __restrict__ a1, const _Float16* __restrict__ wc,
                                                                     const _Float16* __restrict__ h, int ldh, int Tn, int C,
                                                                     float eps, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) _Float16 sw[];   // [CPB / 2][WLD] weights, then the partial statistics
  const int cblocks = C / CPB;
  const int b = blockIdx.x / cblocks, cblk = blockIdx.x % cblocks;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 15, quad = lane >> 4;      // MFMA column (frame) / k group and output row group

  // this wave's frames: 16-frame tiles wid, wid + 4, wid + 8, ... (slot j = tile 4 j + wid), so that the tiles past T
  // fall into the LAST slot of some waves, which those waves skip (T = 201: 13 tiles; wave 0 runs 4 slots, the others 3).
  // B fragments for the whole kernel: lane (col, quad) holds a1[t0 + 64 j + col][32 ks + 8 quad .. +7]
  const int t0 = wid * 16;
  const bool full = ((TPW - 1) * 4 + __builtin_amdgcn_readfirstlane(wid)) * 16 < Tn;    // wave-uniform: the last slot has live frames
  h8 af[TPW][AK / 32];
  {
    const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int t = t0 + j * 64 + col;
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

  // h of this lane.  A 16-channel MFMA row tile may be ANY 16 channels, so the tiles are chosen for the loads: groups
  // 2 p and 2 p + 1 share the 32 channels [32 p, 32 p + 32) as  channel(2 p + e, tile row 4 q + r) = 32 p + 8 q + 4 e + r.
  // Lane (frame col, quad) then needs channels 32 p + 8 quad .. + 7 for the pair: ONE 16-byte load per frame and pair,
  // a wave instruction reads 64 contiguous bytes of 16 rows (8-byte loads per group fetched every 128-byte line four
  // times from L2: the L1 is smaller than what the CU's twelve waves have in flight).
  const _Float16* hl = h + ((size_t)b * Tn + t0 + col) * ldh + (size_t)cblk * CPB + 8 * quad;
  bool live[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) live[j] = t0 + j * 64 + col < Tn;
  const h8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
  auto load_h = [&](int p, h8* dst) {
#pragma unroll
#ifdef SD_ASP_NOH
    for (int j = 0; j < TPW; ++j) dst[j] = h8{(_Float16)(float)p, (_Float16)1.f, (_Float16)(float)j, (_Float16)2.f, (_Float16)1.f, (_Float16)0.5f, (_Float16)3.f, (_Float16)2.f};
#else
    for (int j = 0; j < TPW; ++j) dst[j] = live[j] ? *reinterpret_cast<const h8*>(hl + (size_t)j * 64 * ldh + p * 32) : hz;
#endif
  };

  // per-wave partial statistics (max, sum w, sum w h, sum w h^2) per channel -> LDS behind the weights
  float* const st = reinterpret_cast<float*>(sw + (CPB / 2) * WLD);     // [wave][channel][4]
  // frames past T are masked inside the MFMA: the accumulators start at 0 (live) or -inf (dead; a1 is zero there, so
  // the products are finite), and exp2(-inf) = 0 drops them from every sum.  Only the last slot can hold such frames
  // (T > 64 (TPW - 1)).
  f32x4 minit;
  {
    const float v = live[TPW - 1] ? 0.f : -INFINITY;
    minit = f32x4{v, v, v, v};
  }
  constexpr float LOG2E = 1.4426950408889634f;
  // one channel group.  VALU work per (frame, channel) is what bounds this kernel (measured: 2.8 ms with the h loads
  // removed, 1.1 ms with the softmax removed as well), so: masks come out of the MFMA, the exponent is one packed
  // FMA + v_exp_f32, the weighted sums run on packed f32 pairs (channels r, r + 1), and the 16-lane reductions are
  // single DPP instructions (v_max_f32_dpp / v_add_f32_dpp).
  auto group = [&](const int p, const int e, const h8 (&hv)[TPW]) {
    // A fragment: lane (col, quad) holds the weights of tile row col = channel 32 p + 8 (col >> 2) + 4 e + (col & 3)
    const _Float16* wr = sw + ((32 * p + 8 * (col >> 2) + 4 * e + (col & 3)) % (CPB / 2)) * WLD + quad * 8;
    f32x4 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[j] = j == TPW - 1 ? minit : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < AK / 32; ++ks) {
      const h8 wf = *reinterpret_cast<const h8*>(wr + ks * 32);
#pragma unroll
      for (int j = 0; j < (TPW > 1 ? TPW - 1 : 1); ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, af[j][ks], acc[j], 0, 0, 0);
    }
    if (TPW > 1 && full) {          // (skipped: the slot's accumulators stay at -inf and drop out of max and sums)
#pragma unroll
      for (int ks = 0; ks < AK / 32; ++ks) {
        const h8 wf = *reinterpret_cast<const h8*>(wr + ks * 32);
        acc[TPW - 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, af[TPW - 1][ks], acc[TPW - 1], 0, 0, 0);
      }
    }
    float m[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = acc[0][r];
#pragma unroll
      for (int j = 1; j < TPW; ++j) mx = fmaxf(mx, acc[j][r]);
      m[r] = mx;
    }
    SD_DPP_RED4("v_max_f32_dpp", m[0], m[1], m[2], m[3]);     // max over the 16 frames of a DPP row
    float d[4], n[4], q[4];
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
      // a wave whose frames are all past T: max = -inf, every w = exp2(-inf) = 0
      const f32x2 nb = {m[2 * rp] == -INFINITY ? 0.f : -m[2 * rp] * LOG2E, m[2 * rp + 1] == -INFINITY ? 0.f : -m[2 * rp + 1] * LOG2E};
      f32x2 dd = {0.f, 0.f}, nn = {0.f, 0.f}, qq = {0.f, 0.f};
      auto slot = [&](const int j) {
        const f32x2 x = {acc[j][2 * rp], acc[j][2 * rp + 1]};
        const f32x2 ex = x * f32x2{LOG2E, LOG2E} + nb;
        const f32x2 w = {__builtin_amdgcn_exp2f(ex[0]), __builtin_amdgcn_exp2f(ex[1])};
        const f32x2 hh = {(float)hv[j][4 * e + 2 * rp], (float)hv[j][4 * e + 2 * rp + 1]};
        const f32x2 wh = w * hh;
        dd += w;
        nn += wh;
        qq += wh * hh;
      };
#pragma unroll
      for (int j = 0; j < (TPW > 1 ? TPW - 1 : 1); ++j) slot(j);
      if (TPW > 1 && full) slot(TPW - 1);
#pragma unroll
      for (int k = 0; k < 2; ++k) { d[2 * rp + k] = dd[k]; n[2 * rp + k] = nn[k]; q[2 * rp + k] = qq[k]; }
    }
    SD_DPP_RED4("v_add_f32_dpp", d[0], d[1], d[2], d[3]);
    SD_DPP_RED4("v_add_f32_dpp", n[0], n[1], n[2], n[3]);
    SD_DPP_RED4("v_add_f32_dpp", q[0], q[1], q[2], q[3]);
    if (col == 0) {                 // every lane of the row holds the row's results
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<f32x4*>(st + ((size_t)wid * CPB + 32 * p + 8 * quad + 4 * e + r) * 4) = f32x4{m[r], d[r], n[r], q[r]};
    }
  };
  h8 ha[TPW], hb[TPW];
  load_h(0, ha);
#pragma unroll 1
  for (int p = 0; p < NG / 2; p += 2) {     // two pairs of groups per trip: the h registers swap roles instead of being copied
    if (p == NG / 4) {
      __syncthreads();              // every wave is done with the first half of the weights
      stage_w(1);
      __syncthreads();
    }
    load_h(p + 1, hb);
    group(p, 0, ha);
    group(p, 1, ha);
    if (p + 2 < NG / 2) load_h(p + 2, ha);
    group(p + 1, 0, hb);
    group(p + 1, 1, hb);
  }

  // merge the four waves
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
  const size_t lds = (size_t)(CPB / 2) * WLD * sizeof(_Float16) + (size_t)4 * CPB * 4 * sizeof(float);   // 34 + 16 KB: three workgroups per CU
  static_assert(3 * ((size_t)(CPB / 2) * WLD * sizeof(_Float16) + (size_t)4 * CPB * 4 * sizeof(float)) <= 160 * 1024, "three workgroups per CU");
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

// SPLIT (the f32-split16x3 mode; tensors stay f32): the logits product on v_mfma_f32_16x16x32_f16 with every operand value split
// hi + lo and three products per fragment pair (hi.hi + hi.lo + lo.hi, f32 accumulation; 2^-22 relative per product): the a1 tile is
// split ONCE when it is staged (LDS: a hi plane and a lo plane of [frames][128 + 8] halfs), the 16 x 128 weight rows of a group
// are split in registers as they arrive (scaled by 2^8: |w| ~ 0.1, the low halves stay out of the f16 subnormals; the logits
// take 2^-8 back), and a (frame tile, 32 k) pair costs 3 MFMAs of 16 cycles instead of 8 f32 MFMAs of 32.  The accumulator layout
// (lane = frame, 4 channels per lane) is the f32 instruction's, so softmax and pooling below are shared.
constexpr int SLD = AK + 8;   // LDS row stride of a split a1 plane in halfs (272 B)

template <int NT, bool SPLIT = false>   // 16-frame tiles: T <= 16 * NT
__global__ __launch_bounds__(512, 2) void asp_attend_pool_f32_kernel(const float* __restrict__ a1, const float* __restrict__ wc,
                                                                     const float* __restrict__ h, int ldh, int Tn, int C, int cpb,
                                                                     float eps, float* __restrict__ out, float wscale) {
  extern __shared__ __attribute__((aligned(16))) float sf[];   // [NT * 16][FLD]   (SPLIT: two planes of [NT * 16][SLD] halfs)
  const int cblocks = C / cpb;
  const int b = blockIdx.x / cblocks, cblk = blockIdx.x % cblocks;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 15, quad = lane >> 4;
  _Float16* const s_hi = reinterpret_cast<_Float16*>(sf);
  _Float16* const s_lo = s_hi + NT * 16 * SLD;
  {
    const float* ab = a1 + (size_t)b * Tn * AK;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int p = tid; p < NT * 16 * (AK / 4); p += 512) {
      const int row = p / (AK / 4), q = (p % (AK / 4)) * 4;
      const f32x4 v = row < Tn ? *reinterpret_cast<const f32x4*>(ab + (size_t)row * AK + q) : z;
      if constexpr (SPLIT) {
        h4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float c = sd_split16_clamp(v[e]); hi[e] = (_Float16)c; lo[e] = (_Float16)(c - (float)hi[e]); }
        *reinterpret_cast<h4*>(s_hi + row * SLD + q) = hi;
        *reinterpret_cast<h4*>(s_lo + row * SLD + q) = lo;
      } else {
        *reinterpret_cast<f32x4*>(sf + row * FLD + q) = v;
      }
    }
  }
  __syncthreads();

  const int ngroups = cpb / 16;                       // groups of this workgroup; wave w takes w, w + 8, ...
  const float* hl = h + ((size_t)b * Tn + col) * ldh + (size_t)cblk * cpb + 4 * quad;
  // weight fragments: f32 MFMA: lane (channel col, k = 16 ks + 4 quad + e);  SPLIT: lane (channel col, k = 32 s + 8 quad + e)
  const float* wl = wc + ((size_t)cblk * cpb + col) * AK + (SPLIT ? 8 : 4) * quad;
  const float* al = sf + col * FLD + 4 * quad;
  auto load_w = [&](int g, f32x4* dst) {
#pragma unroll
    for (int ks = 0; ks < AK / 16; ++ks)
      dst[ks] = *reinterpret_cast<const f32x4*>(wl + (size_t)g * 16 * AK + (SPLIT ? (ks >> 1) * 32 + (ks & 1) * 4 : ks * 16));
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
    if constexpr (SPLIT) {
      const float WS = wscale;            // a power of two from the host: max |w| WS in [512, 1024) (sd_asp_attend_pool_dt alone: 2^8, clamped)
      h8 whi[4], wlo[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float w = sd_split16_clamp(WS * wf[2 * s4 + (e >> 2)][e & 3]);
          whi[s4][e] = (_Float16)w;
          wlo[s4][e] = (_Float16)(w - (float)whi[s4][e]);
        }
      const _Float16* ahp = s_hi + col * SLD + 8 * quad;
      const _Float16* alp = s_lo + col * SLD + 8 * quad;
      // (frame tile j, k slice s4): two 16-byte LDS reads, three MFMAs; the reads of pair u + 1 are issued in front of the MFMAs of
      // pair u (two fragment sets)
      auto rd = [&](int u, h8& ah, h8& alo_) {
        const int s4 = u / NT, j = u % NT;
        ah = *reinterpret_cast<const h8*>(ahp + j * 16 * SLD + s4 * 32);
        alo_ = *reinterpret_cast<const h8*>(alp + j * 16 * SLD + s4 * 32);
      };
      h8 ah0, al0;
      rd(0, ah0, al0);
#pragma unroll
      for (int u = 0; u < 4 * NT; ++u) {
        const int s4 = u / NT, j = u % NT;
        h8 ah1 = ah0, al1 = al0;
        if (u + 1 < 4 * NT) rd(u + 1, ah1, al1);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[s4], ah0, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[s4], al0, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[s4], ah0, acc[j], 0, 0, 0);
        ah0 = ah1; al0 = al1;
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // the two LDS reads of the next pair first ...
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);   // ... then this pair's MFMAs
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] *= (1.0f / WS);
    } else {
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
    }
    // the next group's weight fragments load under the softmax phase (wf is dead from here on)
    if (g + 8 < ngroups) load_w(g + 8, wf);
    // softmax over the frames + weighted mean / variance (two passes), four channels per lane: packed f32 pairs
    // (channels r, r + 1), one FMA + v_exp_f32 per weight, single-instruction DPP reductions.  Frames past T are
    // masked only in the tiles that can hold them (the dispatch guarantees T > 16 JMIN).
    constexpr int JMIN = NT == 4 ? 0 : NT == 8 ? 4 : NT == 13 ? 8 : 13;
    constexpr float LOG2E = 1.4426950408889634f;
    float m[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (j >= JMIN) acc[j][r] = j * 16 + col < Tn ? acc[j][r] : -INFINITY;
        mx = fmaxf(mx, acc[j][r]);
      }
      m[r] = mx;
    }
    SD_DPP_RED4("v_max_f32_dpp", m[0], m[1], m[2], m[3]);      // (finite: frame 0 of every segment is live)
    float d[4], n[4], v[4];
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
      const f32x2 nb = {-m[2 * rp] * LOG2E, -m[2 * rp + 1] * LOG2E};
      f32x2 dd = {0.f, 0.f}, nn = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const f32x2 e = f32x2{acc[j][2 * rp], acc[j][2 * rp + 1]} * f32x2{LOG2E, LOG2E} + nb;
        const f32x2 w = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
        acc[j][2 * rp] = w[0];
        acc[j][2 * rp + 1] = w[1];
        dd += w;
        nn += w * f32x2{hv[j][2 * rp], hv[j][2 * rp + 1]};
      }
      d[2 * rp] = dd[0]; d[2 * rp + 1] = dd[1];
      n[2 * rp] = nn[0]; n[2 * rp + 1] = nn[1];
    }
    SD_DPP_RED4("v_add_f32_dpp", d[0], d[1], d[2], d[3]);
    SD_DPP_RED4("v_add_f32_dpp", n[0], n[1], n[2], n[3]);
    f32x4 mu4, sd4;
#pragma unroll
    for (int r = 0; r < 4; ++r) mu4[r] = n[r] / d[r];
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
      const f32x2 mu2 = {mu4[2 * rp], mu4[2 * rp + 1]};
      f32x2 vv = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const f32x2 dl = f32x2{hv[j][2 * rp], hv[j][2 * rp + 1]} - mu2;
        vv += f32x2{acc[j][2 * rp], acc[j][2 * rp + 1]} * (dl * dl);
      }
      v[2 * rp] = vv[0]; v[2 * rp + 1] = vv[1];
    }
    SD_DPP_RED4("v_add_f32_dpp", v[0], v[1], v[2], v[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) sd4[r] = sqrtf(fmaxf(v[r] / d[r], eps));
    if (col == 0) {
      float* o = out + (size_t)b * 2 * C + (size_t)cblk * cpb + g * 16 + 4 * quad;
      *reinterpret_cast<f32x4*>(o) = mu4;
      *reinterpret_cast<f32x4*>(o + C) = sd4;
    }
  }
}

template <int NT, bool SPLIT = false>
int launch_f32(const void* a1, const void* wc, const void* h, int ldh, int B, int T, int C, float eps, float* out, hipStream_t s, float wscale = 256.f) {
  auto kern = asp_attend_pool_f32_kernel<NT, SPLIT>;
  const size_t lds = SPLIT ? (size_t)2 * NT * 16 * SLD * sizeof(_Float16) : (size_t)NT * 16 * FLD * sizeof(float);
  // channels per workgroup (one workgroup per CU: the a1 tile fills most of the LDS; a wave takes ~13 us per 16-channel group):
  // 512 (four groups per wave) for launches that fill the chip anyway, fewer while the launch still fits one round of the 256 CUs
  // (16 segments x 3072 channels: 96 workgroups of 512 -> 192 of 256: 64 -> 33 us); every channel is computed alike whatever the split
  int cpb = C % 512 == 0 ? 512 : 256;
  while (cpb > 128 && C % (cpb / 2) == 0 && (long)B * (C / (cpb / 2)) <= 256) cpb /= 2;
  SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(kern), (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)((long)B * (C / cpb))), dim3(512), lds, s, static_cast<const float*>(a1),
                     static_cast<const float*>(wc), static_cast<const float*>(h), ldh, T, C, cpb, eps, out, wscale);
  SD_CHECK_LAUNCH("asp_attend_pool_f32_kernel");
  return SD_OK;
}

}  // namespace

extern "C" int sd_asp_attend_pool_supported(int dtype, int T, int C, int att) {
  return (dtype == SD_DT_F16 || dtype == SD_DT_F32 || dtype == SD_DT_SPLIT16) && att == AK && C > 0 && C % CPB == 0 && T > 0 && T <= 256;
}

extern "C" int sd_asp_attend_pool_dt(const void* a1, const void* wc, const void* h, int dtype, int ldh, int B, int T, int C,
                                     int att, float eps, float* out, sd_stream_t stream) {
  return sd_asp_attend_pool_scaled(a1, wc, h, dtype, ldh, B, T, C, att, eps, 256.f, out, stream);
}

// w_scale (SD_DT_SPLIT16 only): the power of two the f32 attention-conv weights are multiplied with before they are split into f16
// halves; the forward passes the data-dependent 2^s of the layer (max |w| 2^s in [512, 1024), as every other split weight); the
// public entry above passes 2^8, the right magnitude for trained ECAPA weights (|w| ~ 0.1), and values beyond the f16 range clamp.
int sd_asp_attend_pool_scaled(const void* a1, const void* wc, const void* h, int dtype, int ldh, int B, int T, int C,
                              int att, float eps, float w_scale, float* out, sd_stream_t stream) {
  SD_CHECK_ARG(a1 && wc && h && out, "sd_asp_attend_pool_dt: null pointer");
  SD_CHECK_ARG(w_scale > 0.f, "sd_asp_attend_pool_dt: w_scale=%g", (double)w_scale);
  SD_CHECK_ARG(B >= 0 && (long)B * (C > 0 ? C : 1) < (1L << 31), "sd_asp_attend_pool_dt: B=%d", B);
  if (!sd_asp_attend_pool_supported(dtype, T, C, att))
    return sd_set_error(SD_ERR_UNSUPPORTED, "sd_asp_attend_pool_dt: dtype=%d T=%d C=%d att=%d not covered (att=128, C%%256==0, T<=256)",
                        dtype, T, C, att);
  SD_CHECK_ARG(ldh >= C && ldh % 4 == 0, "sd_asp_attend_pool_dt: ldh=%d (need >= C and a multiple of 4)", ldh);
  SD_CHECK_ARG(sd_aligned16(a1) && sd_aligned16(wc) && sd_aligned16(h) && sd_aligned16(out), "sd_asp_attend_pool_dt: pointers must be 16-byte aligned");
  if (B == 0) return SD_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == SD_DT_SPLIT16) {       // f32 tensors, the logits product as three f16 MFMA products per value pair (f32-split16x3 mode)
    if (T <= 64) return launch_f32<4, true>(a1, wc, h, ldh, B, T, C, eps, out, s, w_scale);
    if (T <= 128) return launch_f32<8, true>(a1, wc, h, ldh, B, T, C, eps, out, s, w_scale);
    if (T <= 208) return launch_f32<13, true>(a1, wc, h, ldh, B, T, C, eps, out, s, w_scale);
    return launch_f32<16, true>(a1, wc, h, ldh, B, T, C, eps, out, s, w_scale);
  }
  if (dtype == SD_DT_F32) {
    if (T <= 64) return launch_f32<4>(a1, wc, h, ldh, B, T, C, eps, out, s);
    if (T <= 128) return launch_f32<8>(a1, wc, h, ldh, B, T, C, eps, out, s);
    if (T <= 208) return launch_f32<13>(a1, wc, h, ldh, B, T, C, eps, out, s);
    return launch_f32<16>(a1, wc, h, ldh, B, T, C, eps, out, s);
  }
  if (T <= 64) return launch_f16<1>(a1, wc, h, ldh, B, T, C, eps, out, s);
  if (T <= 128) return launch_f16<2>(a1, wc, h, ldh, B, T, C, eps, out, s);
  if (T <= 192) return launch_f16<3>(a1, wc, h, ldh, B, T, C, eps, out, s);     // (the kernel relies on T > 64 (TPW - 1))
  return launch_f16<4>(a1, wc, h, ldh, B, T, C, eps, out, s);
}
