// Shared epilogue of the implicit-GEMM conv kernels.
//
// Phase 1 (in the kernel): raw f32 accumulators -> LDS tile Cs[rows][ldc].
// Phase 2 (here): every thread owns 8 consecutive output channels for all of its rows, so the
// per-channel parameters (bias, BatchNorm scale/shift) are loaded once; per row it reads two
// float4 from LDS, applies  bias -> act -> affine -> act2  in f32, converts and issues 16-byte
// row-contiguous stores (plus the optional Res2Net tee  y + next chunk).
//
// The activation selector is resolved ONCE per thread, not per element: ReLU / identity (every
// large layer of ECAPA-TDNN) take a branch-free path (max with 0 or -inf); tanh / sigmoid (the
// attention TDNN and the SE gate) take the generic path.  An earlier version evaluated a
// per-element `switch` inside the fully unrolled 64-element accumulator loop, which inlined
// tanhf/expf 128 times per thread (18k instructions per kernel) and cost ~12 us per tile.
#pragma once
#include "sd_common.h"

__device__ __forceinline__ float sd_apply_act(float v, int act) {
  switch (act) {
    case SD_ACT_RELU: return fmaxf(v, 0.0f);
    case SD_ACT_TANH: return tanhf(v);
    case SD_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

template <typename TO> struct SdOut;
template <> struct SdOut<float> {
  static __device__ __forceinline__ void store8(float* p, const float* v) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
  static __device__ __forceinline__ void load8(const float* p, float* v) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
  }
};
template <> struct SdOut<_Float16> {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ void store8(_Float16* p, const float* v) {
    h8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (_Float16)v[e];
    *reinterpret_cast<h8*>(p) = r;
  }
  static __device__ __forceinline__ void load8(const _Float16* p, float* v) {
    const h8 r = *reinterpret_cast<const h8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)r[e];
  }
};

// ROWS x COLS tile at (m0, n0); NT threads; vec != 0 when every touched row slice is 16-byte aligned
// and cout / column offsets are multiples of 8 (decided on the host).
template <typename TO, int ROWS, int COLS, int NT>
__device__ __forceinline__ void sd_store_tile(const sd_conv_args& p, const float* Cs, int ldc, int m0, int n0, int tid, int vec) {
  constexpr int TPR = COLS / 8;     // threads per tile row
  constexpr int RPP = NT / TPR;     // rows per pass
  TO* const Y = static_cast<TO*>(p.y);
  TO* const TEE = static_cast<TO*>(p.tee);
  const TO* const TADD = static_cast<const TO*>(p.tee_add);
  const int cq = (tid % TPR) * 8;
  const int n8 = n0 + cq;
  if (n8 >= p.cout) return;
  const int nvalid = p.cout - n8 < 8 ? p.cout - n8 : 8;

  float b8[8], s8[8], h8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool ok = e < nvalid;
    b8[e] = (ok && p.bias && !p.bias_per_seg) ? p.bias[n8 + e] : 0.f;
    s8[e] = (ok && p.scale) ? p.scale[n8 + e] : 1.f;
    h8[e] = (ok && p.shift) ? p.shift[n8 + e] : 0.f;
  }
  const bool simple = (p.act == SD_ACT_RELU || p.act == SD_ACT_NONE) && p.act2 == SD_ACT_NONE;
  const float lo = p.act == SD_ACT_RELU ? 0.f : -INFINITY;
  const bool tee_q = TEE && n8 >= p.tee_lo && n8 < p.tee_hi;   // exact for the vec path (ranges are multiples of 8)

#pragma unroll 2
  for (int rr = tid / TPR; rr < ROWS; rr += RPP) {
    const int m = m0 + rr;
    if (m >= p.M) break;
    float v[8];
    {
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(Cs + rr * ldc + cq);
      const f32x4 c1 = *reinterpret_cast<const f32x4*>(Cs + rr * ldc + cq + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = c0[e]; v[4 + e] = c1[e]; }
    }
    if (p.bias_per_seg) {
      const float* sb = p.bias + (size_t)(m / p.T) * p.cout + n8;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += e < nvalid ? sb[e] : 0.f;
    }
    if (simple) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] + b8[e], lo) * s8[e] + h8[e];
    } else {
#pragma unroll   // static register indices (a runtime-indexed array would live in scratch)
      for (int e = 0; e < 8; ++e) v[e] = sd_apply_act(sd_apply_act(v[e] + b8[e], p.act) * s8[e] + h8[e], p.act2);
    }
    if (vec) {
      SdOut<TO>::store8(Y + (size_t)m * p.ldo + p.o_col0 + n8, v);
      if (tee_q) {
        if (TADD) {
          float t[8];
          SdOut<TO>::load8(TADD + (size_t)m * p.ld_ta + p.ta_col0 + (n8 - p.tee_lo), t);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += t[e];
        }
        SdOut<TO>::store8(TEE + (size_t)m * p.ldt + (n8 - p.tee_lo), v);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (e >= nvalid) break;
        const int n = n8 + e;
        Y[(size_t)m * p.ldo + p.o_col0 + n] = (TO)v[e];
        if (TEE && n >= p.tee_lo && n < p.tee_hi) {
          float tv = v[e];
          if (TADD) tv += (float)TADD[(size_t)m * p.ld_ta + p.ta_col0 + (n - p.tee_lo)];
          TEE[(size_t)m * p.ldt + (n - p.tee_lo)] = (TO)tv;
        }
      }
    }
  }
}
