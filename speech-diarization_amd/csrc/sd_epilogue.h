// Shared epilogue of the implicit-GEMM conv kernels.
//
// Phase 1 (in the kernel): raw f32 accumulators -> LDS tile Cs[rows][ldc].
// Phase 2 (here): every thread owns 8 consecutive output channels for all of its rows, so the
// per-channel parameters (bias, BatchNorm scale/shift) are loaded once; per row it reads two
// float4 from LDS, applies  bias -> act -> affine -> act2  in f32, converts and issues 16-byte
// row-contiguous stores (plus the optional Res2Net tee  y + next chunk).
//
// Two things shape the code (both measured with in-kernel cycle counters):
// * The activation selector is resolved ONCE per thread, not per element: ReLU / identity (every
//   large layer of ECAPA-TDNN) take a branch-free path (max with 0 or -inf).  tanh / sigmoid and
//   the per-segment bias (the attention TDNN) are applied by a small rolled pre-pass, in place in
//   the LDS tile.  An earlier version evaluated a per-element `switch` inside the fully unrolled
//   accumulator loop, which inlined tanhf/expf 128 times per thread (18k instructions per kernel).
// * The store phase contains NO loads.  On gfx950 loads and stores share `vmcnt` and may retire
//   out of order with respect to each other, so the compiler answers any load that follows a
//   store with `s_waitcnt vmcnt(0)`: a row loop that loaded (or might load) per row waited for
//   the previous row's HBM write to be acknowledged, ~3.2k cycles per row, 25.6k of a 28.8k-cycle
//   epilogue.  Everything a row needs (LDS values, the optional Res2Net `tee_add` rows) is
//   therefore fetched for all rows first, then all stores are issued back to back.
#pragma once
#include "sd_common.h"

__device__ __forceinline__ float sd_apply_act(float v, int act) {
  switch (act) {
    case SD_ACT_RELU: return sd_max_keep_nan(v, 0.0f);
    case SD_ACT_TANH: return tanhf(v);
    case SD_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

template <typename TO> struct SdOut;
template <> struct SdOut<float> {
  struct raw8 { f32x4 a, b; };
  static __device__ __forceinline__ raw8 load_raw(const float* p) {
    return raw8{*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4)};
  }
  static __device__ __forceinline__ void add_raw(float* v, const raw8& r) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] += r.a[e]; v[4 + e] += r.b[e]; }
  }
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
  typedef h8 raw8;
  static __device__ __forceinline__ raw8 load_raw(const _Float16* p) { return *reinterpret_cast<const h8*>(p); }
  static __device__ __forceinline__ void add_raw(float* v, const raw8& r) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
  }
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

// ---- fallback for outputs that cannot take 16-byte stores (cout % 8 != 0 or unaligned slices)
template <typename TO, int ROWS, int COLS, int NT>
__device__ __forceinline__ void sd_store_tile_scalar(const sd_conv_args& p, const float* Cs, int ldc, int m0, int n0, int tid) {
  constexpr int TPR = COLS / 8;
  constexpr int RPP = NT / TPR;
  TO* const Y = static_cast<TO*>(p.y);
  TO* const TEE = static_cast<TO*>(p.tee);
  const TO* const TADD = static_cast<const TO*>(p.tee_add);
  const int cq = (tid % TPR) * 8;
  const int n8 = n0 + cq;
  if (n8 >= p.cout) return;
  const int nvalid = p.cout - n8 < 8 ? p.cout - n8 : 8;
#pragma unroll 1
  for (int rr = tid / TPR; rr < ROWS; rr += RPP) {
    const int m = m0 + rr;
    if (m >= p.M) break;
    const float* sb = p.bias ? (p.bias_per_seg ? p.bias + (size_t)(m / p.T) * p.cout : p.bias) : nullptr;
#pragma unroll 1
    for (int e = 0; e < nvalid; ++e) {
      const int n = n8 + e;
      float v = Cs[rr * ldc + cq + e] + (sb ? sb[n] : 0.f);
      v = sd_apply_act(v, p.act) * (p.scale ? p.scale[n] : 1.f) + (p.shift ? p.shift[n] : 0.f);
      v = sd_apply_act(v, p.act2);
      Y[(size_t)m * p.ldo + p.o_col0 + n] = (TO)v;
      if (TEE && n >= p.tee_lo && n < p.tee_hi) {
        if (TADD) v += (float)TADD[(size_t)m * p.ld_ta + p.ta_col0 + (n - p.tee_lo)];
        TEE[(size_t)m * p.ldt + (n - p.tee_lo)] = (TO)v;
      }
    }
  }
}

// Store phase of the vector path for a FULL tile (no row checks: straight-line code).  TEE adds
// the Res2Net copy of a channel range; TADD also adds another tensor's rows to that copy, which
// are all fetched before the first store (so the rows cannot go out in small chunks).
// FULL: every row of the tile exists (straight-line code); otherwise the thread's first `np` passes
// do (the tile hangs over row M) and the rest are predicated off.
template <typename TO, int PASSES, int RPP, bool TEE, bool TADD, bool STAT, bool FULL, int PARTS = 3, bool YSPLIT = false>
__device__ __forceinline__ void sd_store_rows(const sd_conv_args& p, const float* c, int ldc, size_t row0, int n8,
                                              const float* b8, const float* s8, const float* h8, float lo,
                                              int rr0, int rb, float (*st)[8], int np) {
  // LDS rows are fetched and stored in chunks of 4 (1 with the 32 statistics accumulators live: the
  // 256x256 kernel has 128 VGPRs beside its accumulators).  The tee_add rows are GLOBAL loads and must
  // all be issued before the first store (header comment), so they are prefetched for the whole tile.
  constexpr int CH = STAT ? 1 : (PASSES % 4 == 0 ? 4 : (PASSES < 8 ? PASSES : 1));     // (5, 6, 7 passes: the 80 / 96 / 112-row tiles, in one chunk)
  static_assert(PASSES % CH == 0 && (TEE || !TADD), "");
  TO* const y = static_cast<TO*>(p.y) + row0 * p.ldo + p.o_col0 + n8;
  TO* const tee = TEE ? static_cast<TO*>(p.tee) + row0 * p.ldt + (n8 - p.tee_lo) : nullptr;
  typename SdOut<TO>::raw8 t[TADD ? PASSES : 1];
  if (TADD) {
    const TO* const ta = static_cast<const TO*>(p.tee_add) + row0 * p.ld_ta + p.ta_col0 + (n8 - p.tee_lo);
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int ps = FULL ? i : (i < np ? i : (np > 0 ? np - 1 : 0));   // stay inside the tensor
      t[i] = SdOut<TO>::load_raw(ta + (size_t)ps * RPP * p.ld_ta);
    }
  }
#pragma unroll
  for (int c0 = 0; c0 < PASSES; c0 += CH) {
    float v[CH][8];
#pragma unroll
    for (int i = 0; i < CH; ++i) SdOut<float>::load8(c + (c0 + i) * RPP * ldc, v[i]);
#pragma unroll
    for (int i = 0; i < CH; ++i) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = sd_max_keep_nan(v[i][e] + b8[e], lo) * s8[e] + h8[e];
      const bool live = FULL || c0 + i < np;
      if (YSPLIT && sizeof(TO) == 4 && p.y_dtype == SD_DT_SPLIT16) {      // (YSPLIT: only the kernels whose host entry accepts such a y)
        // y as SD_DT_SPLIT16 rows of p.ldo VALUE columns (what sd_split16_pack_f32 would make of the f32 result, bit for bit:
        // hi = f16(v), lo = f16(v - hi), [hi x 32 | lo x 32] per 32 columns); the host guarantees ldo % 32 == 0, (o_col0 + n8) % 8 == 0
        if (live) {
          typedef _Float16 h8s __attribute__((ext_vector_type(8)));
          const int col = p.o_col0 + n8;
          _Float16* ys = static_cast<_Float16*>(p.y) + (row0 + (size_t)(c0 + i) * RPP) * 2 * p.ldo + (col >> 5) * 64 + (col & 31);
          h8s hi, lo;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float w = sd_split16_clamp(v[i][e]);
            hi[e] = (_Float16)w;
            lo[e] = (_Float16)(w - (float)hi[e]);
          }
          *reinterpret_cast<h8s*>(ys) = hi;
          *reinterpret_cast<h8s*>(ys + 32) = lo;
        }
      } else if (live) {
        SdOut<TO>::store8(y + (size_t)(c0 + i) * RPP * p.ldo, v[i]);
      }
      if (STAT) {
        // column statistics of this thread's rows, split at the segment boundaries rb, rb + T (tile-relative);
        // taken about the pivot h8 (the BatchNorm shift) so that sum((x - pivot)^2) does not cancel
        const int row = rr0 + (c0 + i) * RPP;
        const int part = (row >= rb ? 1 : 0) + (row >= rb + p.T ? 1 : 0);      // segment of the tile this row is in
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = live ? v[i][e] - h8[e] : 0.f;
#pragma unroll
          for (int q = 0; q < PARTS; ++q) {
            st[q][e] += part == q ? x : 0.f;
            st[PARTS + q][e] += part == q ? x * x : 0.f;
          }
        }
      }
      if (TEE) {
        if (TADD) SdOut<TO>::add_raw(v[i], t[c0 + i]);
        if (live) SdOut<TO>::store8(tee + (size_t)(c0 + i) * RPP * p.ldt, v[i]);
      }
    }
    if (STAT || TADD) __builtin_amdgcn_sched_barrier(0);   // keep the next chunk's LDS reads from being hoisted (register budget)
  }
}

// ROWS x COLS tile at (m0, n0); NT threads; vec != 0 when every touched row slice and the
// per-channel parameter vectors are 16-byte aligned and cout / column offsets are multiples of 8
// (decided on the host).  A tile that hangs over the last row runs the same code with its missing rows
// predicated off (an element-wise path there made the one straggling workgroup the critical path of
// small launches: a 32-segment batch has 50.25 row tiles).
// TEE_MODE: 2 = full (tee with or without tee_add), 1 = tee without tee_add only (the host never
// selects such a kernel for a tee_add layer; saves the registers of the prefetched rows).
// STAT_PARTS: segments a tile may span for the column statistics: 3 (T >= ROWS / 2) or 2 (T >= ROWS: the
// 256x256 kernel, which has no registers for the third set of accumulators).
template <typename TO, int ROWS, int COLS, int NT, int TEE_MODE = 2, int STAT_PARTS = 3, bool YSPLIT = false>
__device__ __forceinline__ void sd_store_tile(const sd_conv_args& p, float* Cs, int ldc, int m0, int n0, int tid, int vec) {
  constexpr int TPR = COLS / 8;     // threads per tile row
  constexpr int RPP = NT / TPR;     // rows per pass
  constexpr int PASSES = ROWS / RPP;
  static_assert(ROWS % RPP == 0, "tile rows must be a multiple of the rows covered per pass");
  static_assert(2 * STAT_PARTS * RPP * COLS <= ROWS * COLS, "column statistics are combined inside the C tile");   // (STAT_PARTS = 0: a kernel without them)
  if (!vec) {
    sd_store_tile_scalar<TO, ROWS, COLS, NT>(p, Cs, ldc, m0, n0, tid);
    return;
  }
  const int nrows = p.M - m0 < ROWS ? p.M - m0 : ROWS;     // rows of this tile that exist
  if (nrows <= 0) return;            // second half of a 256-row tile that starts past the last row (uniform)
  const bool full = nrows == ROWS;
  const int cq = (tid % TPR) * 8;
  const int n8 = n0 + cq;
  const int rr0 = tid / TPR;
  if (n8 >= p.cout) return;         // cout % 8 == 0 here: a group is entirely inside or outside

  float b8[8], s8[8], h8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { b8[e] = 0.f; s8[e] = 1.f; h8[e] = 0.f; }
  if (p.bias && !p.bias_per_seg) SdOut<float>::load8(p.bias + n8, b8);
  if (p.scale) SdOut<float>::load8(p.scale + n8, s8);
  if (p.shift) SdOut<float>::load8(p.shift + n8, h8);
  const bool simple = (p.act == SD_ACT_RELU || p.act == SD_ACT_NONE) && p.act2 == SD_ACT_NONE;
  float lo = p.act == SD_ACT_RELU ? 0.f : -INFINITY;
  float* const c = Cs + rr0 * ldc + cq;

  if (!simple || p.bias_per_seg) {
    // rare layers: fold the per-segment bias and/or the transcendental activations into the LDS
    // tile (every thread rewrites exactly the slots it reads back below: no barrier needed)
#pragma unroll 1
    for (int ps = 0; ps < PASSES; ++ps) {
      if (rr0 + ps * RPP >= nrows) break;
      float* cr = c + ps * RPP * ldc;
      float v[8];
      SdOut<float>::load8(cr, v);
      if (p.bias_per_seg) {
        float sb[8];
        SdOut<float>::load8(p.bias + (size_t)((m0 + rr0 + ps * RPP) / p.T) * p.cout + n8, sb);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += sb[e];
      }
      if (!simple) {
#pragma unroll   // static register indices (a runtime-indexed array would live in scratch)
        for (int e = 0; e < 8; ++e) v[e] = sd_apply_act(sd_apply_act(v[e] + b8[e], p.act) * s8[e] + h8[e], p.act2);
      }
      SdOut<float>::store8(cr, v);
    }
    if (!simple) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { b8[e] = 0.f; s8[e] = 1.f; h8[e] = 0.f; }
      lo = -INFINITY;
    }
  }

  const bool tee_q = p.tee && n8 >= p.tee_lo && n8 < p.tee_hi;   // ranges are multiples of 8
  const int np = rr0 < nrows ? (nrows - rr0 + RPP - 1) / RPP : 0;             // this thread's existing rows
  const size_t row0 = (size_t)(m0 + (rr0 < nrows ? rr0 : 0));
#define SD_ROWS(TEE_, TADD_, STAT_, ST_, RB_)                                                                            \
  do {                                                                                                                   \
    if (full) sd_store_rows<TO, PASSES, RPP, TEE_, TADD_, STAT_, true, 3, YSPLIT>(p, c, ldc, row0, n8, b8, s8, h8, lo, rr0, RB_, ST_, PASSES); \
    else sd_store_rows<TO, PASSES, RPP, TEE_, TADD_, STAT_, false, 3, YSPLIT>(p, c, ldc, row0, n8, b8, s8, h8, lo, rr0, RB_, ST_, np);        \
  } while (0)
  if (STAT_PARTS > 0 && p.colstat) {
    constexpr int SP = STAT_PARTS > 0 ? STAT_PARTS : 1;
    // (host: only with relu / identity, a per-channel bias, cout % COLS == 0, no tee and T >= ROWS / 2, so
    // this branch is uniform over the workgroup, nobody returned above, and a tile spans <= 3 segments)
    float st[2 * SP][8];          // [sum | sum of squares] x [first, second(, third) segment of the tile]
#pragma unroll
    for (int k = 0; k < 2 * SP; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) st[k][e] = 0.f;
    const int rb = p.T - m0 % p.T;          // first tile row of the next segment (>= ROWS: none)
    if (full) {
      sd_store_rows<TO, PASSES, RPP, false, false, true, true, SP>(p, c, ldc, row0, n8, b8, s8, h8, lo, rr0, rb, st, PASSES);
    } else {
      // the one tile that hangs over row M: predicated stores, then a rolled pass over this thread's
      // existing rows for the statistics (keeps the unrolled variant's registers out of the common path)
      sd_store_rows<TO, PASSES, RPP, false, false, false, false>(p, c, ldc, row0, n8, b8, s8, h8, lo, rr0, 0, nullptr, np);
#pragma unroll 1
      for (int ps = 0; ps < np; ++ps) {
        float v[8];
        SdOut<float>::load8(c + ps * RPP * ldc, v);
        const int row = rr0 + ps * RPP;
        const int part = (row >= rb ? 1 : 0) + (row >= rb + p.T ? 1 : 0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = sd_max_keep_nan(v[e] + b8[e], lo) * s8[e];     // = y - shift
#pragma unroll
          for (int q = 0; q < SP; ++q) {
            st[q][e] += part == q ? x : 0.f;
            st[SP + q][e] += part == q ? x * x : 0.f;
          }
        }
      }
    }
    // combine the RPP row groups through LDS in a fixed order, then one writer per (quantity, column)
    __syncthreads();                        // every thread has consumed its part of the C tile
    float* red = Cs;                        // [2 * SP][RPP][COLS]
#pragma unroll
    for (int k = 0; k < 2 * SP; ++k) SdOut<float>::store8(red + ((size_t)k * RPP + rr0) * COLS + cq, st[k]);
    __syncthreads();
    // colstat unit: [sum part 0..2 | sum of squares part 0..2][cout]; a 2-part kernel leaves slots 2 and 5 alone
    float* dst = p.colstat + (size_t)(m0 / ROWS) * 6 * p.cout + n0;
    for (int idx = tid; idx < 2 * SP * COLS; idx += NT) {
      const int k = idx / COLS, col = idx - k * COLS;
      float a = 0.f;
#pragma unroll
      for (int g = 0; g < RPP; ++g) a += red[((size_t)k * RPP + g) * COLS + col];
      const int slot = k < SP ? k : 3 + (k - SP);
      dst[(size_t)slot * p.cout + col] = a;
    }
    __syncthreads();                        // the caller may refill the tile
    return;
  }
  float (*nost)[8] = nullptr;
  if (tee_q) {
    if (TEE_MODE == 2 && p.tee_add) SD_ROWS(true, (TEE_MODE == 2), false, nost, 0);
    else SD_ROWS(true, false, false, nost, 0);
  } else {
    SD_ROWS(false, false, false, nost, 0);
  }
#undef SD_ROWS
}
