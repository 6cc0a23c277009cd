// f16-operand / f32-accumulate variant of the channel-last implicit-GEMM conv
// (BASELINE.json configs[4]: "fp16 ECAPA with CDNA4 fp16 MFMA on 1x1 convs").
//
// Same operator contract and schedule as sd_conv_gemm.hip, on v_mfma_f32_32x32x16_f16
// (16x the f32 MFMA rate).  Weights are packed f16 [cout][taps][cin_pad] with cin_pad a
// multiple of 64; activations are f16 in HBM (or f32, converted while staging — the stem
// reads the f32 fbank); accumulation, bias, activation and the BatchNorm affine are f32;
// the result is rounded once to f16 (or kept f32, e.g. the attention logits).
// The reference enables TF32 matmuls/convs on its CUDA path [REF diarization_baseline.py:20-21];
// f16 operands carry the same 11-bit significand, gfx950 has no xf32 MFMA.
//
// Tile 128x128, K step 64 halfs (a 128-byte row segment per operand row), LDS rows padded to
// 144 bytes (conflict-free ds_read_b128 fragments: lane (r, h) reads 8 consecutive k at
// 16*kk + 8h, exactly the A/B fragment of the 32x32x16 MFMA).  Workgroup ids are remapped so
// the n-tiles that share an A row panel run on one XCD and find it in that XCD's L2.
#include "sd_common.h"
#include "sd_epilogue.h"
#include <cstdlib>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 64;        // halfs
constexpr int LDP = BK + 8;   // padded LDS row, halfs (144 bytes)
constexpr int LDC = BN + 4;   // epilogue C tile row, floats
constexpr int STAGE_BYTES = 2 * (BM + BN) * LDP * 2;
static_assert(BM * LDC * 4 <= STAGE_BYTES, "C tile must fit in the operand stage");

template <typename TA>
__device__ __forceinline__ h8 load_a8(const TA* p);
template <>
__device__ __forceinline__ h8 load_a8<_Float16>(const _Float16* p) {
  return *reinterpret_cast<const h8*>(p);
}
template <>
__device__ __forceinline__ h8 load_a8<float>(const float* p) {
  const f32x4 lo = *reinterpret_cast<const f32x4*>(p);
  const f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
  h8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = (_Float16)lo[e]; r[4 + e] = (_Float16)hi[e]; }
  return r;
}

template <typename TA, typename TO>
__global__ __launch_bounds__(256, 2) void conv_gemm_f16_kernel(const sd_conv_args p, const int vec) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  _Float16* As = reinterpret_cast<_Float16*>(smem_raw);  // [2][BM][LDP]
  _Float16* Bs = As + 2 * BM * LDP;                       // [2][BN][LDP]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  // XCD-aware, bijective remap: consecutive tiles (n fastest: they share the A row panel) go
  // to one XCD instead of being dealt round-robin over the eight L2s
  const int n_tiles = (p.cout + BN - 1) / BN;
  int wg;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int tile_n = wg % n_tiles;
  const int tile_m = wg / n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // staging role: 8 threads per 64-half row, 4 rows per thread; all loads unconditional
  // (rows/channels past the end clamp to the last valid one and are never stored, columns past
  // cin re-read column 0 against the zero-filled weight padding)
  const int c8 = tid & 7;
  const int r0 = tid >> 3;
  int a_seg[4], a_t[4];
  const _Float16* wptr[4];
  const TA* aptr[4];
  const int ktot = p.taps * p.cin_pad;
  const _Float16* W = static_cast<const _Float16*>(p.w);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + r0 + 32 * i;
    m = m < p.M ? m : p.M - 1;
    const int seg = (m / p.T) * p.T;
    a_seg[i] = seg;
    a_t[i] = m - seg;
    int n = n0 + r0 + 32 * i;
    n = n < p.cout ? n : p.cout - 1;
    wptr[i] = W + (size_t)n * ktot + c8 * 8;
  }
  const int nk = p.taps * (p.cin_pad / BK);
  const int half = p.taps / 2;
  const TA* X = static_cast<const TA*>(p.x) + p.a_col0;

  auto set_tap = [&](int tap) {
    const int delta = (tap - half) * p.dil;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int tt = a_t[i] + delta;
      tt = tt < 0 ? -tt : tt;
      tt = tt >= p.T ? 2 * (p.T - 1) - tt : tt;
      aptr[i] = X + (size_t)(a_seg[i] + tt) * p.lda;
    }
  };

  // Two register stages: the fetch runs TWO K steps ahead of the MFMAs (one step is only ~0.2 us of
  // matrix work, far less than an HBM/L2 round trip), the LDS stage one step ahead.
  struct Stage { h8 a[4], b[4]; };
  Stage s0, s1;
  int ld_tap = 0, ld_c0 = 0;
  set_tap(0);
  auto gload = [&](Stage& st) {
    const int col = ld_c0 + c8 * 8;
    const int acol = col < p.cin ? col : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      st.a[i] = load_a8<TA>(aptr[i] + acol);
      st.b[i] = *reinterpret_cast<const h8*>(wptr[i]);
      wptr[i] += BK;
    }
    ld_c0 += BK;
    if (ld_c0 >= p.cin_pad) {
      ld_c0 = 0;
      ++ld_tap;
      if (ld_tap < p.taps) set_tap(ld_tap);
    }
  };
  auto lstore = [&](const Stage& st, int buf) {
    _Float16* a = As + buf * BM * LDP;
    _Float16* b = Bs + buf * BN * LDP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<h8*>(a + (r0 + 32 * i) * LDP + c8 * 8) = st.a[i];
      *reinterpret_cast<h8*>(b + (r0 + 32 * i) * LDP + c8 * 8) = st.b[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_row = lane & 31;
  const int frag_k = (lane >> 5) * 8;

  struct Frag { h8 a0, a1, b0, b1; };
  auto fread = [&](const _Float16* a, const _Float16* b, int kk) {
    Frag f;
    f.a0 = *reinterpret_cast<const h8*>(a + kk * 16);
    f.a1 = *reinterpret_cast<const h8*>(a + 32 * LDP + kk * 16);
    f.b0 = *reinterpret_cast<const h8*>(b + kk * 16);
    f.b1 = *reinterpret_cast<const h8*>(b + 32 * LDP + kk * 16);
    return f;
  };
  auto mma = [&](const Frag& f) {
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a0, f.b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a0, f.b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a1, f.b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a1, f.b1, acc[1][1], 0, 0, 0);
  };

  gload(s0);                 // K step 0
  if (nk > 1) gload(s1);     // K step 1
  lstore(s0, 0);
  if (nk > 2) gload(s0);     // K step 2
  __syncthreads();

  int cur = 0;
  // one K step: MFMAs on LDS stage `cur`; `st` holds step kt+1 -> written to the other LDS stage,
  // then refilled with step kt+3
  auto kstep = [&](int kt, Stage& st) {
    const _Float16* a = As + cur * BM * LDP + (wm * 64 + frag_row) * LDP + frag_k;
    const _Float16* b = Bs + cur * BN * LDP + (wn * 64 + frag_row) * LDP + frag_k;
    Frag f0 = fread(a, b, 0);
    Frag f1 = fread(a, b, 1);
    mma(f0);
    f0 = fread(a, b, 2);
    mma(f1);
    f1 = fread(a, b, 3);
    if (kt + 1 < nk) lstore(st, cur ^ 1);
    if (kt + 3 < nk) gload(st);
    mma(f0);
    mma(f1);
    __syncthreads();
    cur ^= 1;
  };
  for (int kt = 0; kt < nk; kt += 2) {
    kstep(kt, s1);
    if (kt + 1 < nk) kstep(kt + 1, s0);
  }

  // ---- epilogue: raw accumulators -> LDS C tile -> sd_store_tile (sd_epilogue.h)
  float* Cs = reinterpret_cast<float*>(smem_raw);
  const int hrow = (lane >> 5) * 4;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int cl = wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + hrow;
        Cs[rl * LDC + cl] = acc[mi][ni][r];
      }
    }
  }
  __syncthreads();
  sd_store_tile<TO, BM, BN, 256>(p, Cs, LDC, m0, n0, tid, vec);
}

// ------------------------------------------------------------------------------------------
// f32-split16x3 for the NARROW outputs (Res2Net 128 -> 128 k = 3, attention TDNN 3C -> 128): the 128x128 register-staged kernel
// above with the split done WHILE STAGING.  Activations arrive as plain f32 (no pack pass, any lda / a_col0 slice, the tee /
// tee_add epilogue of the Res2Net chain); a thread loads 4 values (16 bytes), splits them hi = f16(v), lo = f16(v - hi) and
// stores both halves into the LDS row of its K step: [hi x 32 | lo x 32], the layout of the SD_DT_SPLIT16 weights, which are
// fetched as they lie.  A K step of 32 values = the four 16-half fragment slices hi0, hi1, lo0, lo1 and six MFMA groups,
// hi.hi + hi.lo + lo.hi per value slice.  The weights carry a power-of-two scale 2^s (low halves out of the f16 subnormals);
// the accumulators are multiplied by p.w_scale_inv = 2^-s on their way into the epilogue tile (exact), so bias, per-segment
// bias and the BatchNorm affine are the layer's own.
template <typename TO>
__global__ __launch_bounds__(256, 2) void conv_gemm_split16_n128_kernel(const sd_conv_args p, const int vec) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  _Float16* As = reinterpret_cast<_Float16*>(smem_raw);  // [2][BM][LDP]
  _Float16* Bs = As + 2 * BM * LDP;                       // [2][BN][LDP]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int n_tiles = (p.cout + BN - 1) / BN;
  int wg;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int tile_n = wg % n_tiles;
  const int tile_m = wg / n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // staging role: 8 threads per row, 4 rows per thread.  A: thread c8 owns VALUES 4 c8 .. 4 c8 + 3 of the step's 32 (one 16-byte
  // f32 load); B: halfs 8 c8 .. 8 c8 + 7 of the 64 (one 16-byte load of the packed weights)
  const int c8 = tid & 7;
  const int r0 = tid >> 3;
  int a_seg[4], a_t[4];
  const _Float16* wptr[4];
  const float* aptr[4];
  const int kvals = p.taps * p.cin_pad;                   // values per output channel; 2 halfs each
  const _Float16* W = static_cast<const _Float16*>(p.w);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + r0 + 32 * i;
    m = m < p.M ? m : p.M - 1;
    const int seg = (m / p.T) * p.T;
    a_seg[i] = seg;
    a_t[i] = m - seg;
    int n = n0 + r0 + 32 * i;
    n = n < p.cout ? n : p.cout - 1;
    wptr[i] = W + (size_t)n * 2 * kvals + c8 * 8;
  }
  const int nk = p.taps * (p.cin_pad / 32);
  const int half = p.taps / 2;
  const float* X = static_cast<const float*>(p.x) + p.a_col0;
  auto set_tap = [&](int tap) {
    const int delta = (tap - half) * p.dil;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int tt = a_t[i] + delta;
      tt = tt < 0 ? -tt : tt;
      tt = tt >= p.T ? 2 * (p.T - 1) - tt : tt;
      aptr[i] = X + (size_t)(a_seg[i] + tt) * p.lda;
    }
  };
  struct Stage { f32x4 a[4]; h8 b[4]; };
  Stage s0, s1;
  int ld_tap = 0, ld_c0 = 0;
  set_tap(0);
  auto gload = [&](Stage& st) {
    const int col = ld_c0 + c8 * 4;
    const int acol = col < p.cin ? col : 0;               // columns past cin meet the zero-filled weight padding
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      st.a[i] = *reinterpret_cast<const f32x4*>(aptr[i] + acol);
      st.b[i] = *reinterpret_cast<const h8*>(wptr[i]);
      wptr[i] += BK;
    }
    ld_c0 += 32;
    if (ld_c0 >= p.cin_pad) {
      ld_c0 = 0;
      ++ld_tap;
      if (ld_tap < p.taps) set_tap(ld_tap);
    }
  };
  auto lstore = [&](const Stage& st, int buf) {
    _Float16* a = As + buf * BM * LDP;
    _Float16* b = Bs + buf * BN * LDP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      h4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float w = sd_split16_clamp(st.a[i][e]);
        hi[e] = (_Float16)w;
        lo[e] = (_Float16)(w - (float)hi[e]);
      }
      *reinterpret_cast<h4*>(a + (r0 + 32 * i) * LDP + c8 * 4) = hi;
      *reinterpret_cast<h4*>(a + (r0 + 32 * i) * LDP + 32 + c8 * 4) = lo;
      *reinterpret_cast<h8*>(b + (r0 + 32 * i) * LDP + c8 * 8) = st.b[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frag_row = lane & 31;
  const int frag_k = (lane >> 5) * 8;
  struct Frag { h8 a0, a1, b0, b1; };
  auto fread = [&](const _Float16* a, const _Float16* b, int kk) {
    Frag f;
    f.a0 = *reinterpret_cast<const h8*>(a + kk * 16);
    f.a1 = *reinterpret_cast<const h8*>(a + 32 * LDP + kk * 16);
    f.b0 = *reinterpret_cast<const h8*>(b + kk * 16);
    f.b1 = *reinterpret_cast<const h8*>(b + 32 * LDP + kk * 16);
    return f;
  };
  auto mma = [&](const Frag& fa, const Frag& fb) {        // A fragments of fa against B fragments of fb
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.a0, fb.b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.a0, fb.b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.a1, fb.b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.a1, fb.b1, acc[1][1], 0, 0, 0);
  };

  gload(s0);                 // K step 0
  if (nk > 1) gload(s1);     // K step 1
  lstore(s0, 0);
  if (nk > 2) gload(s0);     // K step 2
  __syncthreads();
  int cur = 0;
  auto kstep = [&](int kt, Stage& st) {
    const _Float16* a = As + cur * BM * LDP + (wm * 64 + frag_row) * LDP + frag_k;
    const _Float16* b = Bs + cur * BN * LDP + (wn * 64 + frag_row) * LDP + frag_k;
    const Frag h0 = fread(a, b, 0);           // hi, values 0-15
    const Frag l0 = fread(a, b, 2);           // lo, values 0-15
    mma(h0, h0);
    const Frag h1 = fread(a, b, 1);           // hi, values 16-31
    mma(h0, l0);
    mma(l0, h0);
    const Frag l1 = fread(a, b, 3);           // lo, values 16-31
    if (kt + 1 < nk) lstore(st, cur ^ 1);
    if (kt + 3 < nk) gload(st);
    mma(h1, h1);
    mma(h1, l1);
    mma(l1, h1);
    __syncthreads();
    cur ^= 1;
  };
  for (int kt = 0; kt < nk; kt += 2) {
    kstep(kt, s1);
    if (kt + 1 < nk) kstep(kt + 1, s0);
  }

  // ---- epilogue: accumulators x 2^-s -> LDS C tile -> sd_store_tile (sd_epilogue.h)
  float* Cs = reinterpret_cast<float*>(smem_raw);
  const float alpha = p.w_scale_inv != 0.f ? p.w_scale_inv : 1.f;
  const int hrow = (lane >> 5) * 4;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int cl = wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + hrow;
        Cs[rl * LDC + cl] = acc[mi][ni][r] * alpha;
      }
    }
  }
  __syncthreads();
  sd_store_tile<TO, BM, BN, 256, 2, 3, true>(p, Cs, LDC, m0, n0, tid, vec);      // (the only kernel that may write y as SD_DT_SPLIT16)
}

#ifdef SD_STAMP
// diagnostic build only (build_native.py --stamp): per-workgroup cycle counters, read by tools/stamp_t256.py
__device__ unsigned long long sd_stamp_buf[8192 * 8];
#endif

#define SD_GLDS16(gptr, lptr)                                                              \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),  \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

constexpr int TBM = 256;
constexpr int TBN = 256;
constexpr int R3_A_STAGE = TBM * 128;                 // [256 rows][64 halfs]
constexpr int R3_B_STAGE = TBN * 128;
constexpr int R3_B_BASE = 3 * R3_A_STAGE;
constexpr int R3_LDS_BYTES = R3_B_BASE + 2 * R3_B_STAGE;   // 163 840 = all of the CU's LDS
static_assert(R3_LDS_BYTES == 160 * 1024, "ring fills the LDS exactly");
static_assert((TBM / 2) * TBN * 4 <= R3_LDS_BYTES, "half C tile must fit in the ring");

// ---- register epilogue of the 256x256 kernel (DIRECT): see the kernel's comment
__device__ __forceinline__ unsigned sd_pack_h2(float a, float b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  h2 v;
  v[0] = (_Float16)a;
  v[1] = (_Float16)b;
  return __builtin_bit_cast(unsigned, v);
}

// sum over the 16 lanes of a DPP row (lanes that share lane >> 4); every lane gets the total
__device__ __forceinline__ float sd_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

// acc[mi][nj][r] = y[row0 + 16 mi + fr][col0 + 16 nj + 4 fq + r] before bias; the wave owns rows [row0, row0 + 128) (one
// column-statistics unit) and channels [col0, col0 + 64)
template <typename TO, typename ACC>
__device__ __forceinline__ void sd_direct_epilogue(const sd_conv_args& p, ACC (&acc)[8][4], int row0, int col0, int fr, int fq) {
  if (row0 >= p.M) return;                         // the half of the last tile that starts past the last row: no rows, no statistics unit
  const float lo = p.act == SD_ACT_RELU ? 0.f : -INFINITY;
  TO* const Y = static_cast<TO*>(p.y) + p.o_col0;
  const int rb = p.T - row0 % p.T;                 // first unit-relative row of the next segment (>= 128: none)
  float* const cs = p.colstat ? p.colstat + (size_t)(row0 / 128) * 6 * p.cout : nullptr;
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) {                 // pairs of 16-channel tiles
    float b[2][4], s[2][4], h[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int c = col0 + 16 * (2 * jp + t) + 4 * fq;
      const int cl = c < p.cout ? c : 0;           // channels past cout (cout % 8 == 0): loads clamped, nothing stored
      const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + cl) : f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 sv = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + cl) : f32x4{1.f, 1.f, 1.f, 1.f};
      const f32x4 hv = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + cl) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) { b[t][r] = bv[r]; s[t][r] = sv[r]; h[t][r] = hv[r]; }
    }
    float st[2][2][2][4];                          // [tile of the pair][sum | sum of squares][segment part][channel]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) st[t][k][q][r] = 0.f;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const int m = row0 + 16 * mi + fr;
      const bool live = m < p.M;
      float v[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[t][r] = sd_max_keep_nan(acc[mi][2 * jp + t][r] + b[t][r], lo) * s[t][r] + h[t][r];
      if (cs) {
        const int part = 16 * mi + fr >= rb ? 1 : 0;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = live ? v[t][r] - h[t][r] : 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              st[t][0][q][r] += part == q ? x : 0.f;
              st[t][1][q][r] += part == q ? x * x : 0.f;
            }
          }
      }
      if constexpr (sizeof(TO) == 2) {
        // lane row q = fq holds channels 4 q .. 4 q + 3 of both tiles; after the swaps it holds 8 consecutive channels
        // of ONE tile: rows 0 / 2 of the even tile (channels 0-7 / 8-15), rows 1 / 3 of the odd tile
        const auto w0 = __builtin_amdgcn_permlane16_swap(sd_pack_h2(v[0][0], v[0][1]), sd_pack_h2(v[1][0], v[1][1]), false, false);
        const auto w1 = __builtin_amdgcn_permlane16_swap(sd_pack_h2(v[0][2], v[0][3]), sd_pack_h2(v[1][2], v[1][3]), false, false);
        const int c8 = col0 + 16 * (2 * jp + (fq & 1)) + 8 * (fq >> 1);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifdef SD_DIAG_NO_STORE      // timing-only diagnostic build: the stores predicated off by a value the compiler cannot see through
        if (live && c8 < p.cout && p.dil == 12345) *reinterpret_cast<u32x4*>(Y + (size_t)m * p.ldo + c8) = u32x4{w0[0], w1[0], w0[1], w1[1]};
#else
        if (live && c8 < p.cout) *reinterpret_cast<u32x4*>(Y + (size_t)m * p.ldo + c8) = u32x4{w0[0], w1[0], w0[1], w1[1]};
#endif
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int c = col0 + 16 * (2 * jp + t) + 4 * fq;
          if (live && c < p.cout) *reinterpret_cast<f32x4*>(Y + (size_t)m * p.ldo + c) = f32x4{v[t][0], v[t][1], v[t][2], v[t][3]};
        }
      }
    }
    if (cs) {
      // colstat unit: [sum part 0..2 | sum of squares part 0..2][cout]; this kernel's tiles span <= 2 segments
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            float tot[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) tot[r] = sd_row16_sum(st[t][k][q][r]);
            if (fr == 0)
              *reinterpret_cast<f32x4*>(cs + (size_t)(3 * k + q) * p.cout + col0 + 16 * (2 * jp + t) + 4 * fq) = f32x4{tot[0], tot[1], tot[2], tot[3]};
          }
    }
  }
}

// ------------------------------------------------------------------------------------------
// 256x256 tile for the wide outputs (cout >= 1024: C -> C, 3C -> 3C), fed by LDS-DMA (global_load_lds_dwordx4 writes the
// operand tiles straight into an LDS ring, no VGPR staging).  8 waves as 2 (M) x 4 (N), each 128 x 64 = 8 x 4 tiles of
// v_mfma_f32_16x16x32_f16 (128 accumulator registers); K step 64 halfs = 128-byte rows, so a DMA piece (one wave
// instruction, 1 KB) lands 8 whole rows and reads 8 FULL cache lines.  Swizzle: two rows share a 256-byte bank row;
// physical 16-byte slot = logical slot ^ ((row >> 1) & 7) makes the ds_read_b128 fragment reads conflict-free
// (applied on the DMA's per-lane SOURCE address, the LDS side of a DMA is lane-linear).
//
// What round 2 measured on this kernel and what the structure answers (tools/sweep_f16.py, tools/stamp_t256.py;
// 1 005 000 rows; 3072 x 3072 / 1024 x 1024, TFLOP/s on random data):
// * The chip is power-limited here: the same binary on zero-filled operands runs 1490 instead of 1140 (the in-kernel
//   clock, s_memtime / s_memrealtime, reads 1.75-1.9 GHz on random data).  Cycles saved return about 2/3 as time, energy
//   saved returns in full.  v_mfma_f32_16x16x32_f16 instead of 32x32x16: same flops per cycle, +6 % (1071 -> 1140): the
//   chip holds a higher clock on it (1.87 vs 1.75 GHz).
// * The DMA costs 28 % (1486 with the K loop's DMA removed vs 1073), and it is the CU-side issue / LDS-write path, not
//   memory: with every tile reading the same 256 rows of both operands (all L2 hits) the rate is unchanged (1082).
//   A wave issues no MFMA while it issues its 8 pieces (550-850 cycles with four waves issuing together).
// * Ring depth by operand: the 160 KB of LDS hold THREE stages of A (activations: streamed from HBM) and TWO of B
//   (weights: L2 / Infinity-Cache resident).  Waves 0-3 fetch the weights of step k + 1, waves 4-7 (their SIMD
//   partners) the activations of step k + 2 (vmcnt(8): one step stays in flight across the barrier), each at a point
//   of the step where the partner has MFMAs to issue.  (Two 64 KB stages, both operands by all waves: 1053 / 924.)
// * LDS fragment reads are software-pipelined under the MFMAs: a K step is four half-slices (k slice of 32 x upper /
//   lower 64 rows of the wave's 128): 4 A + 4 B fragments -> 16 MFMAs; two register sets per operand; the reads of
//   half-slice h + 1 are issued in front of the MFMAs of h, and the barrier that opens step k + 1 sits in front of
//   the LAST half-slice's MFMAs of step k, so the first reads of the next step are covered too: 1140 -> 1227 / 1027.
// * Measured and not kept: s_setprio around the MFMA groups (-2 %), static priority for waves 4-7 (-3 %), the roles
//   swapped between the wave halves (-3 %), issuing the activation pieces later in the step (each later slot -2..-4 %),
//   every wave's 8 pieces issued one at a time between groups of 8 MFMAs instead of as a block (-7 %: 1171 -> 1090).
// * Where this kernel stands (tools/micro/mfma_rate.hip: register operands, no memory operation, random data, all CUs):
//   v_mfma_f32_32x32x16_f16 issues every 32 cycles but the chip drops to 1.6-1.7 GHz: 1.61-1.72 PFLOP/s (2.47 on zeros) is
//   the power limit of the matrix pipe alone; v_mfma_f32_16x16x32_f16 issues at most every ~25 cycles per SIMD: 1.25-1.30 at
//   2.0-2.1 GHz (1.48 on zeros).  This kernel's 1227 at 1.87 GHz, with its LDS reads, DMA and address arithmetic beside the
//   MFMAs, is 25.6 cycles per MFMA and SIMD: the pipe is saturated at this instruction's rate and the clock is what the
//   power limit allows.  The same ring / DMA schedule / read-ahead with 32x32x16 MFMAs (4 x 2 tiles per wave, k-slices of
//   16, register epilogue joined with v_permlane32_swap) measured 1036-1079 against 1196-1212 in the same runs, as the
//   round-1 structure had (1071 vs 1140): its pipe is ~59 % busy — a K step's 64 KB through the 64 B/clk LDS-DMA path is
//   half of a saturated step, plus the fragment reads and the barrier: that instruction needs fewer operand bytes per flop
//   than this tile shape moves.
// The 256x256 f32 C tile does not fit LDS: the epilogue runs once per 128-row half.
// DIRECT: the MFMA operands are swapped (D = W . X^T: accumulator ROWS are output channels, its columns time rows),
// so a lane holds 4 CONSECUTIVE channels of one output row per 16x16 tile and the epilogue runs from registers:
// bias -> ReLU -> BatchNorm affine in f32, f16 pairs, one v_permlane16_swap per dword joins the two 8-byte halves a
// lane pair holds into 16 bytes, 16-byte row-contiguous stores (a wave writes whole 128-byte lines), and the column
// statistics are per-lane sums over the wave's 8 row tiles reduced across 16 lanes with DPP: no LDS round trip, no
// workgroup barrier.  For the plain epilogue only (ReLU / identity, per-channel bias, no tee): the host selects it.
// SPLIT (the "f32-split16x3" mode, sd_conv1d_cl_split16): both operands arrive as interleaved halves of f32 values,
// a 128-byte row piece = [hi(k0 .. k0+31) | lo(k0 .. k0+31)] with hi = f16(v), lo = f16(v - hi), and a K step of 32 values
// takes THREE products, hi.hi + hi.lo + lo.hi (the dropped lo.lo term and the representation error are 2^-22 relative:
// f32-level accuracy on the f16 matrix cores).  Ring, DMA schedule and fragment traffic are those of the f16 step; the
// MFMA count per step is 1.5x, so the feed path (the f16 kernel's limit) has 1.5x the time per byte.  Half-slice order
//   lo(g0).Bhi, hi(g0).Bhi, hi(g0).Blo, lo(g1).Bhi, hi(g1).Bhi, hi(g1).Blo
// reads every fragment set once, keeps two A sets and both B sets live, and ends on Blo, so that the deferred last
// half-slice does not collide with the next step's first B read (Bhi).
// (The N x N cosine affinity had a SYM form of this kernel in round 3: upper-triangle tiles + an LDS-transposed mirror; superseded by
// sd_affinity.hip, whose 128 x 128 tiles leave two workgroups on a CU: 3.2 -> 2.3 ms for 50 k x 50 k.)
// SUPER-TILE WALK (round 5; super_walk != 0, register epilogue only): 256 PERSISTENT workgroups, one per CU, with a static schedule.  In
// every pass the 32 workgroups of an XCD take one super-tile of 8 activation row panels x 4 weight column panels (slot s: row s / 4, column
// s % 4); super-tiles are enumerated column group fastest and an XCD's share of them is contiguous.  All tiles cost the same, so an XCD's
// workgroups move through their passes in step and the 32 co-resident tiles stream the SAME 8 + 4 operand panels through the XCD's 4 MB L2:
// (8 + 4) / 32 panel fetches per tile instead of 1 + 1 / n_tiles under the hardware's own dispatch (one workgroup per tile, column tiles of a
// row panel side by side).  What was measured on the way (tools/ab_t256_walk.sh, 1 005 000 rows, rocprofv3 --pmc FETCH_SIZE):
//   * the fabric traffic VERDICT r4 flagged (2.65 x algorithmic) is NOT activations re-fetched by column tiles that miss each other in L2: the
//     same 1-D tile order as a persistent lock-step schedule (every panel's column tiles start together) fetched the same bytes (81.1 vs
//     82.2 M KiB) in the same time.  It is the WEIGHTS: 3072 x 3072 f16 = 18.9 MB do not fit the 4 MB L2, every row panel streams them again
//     from the Infinity Cache (3926 panels x 18.9 MB = 74 GB worst case, 53 GB measured); at 1024 x 1024 (2 MB of weights) the traffic is
//     1.55 x algorithmic;
//   * the transposed 1-D walk (consecutive workgroups share a weight panel) trades that for activations streamed 12 times from HBM: 76 GB,
//     +11 % time;
//   * this 2-D walk halves the fetched bytes of the two shapes together (92.6 -> 45.9 M KiB) and runs 3072 x 3072 2.5 % faster (16.03 ->
//     15.63 ms, 1184 -> 1214 TFLOP/s, both rounds); 1024 x 1024 is unchanged (2.05 ms).  The chip is power-limited in this kernel: the
//     bytes that no longer cross the fabric come back as clock.
template <typename TO, bool DIRECT, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void conv_gemm_f16_t256_kernel(const sd_conv_args p, const int vec, const int super_walk) {
  constexpr int TBK = 64;
  constexpr int TROW = 128;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
#ifdef SD_STAMP
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime(), r_entry = __builtin_amdgcn_s_memrealtime();
#endif
  // Only the pass counter lives across a tile: the schedule and every lane constant of the body (staging roles, fragment offsets, row
  // pointers) are recomputed per tile from an opaque copy of the thread id, so that nothing else occupies a register through the epilogue.
  // (The LDS-staged epilogues keep the kernel arguments live in ~100 more SGPRs than there are: their walk is one pass, known at compile time.)
  for (int pass = 0; DIRECT || pass < 1; ++pass) {
  int tile_m, tile_n;
  const int n_tiles = (p.cout + TBN - 1) / TBN;
  if (super_walk) {
    const int m_tiles = (p.M + TBM - 1) / TBM;
    const int n_cg = n_tiles >> 2, n_rg = (m_tiles + 7) >> 3;
    const int nst = n_cg * n_rg, b = blockIdx.x;
    const int q = nst >> 3, r = nst & 7, xcd = b & 7, slot = b >> 3;
    const int s_lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int s_cnt = q + (xcd < r ? 1 : 0);
    if (pass >= s_cnt) break;
    const int st = s_lo + pass;
    tile_m = 8 * (st / n_cg) + (slot >> 2);
    tile_n = 4 * (st % n_cg) + (slot & 3);
    if (tile_m >= m_tiles) continue;              // a ragged last row group: this slot idles for the pass (a pass has no barrier between workgroups)
  } else {
    // the hardware's dispatch, one workgroup per tile: XCD x (= blockIdx % 8) owns a contiguous range of the tile list
    if (pass > 0) break;
    const int nwg = (int)gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    tile_n = wg % n_tiles;
    tile_m = wg / n_tiles;
  }
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));           // opaque per pass: the lane constants below are not hoisted out of the walk
  const int lane = tid & 63;
#ifdef SD_T256_UNIFORM_WID
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform for the compiler: LDS piece addresses (M0) on the scalar unit
#else
  const int wid = tid >> 6;
#endif
  const int wm = wid >> 2, wn = wid & 3;

  const int m0 = tile_m * TBM, n0 = tile_n * TBN;

  // staging role: within its group of 256 threads, thread (r0 = lt / 8, ps = lt % 8) fills physical 16-byte slot ps of rows
  // r0 + 32 i (i < 8) of ITS operand: waves 0-3 the weights, waves 4-7 the activations.  The slot holds logical k chunk
  // ps ^ ((row >> 1) & 7), and (row >> 1) & 7 does not depend on i.
  const bool bload = __builtin_amdgcn_readfirstlane(wid) < 4;
  const int slot2 = (__builtin_amdgcn_readfirstlane(wid) >> 1) & 1;       // SD_T256_STAGGER: which of its half's two DMA slots this wave uses
  const int lt = tid & 255;
  const int r0 = lt >> 3;
  const int ls8 = ((lt & 7) ^ ((r0 >> 1) & 7)) * 8;
  const int ktot = p.taps * p.cin_pad;
  const int nk = p.taps * (p.cin_pad / TBK);
  const int half = p.taps / 2;
  const _Float16* ptr[8];
  const _Float16* X = static_cast<const _Float16*>(p.x) + p.a_col0;
  auto set_tap = [&](int tap) {
    const int delta = (tap - half) * p.dil;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int m = m0 + r0 + 32 * i;
      m = m < p.M ? m : p.M - 1;
      const int seg = (m / p.T) * p.T;
      int tt = m - seg + delta;
      tt = tt < 0 ? -tt : tt;
      tt = tt >= p.T ? 2 * (p.T - 1) - tt : tt;
      ptr[i] = X + (size_t)(seg + tt) * p.lda;
    }
  };
  if (bload) {
    const _Float16* W = static_cast<const _Float16*>(p.w);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int n = n0 + r0 + 32 * i;
      n = n < p.cout ? n : p.cout - 1;
      ptr[i] = W + (size_t)n * ktot + ls8;
    }
  } else {
    set_tap(0);
  }
  int ld_tap = 0, ld_c0 = 0;
  char* const dst = smem_raw + ((wid & 3) * 8) * TROW;
  auto issue_b = [&](int st) {
    char* base = dst + R3_B_BASE + st * R3_B_STAGE;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#ifdef SD_DIAG_B_TO_REGS      // timing-only diagnostic: the weight pieces as plain 16-byte loads into (discarded) registers: the same
      // address-path traffic without the LDS write; the MFMAs then read stale weights (results are wrong by construction)
      f32x4 sink;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sink) : "v"(ptr[i]) : "memory");
      (void)base;
#else
      SD_GLDS16(ptr[i], base + i * 32 * TROW);
#endif
      ptr[i] += TBK;
    }
  };
  auto issue_a = [&](int st) {
    char* base = dst + st * R3_A_STAGE;
    const int col = ld_c0 + ls8;
    const int acol = col < p.cin ? col : 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) SD_GLDS16(ptr[i] + acol, base + i * 32 * TROW);
    ld_c0 += TBK;
    if (ld_c0 >= p.cin_pad) {
      ld_c0 = 0;
      ++ld_tap;
      if (ld_tap < p.taps) set_tap(ld_tap);
    }
  };

  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  const int fr = lane & 15, fq = lane >> 4;
  const int ab_sw = (fr >> 1) & 7;
  const char* const a_base = smem_raw + (wm * 128 + fr) * TROW;
  const char* const b_base = smem_raw + R3_B_BASE + (wn * 64 + fr) * TROW;
  // byte offset of this lane's 16 bytes inside a row, for k slice 0 / 1
  const int so0 = (fq ^ ab_sw) << 4, so1 = ((4 + fq) ^ ab_sw) << 4;

  h8 fa0[4], fa1[4], fb0[4], fb1[4];
#define P3_READ_A(dst_, stage_, so_, g_)                                                                        \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
      dst_[i] = *reinterpret_cast<const h8*>(a_base + (stage_) * R3_A_STAGE + (4 * (g_) + i) * 16 * TROW + (so_))
#define P3_READ_B(dst_, stage_, so_)                                                                            \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                 \
      dst_[j] = *reinterpret_cast<const h8*>(b_base + (stage_) * R3_B_STAGE + j * 16 * TROW + (so_))
#define P3_MMA(g_, a_, b_)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                               \
        acc[4 * (g_) + i][j] = DIRECT ? __builtin_amdgcn_mfma_f32_16x16x32_f16(b_[j], a_[i], acc[4 * (g_) + i][j], 0, 0, 0)   \
                                      : __builtin_amdgcn_mfma_f32_16x16x32_f16(a_[i], b_[j], acc[4 * (g_) + i][j], 0, 0, 0)

#ifdef SD_STAMP
  const unsigned long long t_loop0 = __builtin_amdgcn_s_memtime();
#endif
  if (bload) {
    issue_b(0);
  } else {
    issue_a(0);
    if (nk > 1) issue_a(1);
  }
  int sa = 0, sb = 0;                      // stages of step kt
  if constexpr (SPLIT) {
    for (int kt = 0; kt < nk; ++kt) {
      if (bload || kt + 1 >= nk) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int sa2 = sa == 0 ? 2 : sa - 1;
      P3_READ_A(fa0, sa, so1, 0);                                            // lo(g0)
      P3_READ_B(fb0, sb, so0);                                               // Bhi
      if (kt > 0) { P3_MMA(1, fa1, fb1); }                                   // deferred hi(g1).Blo of step kt - 1
      if (bload && kt + 1 < nk) issue_b(sb ^ 1);
#ifndef SD_SPLIT_A_SLOT
#define SD_SPLIT_A_SLOT 0        // behind which of the step's half-slices the activation waves issue their pieces (A/B builds)
#endif
      P3_READ_A(fa1, sa, so0, 0);                                            // hi(g0)
      P3_MMA(0, fa0, fb0);                                                   // lo(g0).Bhi
      if (SD_SPLIT_A_SLOT == 0 && !bload && kt + 2 < nk) issue_a(sa2);
      P3_READ_B(fb1, sb, so1);                                               // Blo
      P3_READ_A(fa0, sa, so1, 1);                                            // lo(g1)
      P3_MMA(0, fa1, fb0);                                                   // hi(g0).Bhi
      if (SD_SPLIT_A_SLOT == 1 && !bload && kt + 2 < nk) issue_a(sa2);
      P3_MMA(0, fa1, fb1);                                                   // hi(g0).Blo
      if (SD_SPLIT_A_SLOT == 2 && !bload && kt + 2 < nk) issue_a(sa2);
      P3_READ_A(fa1, sa, so0, 1);                                            // hi(g1)
      P3_MMA(1, fa0, fb0);                                                   // lo(g1).Bhi
      if (SD_SPLIT_A_SLOT == 3 && !bload && kt + 2 < nk) issue_a(sa2);
      P3_MMA(1, fa1, fb0);                                                   // hi(g1).Bhi
      if (SD_SPLIT_A_SLOT == 4 && !bload && kt + 2 < nk) issue_a(sa2);
      sa = sa == 2 ? 0 : sa + 1;
      sb ^= 1;
    }
  } else {
  for (int kt = 0; kt < nk; ++kt) {
    // ---- barrier(kt): this step's stages have landed, every wave has finished reading the previous step's
    if (bload || kt + 1 >= nk) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");      // the activations of step kt + 1 stay in flight
    __builtin_amdgcn_s_barrier();
    const int sa2 = sa == 0 ? 2 : sa - 1;                                  // (sa + 2) % 3 = the A stage of step kt - 1
    P3_READ_A(fa0, sa, so0, 0);
    P3_READ_B(fb0, sb, so0);
    if (kt > 0) { P3_MMA(1, fa1, fb1); }                                   // deferred half-slice 3 of step kt - 1
#ifdef SD_T256_STAGGER
    // DMA issue slots by wave PAIR: at most two waves (on different SIMDs) push pieces into the CU's one address path at a
    // time, so a wave is blocked for 8 x ~32 cycles instead of 8 x ~64-100 while four issue together
    if (bload && slot2 == 0 && kt + 1 < nk) issue_b(sb ^ 1);
    P3_READ_A(fa1, sa, so0, 1);
    P3_MMA(0, fa0, fb0);
    if (bload && slot2 == 1 && kt + 1 < nk) issue_b(sb ^ 1);
    P3_READ_A(fa0, sa, so1, 0);
    P3_READ_B(fb1, sb, so1);
    P3_MMA(1, fa1, fb0);
    if (!bload && slot2 == 0 && kt + 2 < nk) issue_a(sa2);
    P3_READ_A(fa1, sa, so1, 1);
    P3_MMA(0, fa0, fb1);
    if (!bload && slot2 == 1 && kt + 2 < nk) issue_a(sa2);
#else
    if (bload && kt + 1 < nk) issue_b(sb ^ 1);
    // ---- half-slice 0
    P3_READ_A(fa1, sa, so0, 1);
    P3_MMA(0, fa0, fb0);
    if (!bload && kt + 2 < nk) issue_a(sa2);
    // ---- half-slice 1
    P3_READ_A(fa0, sa, so1, 0);
    P3_READ_B(fb1, sb, so1);
    P3_MMA(1, fa1, fb0);
    // ---- half-slice 2 (half-slice 3 runs behind the next barrier)
    P3_READ_A(fa1, sa, so1, 1);
    P3_MMA(0, fa0, fb1);
#endif
    sa = sa == 2 ? 0 : sa + 1;
    sb ^= 1;
  }
  }
  P3_MMA(1, fa1, fb1);
#undef P3_READ_A
#undef P3_READ_B
#undef P3_MMA
#ifdef SD_STAMP
  const unsigned long long t_loop1 = __builtin_amdgcn_s_memtime();
#endif
  if constexpr (DIRECT) {
    sd_direct_epilogue<TO>(p, acc, m0 + wm * 128, n0 + wn * 64, fr, fq);
  } else {
  __syncthreads();
  
    float* Cs = reinterpret_cast<float*>(smem_raw);
  #pragma unroll
    for (int hm = 0; hm < 2; ++hm) {
      if (wm == hm) {
  #pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int cl = wn * 64 + ni * 16 + fr;
  #pragma unroll
          for (int mi = 0; mi < 8; ++mi) {
  #pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(mi * 16 + fq * 4 + r) * TBN + cl] = acc[mi][ni][r];
          }
        }
      }
      __syncthreads();
      sd_store_tile<TO, TBM / 2, TBN, 512, 1, 2, SPLIT>(p, Cs, TBN, m0 + hm * (TBM / 2), n0, tid, vec);      // (SPLIT: y may be SD_DT_SPLIT16)
      __syncthreads();
    }
}
  // the next tile's first DMA pieces land in stages 0 / 1: not before every wave has read its last fragments of this tile
  if (super_walk) __syncthreads();
#ifndef SD_STAMP
  }   // tile walk
#endif
#ifdef SD_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t_exit = __builtin_amdgcn_s_memtime();
  if (tid == 0 && blockIdx.x < 8192) {     // [prologue, K loop, epilogue incl. store drain, total] cycles, total in 100 MHz ticks
    sd_stamp_buf[blockIdx.x * 8 + 0] = t_loop0 - t_entry;
    sd_stamp_buf[blockIdx.x * 8 + 1] = t_loop1 - t_loop0;
    sd_stamp_buf[blockIdx.x * 8 + 2] = t_exit - t_loop1;
    sd_stamp_buf[blockIdx.x * 8 + 3] = t_exit - t_entry;
    sd_stamp_buf[blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime() - r_entry;
  }
  }   // tile walk (stamp builds: the stamps are those of the workgroup's last tile)
#endif
}

// ------------------------------------------------------------------------------------------
// EXPERIMENT, compiled only into variant builds (build_native.py --variant w4 "-DSD_WITH_W4", selected at run time with
// SD_EXPERIMENT=1 SD_F16_W4=1): the same 256x256 tile and LDS-DMA ring with FOUR waves of 128 x 128.  A K step's
// fragment reads drop from 8 x 24 KB to 4 x 32 KB (the 8-wave form moves 192 KB of reads + 64 KB of DMA writes per step through a
// 128 B/clk LDS for 2048 cycles of matrix work), at the price of one wave per SIMD: its 256 accumulator registers live in AGPRs
// (200 VGPRs + 256 AGPRs, no spill, no v_accvgpr copy in the loop), nothing hides a stall but the wave's own instruction stream.
// f16 operands, register (DIRECT) epilogue only.  Per step and wave: 128 MFMAs, 32 ds_read_b128 in two sets of 64 registers
// (k slice 0 / 1), 16 DMA pieces (the weights of step k + 1, the activations of step k + 2); the barrier sits in front of the
// step's LAST 64 MFMAs.  The loop body is straight-line and pinned chunk by chunk with sched_barrier (ISA: MMMMMMMM RRR D x 16 per
// step, two waits, one barrier).  Passes tests/test_gpu_f16.py (158 tests).  Measured against the shipped 8-wave kernel, same box,
// 1 005 000 rows: 1024 x 1024 871-873 vs 900-912 TFLOP/s, 3072 x 3072 1147-1149 vs 1120-1121 (first cut, compiler-scheduled: 858 /
// 1159 vs 898 / 1151).  Not shipped: a wave that is alone on its SIMD stands still whenever one of its 16 DMA pieces waits for a
// slot in the CU's vector-memory queue (round 2: 70-100 cycles per piece with four waves issuing), and with 456 registers per wave
// there is no room for loader waves beside it.
#ifdef SD_WITH_W4
template <typename TO>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_gemm_f16_w4_kernel(const sd_conv_args p, const int vec) {
  constexpr int TBK = 64;
  constexpr int TROW = 128;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int n_tiles = (p.cout + TBN - 1) / TBN;
  int wg;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int tile_n = wg % n_tiles, tile_m = wg / n_tiles;
  const int m0 = tile_m * TBM, n0 = tile_n * TBN;

  // staging role: thread (r0 = tid / 8, ps = tid % 8) fills physical 16-byte slot ps of rows r0 + 32 i (i < 8) of BOTH operands.
  // Addresses = a workgroup-uniform base (SGPRs) + a 32-bit per-lane element offset: 16 address registers instead of 32.
  const int r0 = tid >> 3;
  const int ls8 = ((tid & 7) ^ ((r0 >> 1) & 7)) * 8;
  const int ktot = p.taps * p.cin_pad;
  const int nk = p.taps * (p.cin_pad / TBK);
  const int half = p.taps / 2;
  int m0c = m0 < p.M ? m0 : p.M - 1;
  const _Float16* const Ab = static_cast<const _Float16*>(p.x) + p.a_col0 + (size_t)m0c * p.lda;      // row m0 of the activations
  int n0c = n0 < p.cout ? n0 : p.cout - 1;
  const _Float16* Wb = static_cast<const _Float16*>(p.w) + (size_t)n0c * ktot;                        // row n0 of the weights, advanced per K step
  int pa[8], pb[8];
  auto set_tap = [&](int tap) {
    const int delta = (tap - half) * p.dil;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int m = m0 + r0 + 32 * i;
      m = m < p.M ? m : p.M - 1;
      const int seg = (m / p.T) * p.T;
      int tt = m - seg + delta;
      tt = tt < 0 ? -tt : tt;
      tt = tt >= p.T ? 2 * (p.T - 1) - tt : tt;
      pa[i] = (seg + tt - m0c) * p.lda;                 // |rows| < 256 + T: fits 32 bits for any lda this library accepts
    }
  };
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int n = n0 + r0 + 32 * i;
    n = n < p.cout ? n : p.cout - 1;
    pb[i] = (n - n0c) * ktot + ls8;
  }
  set_tap(0);
  int ld_tap = 0, ld_c0 = 0;
  char* const dst = smem_raw + (wid * 8) * TROW;
  auto issue_b = [&](int st) {
    char* base = dst + R3_B_BASE + st * R3_B_STAGE;
#pragma unroll
    for (int i = 0; i < 8; ++i) SD_GLDS16(Wb + pb[i], base + i * 32 * TROW);
    Wb += TBK;
  };
  auto issue_a = [&](int st) {
    char* base = dst + st * R3_A_STAGE;
    const int col = ld_c0 + ls8;
    const int acol = col < p.cin ? col : 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) SD_GLDS16(Ab + (pa[i] + acol), base + i * 32 * TROW);
    ld_c0 += TBK;
    if (ld_c0 >= p.cin_pad) {
      ld_c0 = 0;
      ++ld_tap;
      if (ld_tap < p.taps) set_tap(ld_tap);
    }
  };

  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v acc0[8][4], acc1[8][4];          // [16-row time tile][16-channel tile]: channels 0-63 / 64-127 of the wave's 128
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
  const int fr = lane & 15, fq = lane >> 4;
  const int ab_sw = (fr >> 1) & 7;
  const char* const a_base = smem_raw + (wm * 128 + fr) * TROW;
  const char* const b_base = smem_raw + R3_B_BASE + (wn * 128 + fr) * TROW;
  const int so0 = (fq ^ ab_sw) << 4, so1 = ((4 + fq) ^ ab_sw) << 4;

  h8 xa[8], xb[8], ya[8], yb[8];          // fragment sets of k slice 0 (x) and 1 (y)
#define W4_READ(fa_, fb_, sa_, sb_, so_)                                                                          \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                 \
    fa_[i] = *reinterpret_cast<const h8*>(a_base + (sa_) * R3_A_STAGE + i * 16 * TROW + (so_));                   \
    fb_[i] = *reinterpret_cast<const h8*>(b_base + (sb_) * R3_B_STAGE + i * 16 * TROW + (so_));                   \
  }
#define W4_MMA(fa_, fb_)                                                                                          \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                 \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                               \
      acc0[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb_[j], fa_[i], acc0[i][j], 0, 0, 0);                   \
      acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb_[4 + j], fa_[i], acc1[i][j], 0, 0, 0);               \
    }                                                                                                             \
  }

  issue_b(0);
  issue_a(0);
  if (nk > 1) issue_a(1);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { ya[i][e] = (_Float16)0.f; yb[i][e] = (_Float16)0.f; }     // the first step's "deferred" slice adds zeros
  int sa = 0, sb = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // ---- barrier(kt): this step's stages have landed, every wave holds the previous step's last fragments in registers
    if (kt + 1 >= nk) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");        // the activations of step kt + 1 stay in flight
    __builtin_amdgcn_s_barrier();
    const int sa2 = sa == 0 ? 2 : sa - 1;                                    // the A stage of step kt - 1
    // One straight-line body, instruction order pinned chunk by chunk (left to itself hipcc sinks the 16 fragment reads of a half
    // step behind ~58 of the 64 MFMAs they should run under and waits for them right after): a chunk = the two fragment reads of row
    // tile i of the NEXT k slice, one DMA piece, and the 8 MFMAs of row tile i of the slice whose fragments are complete.  Past the
    // last step the DMA pieces are re-reads of valid addresses into stages nobody reads (no branch in the body).
    const bool hb = kt + 1 < nk, ha = kt + 2 < nk;
    const char* const ar = a_base + sa * R3_A_STAGE;
    const char* const br = b_base + sb * R3_B_STAGE;
    char* const bdst = dst + R3_B_BASE + (sb ^ 1) * R3_B_STAGE;
    char* const adst = dst + sa2 * R3_A_STAGE;
    const int col = ld_c0 + ls8;
    const int acol = (ha && col < p.cin) ? col : 0;
    const _Float16* const Wk = hb ? Wb : Wb - TBK;
    // chunk i = the 8 MFMAs of row tile i, then (chunks 0-5 only) fragment reads of the next slice: 3, 3, 3, 3, 2, 2 — the last two
    // chunks' MFMAs cover the latency of the last reads, so the next half step starts without a wait
#define W4_Q(k_) ((k_) < 8 ? 8 + (k_) : (k_) - 8)          /* k-th fragment read: all 8 b, then a[0..7], the order the consumer needs them */
#define W4_RD(fa_, fb_, q_, so_)                                                                              \
  do {                                                                                                       \
    if ((q_) < 8) fa_[(q_)] = *reinterpret_cast<const h8*>(ar + (q_) * 16 * TROW + (so_));                    \
    else fb_[(q_) - 8] = *reinterpret_cast<const h8*>(br + ((q_) - 8) * 16 * TROW + (so_));                   \
  } while (0)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {                                            // deferred k slice 1 of step kt - 1
        acc0[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yb[j], ya[i], acc0[i][j], 0, 0, 0);
        acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yb[4 + j], ya[i], acc1[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (i < 4) { W4_RD(xa, xb, W4_Q(3 * i), so0); W4_RD(xa, xb, W4_Q(3 * i + 1), so0); W4_RD(xa, xb, W4_Q(3 * i + 2), so0); }
      else if (i < 6) { W4_RD(xa, xb, W4_Q(12 + 2 * (i - 4)), so0); W4_RD(xa, xb, W4_Q(13 + 2 * (i - 4)), so0); }
      SD_GLDS16(Wk + pb[i], bdst + i * 32 * TROW);                             // the weights of step kt + 1
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc0[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[j], xa[i], acc0[i][j], 0, 0, 0);
        acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[4 + j], xa[i], acc1[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (i < 4) { W4_RD(ya, yb, W4_Q(3 * i), so1); W4_RD(ya, yb, W4_Q(3 * i + 1), so1); W4_RD(ya, yb, W4_Q(3 * i + 2), so1); }
      else if (i < 6) { W4_RD(ya, yb, W4_Q(12 + 2 * (i - 4)), so1); W4_RD(ya, yb, W4_Q(13 + 2 * (i - 4)), so1); }
      SD_GLDS16(Ab + (pa[i] + acol), adst + i * 32 * TROW);                    // the activations of step kt + 2
    }
#undef W4_RD
#undef W4_Q
    __builtin_amdgcn_sched_barrier(0);
    if (hb) Wb += TBK;
    if (ha) {
      ld_c0 += TBK;
      if (ld_c0 >= p.cin_pad) {
        ld_c0 = 0;
        ++ld_tap;
        if (ld_tap < p.taps) set_tap(ld_tap);
      }
    }
    sa = sa == 2 ? 0 : sa + 1;
    sb ^= 1;
  }
  W4_MMA(ya, yb);
#undef W4_READ
#undef W4_MMA
  sd_direct_epilogue<TO>(p, acc0, m0 + wm * 128, n0 + wn * 128, fr, fq);
  sd_direct_epilogue<TO>(p, acc1, m0 + wm * 128, n0 + wn * 128 + 64, fr, fq);
}

template <typename TO>
int launch_w4(const sd_conv_args* a, int vec, hipStream_t stream) {
  const long tiles_m = (a->M + TBM - 1) / TBM;
  const long tiles_n = (a->cout + TBN - 1) / TBN;
  auto kern = conv_gemm_f16_w4_kernel<TO>;
  SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(kern), R3_LDS_BYTES));
  {
    SdProfScope prof(SD_PROF_CONV_WIDE, stream, 2.0 * (double)a->M * (double)a->cout * (double)a->taps * (double)a->cin);
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), R3_LDS_BYTES, stream, *a, vec);
  }
  SD_CHECK_LAUNCH("conv_gemm_f16_w4_kernel");
  return SD_OK;
}
#endif  // SD_WITH_W4

template <typename TO, bool DIRECT, bool SPLIT = false>
int launch_t256(const sd_conv_args* a, int vec, hipStream_t stream) {
  const long tiles_m = (a->M + TBM - 1) / TBM;
  const long tiles_n = (a->cout + TBN - 1) / TBN;
  auto kern = conv_gemm_f16_t256_kernel<TO, DIRECT, SPLIT>;
  SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(kern), R3_LDS_BYTES));
  // the super-tile walk (256 persistent workgroups, 8 x 4 tiles per XCD and pass) when the tile list is several rounds long, the chip has 8 x 32
  // CUs and the column tiles come in fours (sd_set_tuning(SD_TUNE_T256_LOCKSTEP_TILES): from how many tiles; default 4 x the CU count)
  static const int n_cu = [] { int dev = 0, n = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0; return n; }();
  const long total = tiles_m * tiles_n;
  long from = sd_t256_lockstep_tiles().load(std::memory_order_relaxed);
  if (from < 0) from = 4L * 256;
  const bool super = DIRECT && n_cu == 256 && tiles_n % 4 == 0 && total < (1L << 30) && total >= from;
  {
    // work = the algorithmic (f32-equivalent) flops: a split row carries cin / 2 values
    SdProfScope prof(SD_PROF_CONV_WIDE, stream, (SPLIT ? 1.0 : 2.0) * (double)a->M * (double)a->cout * (double)a->taps * (double)a->cin);
    if (super) hipLaunchKernelGGL(kern, dim3(256), dim3(512), R3_LDS_BYTES, stream, *a, vec, 1);
    else hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(512), R3_LDS_BYTES, stream, *a, vec, 0);
  }
  SD_CHECK_LAUNCH("conv_gemm_f16_t256_kernel");
  return SD_OK;
}

template <typename TA, typename TO>
int launch(const sd_conv_args* a, int vec, hipStream_t stream) {
  const long tiles_m = (a->M + BM - 1) / BM;
  const long tiles_n = (a->cout + BN - 1) / BN;
  auto kern = conv_gemm_f16_kernel<TA, TO>;
  SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(kern), STAGE_BYTES));
  {
    SdProfScope prof(SD_PROF_CONV_GEMM, stream, 2.0 * (double)a->M * (double)a->cout * (double)a->taps * (double)a->cin);
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), STAGE_BYTES, stream, *a, vec);
  }
  SD_CHECK_LAUNCH("conv_gemm_f16_kernel");
  return SD_OK;
}

}  // namespace

#ifdef SD_STAMP
extern "C" int sd_debug_read_stamps(unsigned long long* out, int n) {
  SD_CHECK_HIP(hipDeviceSynchronize());
  SD_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(sd_stamp_buf), (size_t)n * sizeof(unsigned long long)));
  return SD_OK;
}
#endif

std::atomic<long>& sd_t256_lockstep_tiles() {
  static std::atomic<long> v{-1L};
  return v;
}

std::atomic<long>& sd_f16_narrow_tiles() {
  static std::atomic<long> v{128L};
  return v;
}

extern "C" int sd_conv1d_cl_f16(const sd_conv_args* a, sd_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SD_CHECK_ARG(a != nullptr, "sd_conv1d_cl_f16: null args");
  SD_CHECK_ARG(a->w_dtype == SD_DT_F16, "sd_conv1d_cl_f16: weights must be packed f16 (w_dtype=%d)", a->w_dtype);
  SD_CHECK_ARG(a->x && a->w && a->y, "sd_conv1d_cl_f16: null x/w/y");
  SD_CHECK_ARG((a->x_dtype == SD_DT_F32 || a->x_dtype == SD_DT_F16) && (a->y_dtype == SD_DT_F32 || a->y_dtype == SD_DT_F16),
               "sd_conv1d_cl_f16: bad x_dtype/y_dtype %d/%d", a->x_dtype, a->y_dtype);
  SD_CHECK_ARG(a->M > 0 && a->T > 0 && a->M % a->T == 0, "sd_conv1d_cl_f16: M=%d must be a positive multiple of T=%d", a->M, a->T);
  SD_CHECK_ARG(a->cin > 0 && a->cin % 8 == 0, "sd_conv1d_cl_f16: cin=%d must be a positive multiple of 8", a->cin);
  SD_CHECK_ARG(a->cin_pad >= a->cin && a->cin_pad % BK == 0, "sd_conv1d_cl_f16: cin_pad=%d must be >= cin and a multiple of %d", a->cin_pad, BK);
  SD_CHECK_ARG(a->cout > 0, "sd_conv1d_cl_f16: cout=%d", a->cout);
  SD_CHECK_ARG(a->taps >= 1 && (a->taps & 1) && a->dil >= 1, "sd_conv1d_cl_f16: taps=%d (odd) dil=%d", a->taps, a->dil);
  SD_CHECK_ARG((a->taps / 2) * a->dil < a->T, "sd_conv1d_cl_f16: reflect padding %d needs T > pad (T=%d)", (a->taps / 2) * a->dil, a->T);
  SD_CHECK_ARG(a->lda % 8 == 0 && a->a_col0 % 8 == 0 && a->a_col0 + a->cin <= a->lda,
               "sd_conv1d_cl_f16: lda=%d a_col0=%d cin=%d (need multiples of 8, slice inside row)", a->lda, a->a_col0, a->cin);
  SD_CHECK_ARG(a->o_col0 >= 0 && a->o_col0 + a->cout <= a->ldo, "sd_conv1d_cl_f16: output slice outside row");
  SD_CHECK_ARG(sd_aligned16(a->x) && sd_aligned16(a->w), "sd_conv1d_cl_f16: x and w must be 16-byte aligned");
  if (a->tee) {
    SD_CHECK_ARG(a->tee_lo >= 0 && a->tee_lo < a->tee_hi && a->tee_hi <= a->cout && a->tee_hi - a->tee_lo <= a->ldt,
                 "sd_conv1d_cl_f16: bad tee range [%d,%d) ldt=%d", a->tee_lo, a->tee_hi, a->ldt);
    if (a->tee_add)
      SD_CHECK_ARG(a->ta_col0 >= 0 && a->ta_col0 + (a->tee_hi - a->tee_lo) <= a->ld_ta, "sd_conv1d_cl_f16: tee_add slice outside row");
  }
  const long tiles = (long)((a->M + BM - 1) / BM) * ((a->cout + BN - 1) / BN);
  SD_CHECK_ARG(tiles < (1L << 31), "sd_conv1d_cl_f16: grid too large");
  int vec = a->cout % 8 == 0 && a->ldo % 8 == 0 && a->o_col0 % 8 == 0 && sd_aligned16(a->y);
  vec = vec && sd_aligned16(a->bias) && sd_aligned16(a->scale) && sd_aligned16(a->shift);   // null is aligned
  if (a->tee) {
    vec = vec && a->tee_lo % 8 == 0 && a->tee_hi % 8 == 0 && a->ldt % 8 == 0 && sd_aligned16(a->tee);
    if (a->tee_add) vec = vec && a->ld_ta % 8 == 0 && a->ta_col0 % 8 == 0 && sd_aligned16(a->tee_add);
  }
  if (a->colstat) {
    const bool simple = (a->act == SD_ACT_RELU || a->act == SD_ACT_NONE) && a->act2 == SD_ACT_NONE && !a->bias_per_seg;
    if (!(vec && simple && a->T >= 64 && a->cout % 256 == 0 && !a->tee))
      return sd_set_error(SD_ERR_UNSUPPORTED, "sd_conv1d_cl_f16: colstat needs T >= 64, cout %% 256 == 0, relu/identity, per-channel bias, "
                          "aligned slices and no tee (T=%d cout=%d act=%d/%d)", a->T, a->cout, a->act, a->act2);
  }
  const bool xa = a->x_dtype == SD_DT_F16, ya = a->y_dtype == SD_DT_F16;
  // Kernel choice, measured per shape on MI355X (tools/probe_conv.py, B*T = 1 005 000 rows): the 256x256
  // LDS-DMA kernel wins wherever the output is wide (3072x3072: 987 vs ~800 TFLOP/s, 1024x1024: 789 vs
  // 726, 128->3072: 342 vs 315); the register-staged 128x128 kernel with two workgroups per CU is faster
  // on the narrow outputs (Res2Net 128x384: 555, 3072->128: 530 vs 408) and is the only one with the
  // tee_add epilogue.  SD_F16_KERNEL=reg|t256 forces one kernel for A/B runs.
  static const int forced = [] {
    const char* e = sd_experiment_env("SD_F16_KERNEL");
    if (!e) return -1;
    return e[0] == 'r' ? 0 : e[0] == 't' ? 2 : -1;
  }();
  // ... except for small launches of the C-wide layers (round 3, tools/sweep_f16.py: 1024 -> 1024 at 16 / 32 segments 0.024 / 0.030 ms
  // on the 128x128 kernel against 0.035 / 0.038; from 64 segments up, and for 3C -> 3C always, the 256x256 kernel wins): at most 128
  // tiles of 256x256 -> two workgroups of 128x128 per CU fill the chip better than a fraction of one round of big tiles
  const long t256 = (long)((a->M + TBM - 1) / TBM) * ((a->cout + TBN - 1) / TBN);
  const bool wide = a->cout >= 1024 && !(a->cout <= 1024 && t256 <= sd_f16_narrow_tiles().load(std::memory_order_relaxed));
  const int choice = forced >= 0 ? forced : (wide ? 2 : 0);
  // (the 256x256 kernel: no tee_add epilogue, and column statistics only for tiles that span <= 2 segments)
  if (xa && choice == 2 && !(a->tee && a->tee_add) && !(a->colstat && a->T < 128)) {
    static const bool direct_ok = [] {     // SD_T256_DIRECT=0: the LDS-staged epilogue for every layer (A/B runs)
      const char* e = sd_experiment_env("SD_T256_DIRECT");
      return !(e && e[0] == '0');
    }();
    const bool plain = vec && (a->act == SD_ACT_RELU || a->act == SD_ACT_NONE) && a->act2 == SD_ACT_NONE && !a->bias_per_seg && !a->tee;
#ifdef SD_WITH_W4
    static const bool w4 = [] { const char* e = sd_experiment_env("SD_F16_W4"); return e && e[0] == '1'; }();     // A/B: the 4-wave form
    if (plain && direct_ok && w4) return ya ? launch_w4<_Float16>(a, vec, stream) : launch_w4<float>(a, vec, stream);
#endif
    if (plain && direct_ok) return ya ? launch_t256<_Float16, true>(a, vec, stream) : launch_t256<float, true>(a, vec, stream);
    return ya ? launch_t256<_Float16, false>(a, vec, stream) : launch_t256<float, false>(a, vec, stream);
  }
  if (xa && ya) return launch<_Float16, _Float16>(a, vec, stream);
  if (xa && !ya) return launch<_Float16, float>(a, vec, stream);
  if (!xa && ya) return launch<float, _Float16>(a, vec, stream);
  return launch<float, float>(a, vec, stream);
}

// ------------------------------------------------------------------------------------------ f32-split16x3
namespace {

// f32 [M][ldx] columns [col0, col0 + C) -> split-packed rows: value column c at halfs 64 (c / 32) + (c % 32) (hi = f16(v)) and
// + 32 (lo = f16(v - hi)); columns C .. Cp - 1 (Cp = C rounded up to 32) are zero.  |v| is clamped to the f16 range first
// (activations of this network are O(1..100); a value beyond 65504 would otherwise become inf - inf = NaN).
__global__ __launch_bounds__(256) void split16_pack_kernel(const float* __restrict__ x, int ldx, int col0, long M, int C, int Cp,
                                                           _Float16* __restrict__ out, long ldo_halfs, float mul) {
  const int groups = Cp / 8;                                   // 8 values per thread
  const long total = M * groups;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / groups;
    const int c = (int)(i - m * groups) * 8;
    float v[8];
    if (c + 8 <= C) {
      const float* src = x + (size_t)m * ldx + col0 + c;
      const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = c + e < C ? x[(size_t)m * ldx + col0 + c + e] : 0.f;
    }
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float w = sd_split16_clamp(v[e] * mul);
      hi[e] = (_Float16)w;
      lo[e] = (_Float16)(w - (float)hi[e]);
    }
    _Float16* dst = out + (size_t)m * ldo_halfs + 64 * (c / 32) + (c % 32);
    *reinterpret_cast<h8*>(dst) = hi;
    *reinterpret_cast<h8*>(dst + 32) = lo;
  }
}

}  // namespace

extern "C" int sd_split16_pack_f32(const float* x, int ldx, int col0, int M, int C, float mul, void* out, int ldo, sd_stream_t stream) {
  SD_CHECK_ARG(x && out && M > 0 && C > 0, "sd_split16_pack_f32: null pointer or empty shape");
  const int Cp = (C + 31) / 32 * 32;
  SD_CHECK_ARG(ldo >= Cp && ldo % 32 == 0, "sd_split16_pack_f32: ldo=%d must be a multiple of 32 and >= %d", ldo, Cp);
  SD_CHECK_ARG(col0 >= 0 && col0 + C <= ldx, "sd_split16_pack_f32: slice outside row");
  SD_CHECK_ARG(sd_aligned16(x) && sd_aligned16(out) && ldx % 4 == 0 && col0 % 4 == 0, "sd_split16_pack_f32: x / out must be 16-byte aligned, ldx and col0 multiples of 4");
  const long total = (long)M * (Cp / 8);
  const long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(split16_pack_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     x, ldx, col0, (long)M, C, Cp, static_cast<_Float16*>(out), (long)ldo * 2, mul);
  SD_CHECK_LAUNCH("split16_pack_kernel");
  return SD_OK;
}


// x plain f32 (split while staging): the 128x128 kernel
static int conv1d_cl_split16_narrow(const sd_conv_args* a, hipStream_t stream) {
  SD_CHECK_ARG(a->w_dtype == SD_DT_SPLIT16 && (a->y_dtype == SD_DT_F32 || a->y_dtype == SD_DT_SPLIT16), "sd_conv1d_cl_split16: w must be SD_DT_SPLIT16, y f32 or SD_DT_SPLIT16");
  SD_CHECK_ARG(a->M > 0 && a->T > 0 && a->M % a->T == 0, "sd_conv1d_cl_split16: M=%d must be a positive multiple of T=%d", a->M, a->T);
  SD_CHECK_ARG(a->cin > 0 && a->cin % 4 == 0 && a->cin_pad >= a->cin && a->cin_pad % 32 == 0, "sd_conv1d_cl_split16: cin=%d (a multiple of 4) cin_pad=%d (of 32)", a->cin, a->cin_pad);
  SD_CHECK_ARG(a->cout > 0 && a->taps >= 1 && (a->taps & 1) && a->dil >= 1 && (a->taps / 2) * a->dil < a->T, "sd_conv1d_cl_split16: cout=%d taps=%d dil=%d T=%d", a->cout, a->taps, a->dil, a->T);
  SD_CHECK_ARG(a->lda % 4 == 0 && a->a_col0 % 4 == 0 && a->a_col0 + a->cin <= a->lda && sd_aligned16(a->x) && sd_aligned16(a->w),
               "sd_conv1d_cl_split16: f32 x needs lda / a_col0 multiples of 4, the slice inside the row, 16-byte aligned x and w");
  SD_CHECK_ARG(a->o_col0 >= 0 && a->o_col0 + a->cout <= a->ldo, "sd_conv1d_cl_split16: output slice outside row");
  SD_CHECK_ARG(!a->colstat, "sd_conv1d_cl_split16: column statistics come from the 256x256 kernel (SD_DT_SPLIT16 x) only");
  if (a->tee) {
    SD_CHECK_ARG(a->tee_lo >= 0 && a->tee_lo < a->tee_hi && a->tee_hi <= a->cout && a->tee_hi - a->tee_lo <= a->ldt,
                 "sd_conv1d_cl_split16: bad tee range [%d,%d) ldt=%d", a->tee_lo, a->tee_hi, a->ldt);
    if (a->tee_add)
      SD_CHECK_ARG(a->ta_col0 >= 0 && a->ta_col0 + (a->tee_hi - a->tee_lo) <= a->ld_ta, "sd_conv1d_cl_split16: tee_add slice outside row");
  }
  const long tiles_m = (a->M + BM - 1) / BM, tiles_n = (a->cout + BN - 1) / BN;
  SD_CHECK_ARG(tiles_m * tiles_n < (1L << 31), "sd_conv1d_cl_split16: grid too large");
  int vec = a->cout % 8 == 0 && a->ldo % 8 == 0 && a->o_col0 % 8 == 0 && sd_aligned16(a->y);      // sd_store_tile's groups of 8 columns
  vec = vec && sd_aligned16(a->bias) && sd_aligned16(a->scale) && sd_aligned16(a->shift);
  if (a->tee) {
    vec = vec && a->tee_lo % 8 == 0 && a->tee_hi % 8 == 0 && a->ldt % 8 == 0 && sd_aligned16(a->tee);
    if (a->tee_add) vec = vec && a->ld_ta % 8 == 0 && a->ta_col0 % 8 == 0 && sd_aligned16(a->tee_add);
  }
  if (a->y_dtype == SD_DT_SPLIT16)      // y written as split halves by the shared epilogue's vector path (ldo in VALUE columns)
    SD_CHECK_ARG(vec && a->ldo % 32 == 0, "sd_conv1d_cl_split16: an SD_DT_SPLIT16 output needs ldo %% 32 == 0 and the aligned (vector) epilogue (ldo=%d o_col0=%d cout=%d)",
                 a->ldo, a->o_col0, a->cout);
  auto kern = conv_gemm_split16_n128_kernel<float>;
  SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(kern), STAGE_BYTES));
  {
    SdProfScope prof(SD_PROF_CONV_GEMM, stream, 2.0 * (double)a->M * (double)a->cout * (double)a->taps * (double)a->cin);
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), STAGE_BYTES, stream, *a, vec);
  }
  SD_CHECK_LAUNCH("conv_gemm_split16_n128_kernel");
  return SD_OK;
}

extern "C" int sd_conv1d_cl_split16(const sd_conv_args* a, sd_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SD_CHECK_ARG(a != nullptr, "sd_conv1d_cl_split16: null args");
  SD_CHECK_ARG(a->x && a->w && a->y, "sd_conv1d_cl_split16: null x/w/y");
  if (a->x_dtype == SD_DT_F32) return conv1d_cl_split16_narrow(a, stream);
  SD_CHECK_ARG(a->w_dtype == SD_DT_SPLIT16 && a->x_dtype == SD_DT_SPLIT16 && (a->y_dtype == SD_DT_F32 || a->y_dtype == SD_DT_SPLIT16),
               "sd_conv1d_cl_split16: x and w must be split-packed (SD_DT_SPLIT16), y f32 or SD_DT_SPLIT16 (got %d/%d/%d)", a->x_dtype, a->w_dtype, a->y_dtype);
  SD_CHECK_ARG(a->M > 0 && a->T > 0 && a->M % a->T == 0, "sd_conv1d_cl_split16: M=%d must be a positive multiple of T=%d", a->M, a->T);
  SD_CHECK_ARG(a->cin > 0 && a->cin_pad >= a->cin && a->cin_pad % 32 == 0, "sd_conv1d_cl_split16: cin=%d cin_pad=%d (a multiple of 32)", a->cin, a->cin_pad);
  SD_CHECK_ARG(a->cout > 0, "sd_conv1d_cl_split16: cout=%d", a->cout);
  SD_CHECK_ARG(a->taps >= 1 && (a->taps & 1) && a->dil >= 1, "sd_conv1d_cl_split16: taps=%d (odd) dil=%d", a->taps, a->dil);
  SD_CHECK_ARG((a->taps / 2) * a->dil < a->T, "sd_conv1d_cl_split16: reflect padding %d needs T > pad (T=%d)", (a->taps / 2) * a->dil, a->T);
  SD_CHECK_ARG(a->lda % 32 == 0 && a->a_col0 % 32 == 0 && a->a_col0 + a->cin_pad <= a->lda,
               "sd_conv1d_cl_split16: lda=%d a_col0=%d cin_pad=%d (value columns: multiples of 32, slice inside row)", a->lda, a->a_col0, a->cin_pad);
  SD_CHECK_ARG(a->o_col0 >= 0 && a->o_col0 + a->cout <= a->ldo, "sd_conv1d_cl_split16: output slice outside row");
  SD_CHECK_ARG(sd_aligned16(a->x) && sd_aligned16(a->w), "sd_conv1d_cl_split16: x and w must be 16-byte aligned");
  SD_CHECK_ARG(!(a->tee && a->tee_add), "sd_conv1d_cl_split16: the tee_add epilogue is not available on this kernel");
  if (a->tee)
    SD_CHECK_ARG(a->tee_lo >= 0 && a->tee_lo < a->tee_hi && a->tee_hi <= a->cout && a->tee_hi - a->tee_lo <= a->ldt,
                 "sd_conv1d_cl_split16: bad tee range [%d,%d) ldt=%d", a->tee_lo, a->tee_hi, a->ldt);
  const long tiles = (long)((a->M + TBM - 1) / TBM) * ((a->cout + TBN - 1) / TBN);
  SD_CHECK_ARG(tiles < (1L << 31), "sd_conv1d_cl_split16: grid too large");
  int vec = a->cout % 8 == 0 && a->ldo % 8 == 0 && a->o_col0 % 8 == 0 && sd_aligned16(a->y);
  vec = vec && sd_aligned16(a->bias) && sd_aligned16(a->scale) && sd_aligned16(a->shift);
  if (a->tee) vec = vec && a->tee_lo % 8 == 0 && a->tee_hi % 8 == 0 && a->ldt % 8 == 0 && sd_aligned16(a->tee);
  if (a->colstat) {
    const bool simple = (a->act == SD_ACT_RELU || a->act == SD_ACT_NONE) && a->act2 == SD_ACT_NONE && !a->bias_per_seg;
    if (!(vec && simple && a->T >= 128 && a->cout % 256 == 0 && !a->tee))
      return sd_set_error(SD_ERR_UNSUPPORTED, "sd_conv1d_cl_split16: colstat needs T >= 128, cout %% 256 == 0, relu/identity, per-channel bias, "
                          "aligned slices and no tee (T=%d cout=%d act=%d/%d)", a->T, a->cout, a->act, a->act2);
  }
  // the kernel sees rows of halfs: a value column is two halfs, a K step of 64 halfs is 32 values
  sd_conv_args k = *a;
  k.lda = 2 * a->lda; k.a_col0 = 2 * a->a_col0;
  k.cin = 2 * a->cin_pad; k.cin_pad = 2 * a->cin_pad;
  k.x_dtype = SD_DT_F16; k.w_dtype = SD_DT_F16;
  const bool plain = vec && (a->act == SD_ACT_RELU || a->act == SD_ACT_NONE) && a->act2 == SD_ACT_NONE && !a->bias_per_seg && !a->tee;
  if (a->y_dtype == SD_DT_SPLIT16) {    // y as split halves: the LDS-staged epilogue's store phase (the register epilogue writes f32 only)
    SD_CHECK_ARG(vec && a->ldo % 32 == 0 && !a->colstat, "sd_conv1d_cl_split16: an SD_DT_SPLIT16 output needs ldo %% 32 == 0, aligned slices and no column statistics");
    return launch_t256<float, false, true>(&k, vec, stream);
  }
  if (plain) return launch_t256<float, true, true>(&k, vec, stream);
  return launch_t256<float, false, true>(&k, vec, stream);
}
