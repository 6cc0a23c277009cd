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

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case SD_ACT_RELU: return fmaxf(v, 0.0f);
    case SD_ACT_TANH: return tanhf(v);
    case SD_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

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

template <typename TO>
__device__ __forceinline__ void store8(TO* p, const f32x4& a, const f32x4& b);
template <>
__device__ __forceinline__ void store8<_Float16>(_Float16* p, const f32x4& a, const f32x4& b) {
  h8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = (_Float16)a[e]; r[4 + e] = (_Float16)b[e]; }
  *reinterpret_cast<h8*>(p) = r;
}
template <>
__device__ __forceinline__ void store8<float>(float* p, const f32x4& a, const f32x4& b) {
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
template <typename TO>
__device__ __forceinline__ void load8(const TO* p, f32x4& a, f32x4& b);
template <>
__device__ __forceinline__ void load8<_Float16>(const _Float16* p, f32x4& a, f32x4& b) {
  const h8 r = *reinterpret_cast<const h8*>(p);
#pragma unroll
  for (int e = 0; e < 4; ++e) { a[e] = (float)r[e]; b[e] = (float)r[4 + e]; }
}
template <>
__device__ __forceinline__ void load8<float>(const float* p, f32x4& a, f32x4& b) {
  a = *reinterpret_cast<const f32x4*>(p);
  b = *reinterpret_cast<const f32x4*>(p + 4);
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

  h8 ra[4], rb[4];
  int ld_tap = 0, ld_c0 = 0;
  set_tap(0);
  auto gload = [&]() {
    const int col = ld_c0 + c8 * 8;
    const int acol = col < p.cin ? col : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = load_a8<TA>(aptr[i] + acol);
      rb[i] = *reinterpret_cast<const h8*>(wptr[i]);
      wptr[i] += BK;
    }
    ld_c0 += BK;
    if (ld_c0 >= p.cin_pad) {
      ld_c0 = 0;
      ++ld_tap;
      if (ld_tap < p.taps) set_tap(ld_tap);
    }
  };
  auto lstore = [&](int buf) {
    _Float16* a = As + buf * BM * LDP;
    _Float16* b = Bs + buf * BN * LDP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<h8*>(a + (r0 + 32 * i) * LDP + c8 * 8) = ra[i];
      *reinterpret_cast<h8*>(b + (r0 + 32 * i) * LDP + c8 * 8) = rb[i];
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

  gload();
  lstore(0);
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    const _Float16* a = As + cur * BM * LDP + (wm * 64 + frag_row) * LDP + frag_k;
    const _Float16* b = Bs + cur * BN * LDP + (wn * 64 + frag_row) * LDP + frag_k;
    Frag f0 = fread(a, b, 0);
    Frag f1 = fread(a, b, 1);
    if (more) gload();
    mma(f0);
    f0 = fread(a, b, 2);
    mma(f1);
    f1 = fread(a, b, 3);
    mma(f0);
    if (more) lstore(cur ^ 1);
    mma(f1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue phase 1: f32 math on the accumulators, C tile staged in LDS
  float* Cs = reinterpret_cast<float*>(smem_raw);
  const int hrow = (lane >> 5) * 4;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int cl = wn * 64 + ni * 32 + (lane & 31);
    const int n = n0 + cl;
    const bool nok = n < p.cout;
    const float bias_n = (nok && p.bias && !p.bias_per_seg) ? p.bias[n] : 0.f;
    const float sc = (nok && p.scale) ? p.scale[n] : 1.f;
    const float sh = (nok && p.shift) ? p.shift[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + hrow;
        float v = acc[mi][ni][r];
        if (p.bias_per_seg) {
          const int m = m0 + rl;
          v += (nok && m < p.M) ? p.bias[(size_t)(m / p.T) * p.cout + n] : 0.f;
        } else {
          v += bias_n;
        }
        v = apply_act(v, p.act);
        v = v * sc + sh;
        v = apply_act(v, p.act2);
        Cs[rl * LDC + cl] = v;
      }
    }
  }
  __syncthreads();

  // ---- phase 2: 8 output channels per lane, row-contiguous stores
  TO* const Y = static_cast<TO*>(p.y);
  TO* const TEE = static_cast<TO*>(p.tee);
  const TO* const TADD = static_cast<const TO*>(p.tee_add);
  const int cq = (tid & 15) * 8;
  const int n8 = n0 + cq;
  if (vec) {
    if (n8 < p.cout) {
      const bool tee_q = TEE && n8 >= p.tee_lo && n8 < p.tee_hi;
#pragma unroll 4
      for (int rr = tid >> 4; rr < BM; rr += 16) {
        const int m = m0 + rr;
        if (m >= p.M) break;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(Cs + rr * LDC + cq);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(Cs + rr * LDC + cq + 4);
        store8<TO>(Y + (size_t)m * p.ldo + p.o_col0 + n8, v0, v1);
        if (tee_q) {
          f32x4 t0 = v0, t1 = v1;
          if (TADD) {
            f32x4 a0, a1;
            load8<TO>(TADD + (size_t)m * p.ld_ta + p.ta_col0 + (n8 - p.tee_lo), a0, a1);
            t0 += a0; t1 += a1;
          }
          store8<TO>(TEE + (size_t)m * p.ldt + (n8 - p.tee_lo), t0, t1);
        }
      }
    }
  } else {
    for (int rr = tid >> 4; rr < BM; rr += 16) {
      const int m = m0 + rr;
      if (m >= p.M) break;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int n = n8 + e;
        if (n >= p.cout) break;
        const float v = Cs[rr * LDC + cq + e];
        Y[(size_t)m * p.ldo + p.o_col0 + n] = (TO)v;
        if (TEE && n >= p.tee_lo && n < p.tee_hi) {
          float tv = v;
          if (TADD) tv += (float)TADD[(size_t)m * p.ld_ta + p.ta_col0 + (n - p.tee_lo)];
          TEE[(size_t)m * p.ldt + (n - p.tee_lo)] = (TO)tv;
        }
      }
    }
  }
}

template <typename TA, typename TO>
int launch(const sd_conv_args* a, int vec, hipStream_t stream) {
  const long tiles_m = (a->M + BM - 1) / BM;
  const long tiles_n = (a->cout + BN - 1) / BN;
  auto kern = conv_gemm_f16_kernel<TA, TO>;
  static bool attr_set = false;
  if (!attr_set) {
    SD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, STAGE_BYTES));
    attr_set = true;
  }
  {
    SdProfScope prof(SD_PROF_CONV_GEMM, stream, 2.0 * (double)a->M * (double)a->cout * (double)a->taps * (double)a->cin);
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), STAGE_BYTES, stream, *a, vec);
  }
  SD_CHECK_LAUNCH("conv_gemm_f16_kernel");
  return SD_OK;
}

}  // namespace

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
  if (a->tee) {
    vec = vec && a->tee_lo % 8 == 0 && a->tee_hi % 8 == 0 && a->ldt % 8 == 0 && sd_aligned16(a->tee);
    if (a->tee_add) vec = vec && a->ld_ta % 8 == 0 && a->ta_col0 % 8 == 0 && sd_aligned16(a->tee_add);
  }
  const bool xa = a->x_dtype == SD_DT_F16, ya = a->y_dtype == SD_DT_F16;
  if (xa && ya) return launch<_Float16, _Float16>(a, vec, stream);
  if (xa && !ya) return launch<_Float16, float>(a, vec, stream);
  if (!xa && ya) return launch<float, _Float16>(a, vec, stream);
  return launch<float, float>(a, vec, stream);
}
