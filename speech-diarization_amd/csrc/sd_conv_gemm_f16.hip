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
// f16 activations: LDS-DMA ("glds") pipeline.  At the f16 MFMA rate one K step is ~0.2 us of
// matrix work per wave, so (a) staging through VGPRs + ds_write_b128 makes the LDS write port
// the bottleneck (the 128x128 kernel above measures 400-700 TFLOP/s), and (b) a fetch issued
// one step ahead cannot cover an L2/HBM round trip.  Here global_load_lds_dwordx4 writes the
// operand tiles straight into a 3-stage LDS ring, two K steps ahead, with a counted
// s_waitcnt vmcnt + raw s_barrier per step (never vmcnt(0) in the loop).
//
// Tile 256x128, 8 waves (4x2) of 64x64, BK = 64 halfs; a stage is [256+128 rows][128 bytes],
// unpadded because one LDS-DMA wave-instruction writes 64 lanes x 16 B = 8 whole rows linearly.
// Bank conflicts of the ds_read_b128 fragment reads are removed by an XOR swizzle applied on
// the SOURCE side (each lane fetches the 16-byte chunk that belongs in its physical slot) and
// again on the read: physical slot = logical slot ^ ((row >> 1) & 7); rows 2j, 2j+1 share a
// 256-byte bank row, so 16 consecutive rows at one logical slot land on 16 distinct slots.
constexpr int GBM = 256;
constexpr int GBN = 128;
constexpr int GROW = BK * 2;                       // bytes per LDS row (128)
constexpr int GSTAGE = (GBM + GBN) * GROW;         // 49152 bytes
constexpr int GNST = 3;
constexpr int GRING_BYTES = GNST * GSTAGE;         // 147456
constexpr int GLDS_BYTES = GRING_BYTES;
constexpr int GLDS_PER_STEP = 6;                   // LDS-DMA instructions per thread per K step
constexpr int GTHREADS = 512;                      // 8 waves: 4 (M) x 2 (N) of 64x64
static_assert(GBM * GBN * 4 <= GRING_BYTES, "C tile must fit in the ring");

#ifdef SD_STAMP
// diagnostic build only (build_native.py --stamp): per-workgroup timeline in 10 ns ticks
__device__ unsigned long long sd_stamp_buf[8192 * 8];
#define SD_STAMP_AT(i) do { if (tid == 0) stamp_[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SD_STAMP_AT(i) do { } while (0)
#endif

#define SD_GLDS16(gptr, lptr)                                                              \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),  \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// Measured (in-kernel stamps, tools/stamp_probe.py): a K step takes ~1.2 us against 0.43 us of MFMA
// work; what paces it is the CU's operand ingest (~41 GB/s per CU, L2-resident panels), so this
// 256x128 tile tops out near 0.9 PFLOP/s.  A ninth wave touching the lines six steps ahead (an
// L2 prefetcher with its own vmcnt) was tried and made every step slower; the next lever is a
// 256x256 tile (half the bytes per flop), not more bytes in flight.
template <typename TO>
__global__ __launch_bounds__(GTHREADS, 2) void conv_gemm_f16_glds_kernel(const sd_conv_args p, const int vec) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];

  const int tid = threadIdx.x;
#ifdef SD_STAMP
  unsigned long long stamp_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  SD_STAMP_AT(0);
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  const int n_tiles = (p.cout + GBN - 1) / GBN;
  int wg;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int tile_n = wg % n_tiles;
  const int tile_m = wg / n_tiles;
  const int m0 = tile_m * GBM, n0 = tile_n * GBN;

  // staging role: thread (r0 = tid >> 3, ps = tid & 7) fills physical slot ps of rows r0 + 64 i;
  // a wave-instruction therefore writes rows 8w .. 8w+7 (+64 i) = 1 KB of contiguous LDS.
  const int r0 = tid >> 3;
  const int ps = tid & 7;
  int a_seg[4], a_t[4], a_ls[4];
  const _Float16* aptr[4];
  const _Float16* wptr[2];
  const int ktot = p.taps * p.cin_pad;
  const _Float16* W = static_cast<const _Float16*>(p.w);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = r0 + 64 * i;
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;
    const int seg = (m / p.T) * p.T;
    a_seg[i] = seg;
    a_t[i] = m - seg;
    a_ls[i] = (ps ^ ((row >> 1) & 7)) * 8;     // logical k offset (halfs) this lane fetches for that row
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = r0 + 64 * i;
    int n = n0 + row;
    n = n < p.cout ? n : p.cout - 1;
    wptr[i] = W + (size_t)n * ktot + (ps ^ ((row >> 1) & 7)) * 8;
  }
  const int nk = p.taps * (p.cin_pad / BK);
  const int half = p.taps / 2;
  const _Float16* X = static_cast<const _Float16*>(p.x) + p.a_col0;

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
  int ld_tap = 0, ld_c0 = 0;
  set_tap(0);
  // wave-uniform LDS destinations of this wave's DMA pieces inside a stage
  const int dst_a = (wid * 8) * GROW;                 // + 64*i rows
  const int dst_b = GBM * GROW + (wid * 8) * GROW;
  auto issue = [&](int stage) {
    char* base = smem_raw + stage * GSTAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = ld_c0 + a_ls[i];
      SD_GLDS16(aptr[i] + (col < p.cin ? col : 0), base + dst_a + i * 64 * GROW);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      SD_GLDS16(wptr[i], base + dst_b + i * 64 * GROW);
      wptr[i] += BK;
    }
    ld_c0 += BK;
    if (ld_c0 >= p.cin_pad) {
      ld_c0 = 0;
      ++ld_tap;
      if (ld_tap < p.taps) set_tap(ld_tap);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addressing: lane (r, h) reads logical slot 2*kk + h of its row, swizzled
  const int fr = lane & 31, fh = lane >> 5;
  int a_off[2], b_off[2], a_sw[2], b_sw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wm * 64 + i * 32 + fr;
    const int rb = wn * 64 + i * 32 + fr;
    a_off[i] = ra * GROW;            a_sw[i] = (ra >> 1) & 7;
    b_off[i] = GBM * GROW + rb * GROW; b_sw[i] = (rb >> 1) & 7;
  }
  auto mma_step = [&](const char* st, int kk) {
    const int ls = 2 * kk + fh;
    const h8 a0 = *reinterpret_cast<const h8*>(st + a_off[0] + ((ls ^ a_sw[0]) << 4));
    const h8 a1 = *reinterpret_cast<const h8*>(st + a_off[1] + ((ls ^ a_sw[1]) << 4));
    const h8 b0 = *reinterpret_cast<const h8*>(st + b_off[0] + ((ls ^ b_sw[0]) << 4));
    const h8 b1 = *reinterpret_cast<const h8*>(st + b_off[1] + ((ls ^ b_sw[1]) << 4));
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
  };

  SD_STAMP_AT(1);
  issue(0);
  if (nk > 1) issue(1);
  int st_rd = 0, st_wr = 2;
  for (int kt = 0; kt < nk; ++kt) {
    // retire this step's stage: everything but the youngest K step's DMA pieces must have landed
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // all waves' pieces of stage st_rd landed; stage st_wr is no longer being read
    if (kt == 0) SD_STAMP_AT(2);
    if (kt + 2 < nk) issue(st_wr);
    const char* st = smem_raw + st_rd * GSTAGE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) mma_step(st, kk);
    st_rd = st_rd == GNST - 1 ? 0 : st_rd + 1;
    st_wr = st_wr == GNST - 1 ? 0 : st_wr + 1;
  }
  static_assert(GLDS_PER_STEP == 6, "the counted vmcnt above assumes 6 LDS-DMA pieces per thread per K step");
  SD_STAMP_AT(3);
  __syncthreads();   // every wave is done with the ring before it becomes the C tile

  // ---- epilogue: raw accumulators -> LDS C tile [256][128] f32 -> sd_store_tile (sd_epilogue.h)
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
        Cs[rl * GBN + cl] = acc[mi][ni][r];
      }
    }
  }
  __syncthreads();
  SD_STAMP_AT(4);
  sd_store_tile<TO, GBM, GBN, 512>(p, Cs, GBN, m0, n0, tid, vec);
  SD_STAMP_AT(5);
#ifdef SD_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SD_STAMP_AT(6);
  if (tid == 0 && blockIdx.x < 8192)
    for (int i = 0; i < 8; ++i) sd_stamp_buf[blockIdx.x * 8 + i] = stamp_[i];
#endif
}

template <typename TO>
int launch_glds(const sd_conv_args* a, int vec, hipStream_t stream) {
  const long tiles_m = (a->M + GBM - 1) / GBM;
  const long tiles_n = (a->cout + GBN - 1) / GBN;
  auto kern = conv_gemm_f16_glds_kernel<TO>;
  static bool attr_set = false;
  if (!attr_set) {
    SD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GLDS_BYTES));
    attr_set = true;
  }
  {
    SdProfScope prof(SD_PROF_CONV_GEMM, stream, 2.0 * (double)a->M * (double)a->cout * (double)a->taps * (double)a->cin);
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * tiles_n)), dim3(GTHREADS), GLDS_BYTES, stream, *a, vec);
  }
  SD_CHECK_LAUNCH("conv_gemm_f16_glds_kernel");
  return SD_OK;
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

#ifdef SD_STAMP
extern "C" int sd_debug_read_stamps(unsigned long long* out, int n) {
  SD_CHECK_HIP(hipDeviceSynchronize());
  SD_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(sd_stamp_buf), (size_t)n * sizeof(unsigned long long)));
  return SD_OK;
}
#endif

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
  // Kernel choice, measured per shape on MI355X (tools/probe_conv.py, B*T = 205 824 rows): the 256x128
  // LDS-DMA ring wins only on the big square MFA conv (3072x3072: 800 vs 785 TFLOP/s); the
  // register-staged 128x128 kernel with two workgroups per CU is faster everywhere else
  // (1024x1024: 588 vs 541, Res2Net 128x384: 327 vs 275, 128->3072: 205 vs 185).
  // SD_F16_GLDS=0 / 1 forces one or the other for A/B runs.
  static const int force_glds = [] { const char* e = getenv("SD_F16_GLDS"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
  const bool big = a->cout >= 2048 && (long)a->taps * a->cin >= 2048;
  const bool use_glds = force_glds < 0 ? big : force_glds == 1;
  if (xa && use_glds) return ya ? launch_glds<_Float16>(a, vec, stream) : launch_glds<float>(a, vec, stream);
  if (xa && ya) return launch<_Float16, _Float16>(a, vec, stream);
  if (xa && !ya) return launch<_Float16, float>(a, vec, stream);
  if (!xa && ya) return launch<float, _Float16>(a, vec, stream);
  return launch<float, float>(a, vec, stream);
}
