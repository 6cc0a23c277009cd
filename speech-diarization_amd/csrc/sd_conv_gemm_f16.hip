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

#ifdef SD_STAMP
// diagnostic build only (build_native.py --stamp): per-workgroup cycle counters, read by tools/stamp_t256.py
__device__ unsigned long long sd_stamp_buf[8192 * 8];
#endif

#define SD_GLDS16(gptr, lptr)                                                              \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),  \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// ------------------------------------------------------------------------------------------
// 256x256 tile for the large square convs, fed by LDS-DMA (global_load_lds_dwordx4 writes the operand
// tiles straight into an LDS ring, no VGPR staging).  What paces every f16 variant is the CU's
// vector-memory path: a 1 KB DMA piece takes ~57 cycles of it, i.e. ~18 B/clk per CU of operand ingest
// (in-kernel cycle counters, tools/stamp_t256.py: a K step is 1789 cycles for 32 KB and 1024 cycles of
// MFMA), so the lever is bytes per flop, and this tile has half those of the 128x128 kernel above.  (A 256x128 LDS-DMA variant and a ninth "L2 prefetch" wave were tried and
// dropped: 800 vs 864 TFLOP/s on 3072x3072, and slower, respectively.)  8 waves as 2 (M) x 4 (N), each
// 128x64 = 4x2 MFMA tiles (128 accumulator registers); K step 32 halfs, i.e. 64-byte LDS rows,
// [512 rows] = 32 KB per stage, a 4-stage LDS-DMA ring (three K steps in flight), counted vmcnt
// + raw barrier per step.  Swizzle for 64-byte rows: four rows share a 256-byte bank row, so
// physical slot = logical slot ^ ((row >> 2) & 3) puts 16 consecutive rows on 16 distinct slots.
// The 256x256 f32 C tile does not fit LDS: the epilogue runs once per 128-row half.  (A persistent walk over
// the tiles, meant to let a tile's stores drain under the next K loop, measured no better: the loop state costs
// 13-26 spilled registers in this 256-register kernel and the next tile's first vmcnt(0) waits for the stores anyway.)
constexpr int TBM = 256;
constexpr int TBN = 256;
#ifndef SD_T256_LATE
#define SD_T256_LATE 1      // the second half of the waves issues its DMA pieces after this many groups of 8 MFMAs
#endif
#ifndef SD_T256_K_DEFAULT
#define SD_T256_K_DEFAULT 64
#endif
constexpr int TLDS_BYTES = 131072;                 // the operand ring: 4 stages of K = 32 or 2 stages of K = 64
static_assert((TBM / 2) * TBN * 4 <= TLDS_BYTES, "half C tile must fit in the ring");
// TBK = 64 (128-byte rows): a DMA piece lands 8 rows and reads 8 FULL 128-byte cache lines.  With TBK = 32 a
// piece reads half of 16 lines, and the other halves are fetched again one K step later: the L1 / vector-memory
// path then moves twice the lines per useful byte (tools/micro/dma_rate.hip: whole lines stream at 76 B/clk/CU
// from L2; the TBK = 32 loop was held at 18 B/clk).

template <typename TO, int TBK>
__global__ __launch_bounds__(512, 2) void conv_gemm_f16_t256_kernel(const sd_conv_args p, const int vec) {
  constexpr int TROW = TBK * 2;                      // bytes per staged row: 64 or 128
  constexpr int SLOTS = TROW / 16;                   // 16-byte slots per row: 4 or 8
  constexpr int SWSH = TROW == 64 ? 2 : 1;           // rows per 256-byte bank row = 1 << SWSH
  constexpr int TSTAGE = (TBM + TBN) * TROW;         // 32 KB or 64 KB
  constexpr int TNST = TLDS_BYTES / TSTAGE;          // 4 or 2
  constexpr int RPI = 1024 / TROW;                   // rows one wave instruction lands: 16 or 8
  constexpr int RSTEP = 512 / SLOTS;                 // rows covered by the 512 threads per piece: 128 or 64
  constexpr int NR = TBM / RSTEP;                    // pieces per operand per K step and thread: 2 or 4
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
#ifdef SD_STAMP
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;

  const int n_tiles = (p.cout + TBN - 1) / TBN;
  int wg;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int tile_n = wg % n_tiles;
  const int tile_m = wg / n_tiles;
  const int m0 = tile_m * TBM, n0 = tile_n * TBN;

  // staging role: thread (r0 = tid / SLOTS, ps = tid % SLOTS) fills physical slot ps of A rows r0 + RSTEP i and
  // of the same B rows; one wave-instruction writes RPI whole rows (1 KB) of LDS
  const int r0 = tid / SLOTS;
  const int ps = tid % SLOTS;
  int a_seg[NR], a_t[NR], a_ls[NR];
  const _Float16* aptr[NR];
  const _Float16* wptr[NR];
  const int ktot = p.taps * p.cin_pad;
  const _Float16* W = static_cast<const _Float16*>(p.w);
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int row = r0 + RSTEP * i;
    const int ls = (ps ^ ((row >> SWSH) & (SLOTS - 1))) * 8;    // logical k offset (halfs) that belongs in this lane's slot
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;
    const int seg = (m / p.T) * p.T;
    a_seg[i] = seg;
    a_t[i] = m - seg;
    a_ls[i] = ls;
    int n = n0 + row;
    n = n < p.cout ? n : p.cout - 1;
    wptr[i] = W + (size_t)n * ktot + ls;
  }
  const int nk = p.taps * (p.cin_pad / TBK);
  const int half = p.taps / 2;
  const _Float16* X = static_cast<const _Float16*>(p.x) + p.a_col0;
  auto set_tap = [&](int tap) {
    const int delta = (tap - half) * p.dil;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      int tt = a_t[i] + delta;
      tt = tt < 0 ? -tt : tt;
      tt = tt >= p.T ? 2 * (p.T - 1) - tt : tt;
      aptr[i] = X + (size_t)(a_seg[i] + tt) * p.lda;
    }
  };
  int ld_tap = 0, ld_c0 = 0;
  set_tap(0);
  const int dst_a = (wid * RPI) * TROW;                 // + RSTEP * i rows
  const int dst_b = TBM * TROW + (wid * RPI) * TROW;
  // DMA piece g of a K step: g < NR -> A rows r0 + RSTEP g; else B rows r0 + RSTEP (g - NR)
  auto piece = [&](char* base, int g) {
    if (g < NR) {
      const int col = ld_c0 + a_ls[g];
      SD_GLDS16(aptr[g] + (col < p.cin ? col : 0), base + dst_a + g * RSTEP * TROW);
    } else {
      SD_GLDS16(wptr[g - NR], base + dst_b + (g - NR) * RSTEP * TROW);
    }
  };
  auto advance = [&]() {
#pragma unroll
    for (int i = 0; i < NR; ++i) wptr[i] += TBK;
    ld_c0 += TBK;
    if (ld_c0 >= p.cin_pad) {
      ld_c0 = 0;
      ++ld_tap;
      if (ld_tap < p.taps) set_tap(ld_tap);
    }
  };
  auto issue = [&](int stage) {
    char* base = smem_raw + stage * TSTAGE;
#pragma unroll
    for (int g = 0; g < 2 * NR; ++g) piece(base, g);
    advance();
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  int a_off[4], a_sw[4], b_off[2], b_sw[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = wm * 128 + i * 32 + fr;
    a_off[i] = ra * TROW;
    a_sw[i] = (ra >> SWSH) & (SLOTS - 1);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rb = wn * 64 + i * 32 + fr;
    b_off[i] = TBM * TROW + rb * TROW;
    b_sw[i] = (rb >> SWSH) & (SLOTS - 1);
  }

#ifdef SD_STAMP
  unsigned long long tacc[4] = {0, 0, 0, 0};   // wave 0's cycles in: DMA wait, barrier, DMA issue, LDS reads + MFMA
  const unsigned long long t_loop0 = __builtin_amdgcn_s_memtime();
#define SD_TSEG(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tprev; tprev = now_; } while (0)
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#else
#define SD_TSEG(i) do { } while (0)
#endif
  for (int s0 = 0; s0 < TNST - 1 && s0 < nk; ++s0) issue(s0);
  int st_rd = 0, st_wr = TNST - 1;
  // SIMD partners (waves w and w + 4) are staggered: the first half issues its DMA pieces right after the
  // barrier, the second half after its first 8 MFMAs, so one partner's DMA issue runs under the other's MFMAs.
  // A piece costs its wave 85-90 cycles with four waves issuing together (in-kernel counters); the vector-memory
  // path itself takes 13.5 cycles per piece at saturation (tools/micro/dma_rate.hip).  Finer schedules measured
  // slower: four issue slots (after 0 / 4 / 8 / 12 MFMAs, two waves each) 973 vs 1040 TFLOP/s on 3072x3072,
  // single pieces between the MFMA groups of one wave likewise.
  const bool early = __builtin_amdgcn_readfirstlane(wid) < 4;
  for (int kt = 0; kt < nk; ++kt) {
    SD_TSEG(3);
    // 2 NR DMA pieces per thread per step; the TNST - 2 youngest steps may stay in flight
    if (TNST == 4) {
      if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    SD_TSEG(0);
    __builtin_amdgcn_s_barrier();
    SD_TSEG(1);
    const bool more = kt + TNST - 1 < nk;
    if (more && early) issue(st_wr);
    SD_TSEG(2);
    const char* st = smem_raw + st_rd * TSTAGE;
#pragma unroll
    for (int pp = 0; pp < TBK / 32; ++pp) {          // pairs of 16-wide k slices
      h8 fa[2][4], fb[2][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int ls = 2 * (2 * pp + kk) + fh;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[kk][i] = *reinterpret_cast<const h8*>(st + a_off[i] + ((ls ^ a_sw[i]) << 4));
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[kk][j] = *reinterpret_cast<const h8*>(st + b_off[j] + ((ls ^ b_sw[j]) << 4));
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
        }
        if (2 * pp + kk == SD_T256_LATE - 1 && more && !early) issue(st_wr);
      }
    }
    st_rd = st_rd == TNST - 1 ? 0 : st_rd + 1;
    st_wr = st_wr == TNST - 1 ? 0 : st_wr + 1;
  }
#ifdef SD_STAMP
  SD_TSEG(3);
  const unsigned long long t_loop1 = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();

  // ---- epilogue, one 128-row half at a time: the owning waves stage raw accumulators as a
  // [128][256] f32 tile in the ring, then all 512 threads run sd_store_tile on it
  float* Cs = reinterpret_cast<float*>(smem_raw);
  const int hrow = (lane >> 5) * 4;
#pragma unroll
  for (int hm = 0; hm < 2; ++hm) {
    if (wm == hm) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int cl = wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = mi * 32 + (r & 3) + 8 * (r >> 2) + hrow;
            Cs[rl * TBN + cl] = acc[mi][ni][r];
          }
        }
      }
    }
    __syncthreads();
    sd_store_tile<TO, TBM / 2, TBN, 512, 1, 2>(p, Cs, TBN, m0 + hm * (TBM / 2), n0, tid, vec);
    __syncthreads();
  }
#ifdef SD_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t_exit = __builtin_amdgcn_s_memtime();
  if (tid == 0 && blockIdx.x < 8192) {
    for (int i = 0; i < 4; ++i) sd_stamp_buf[blockIdx.x * 8 + i] = tacc[i];
    sd_stamp_buf[blockIdx.x * 8 + 4] = t_loop0 - t_entry;    // prologue
    sd_stamp_buf[blockIdx.x * 8 + 5] = t_exit - t_loop1;     // epilogue incl. store drain
    sd_stamp_buf[blockIdx.x * 8 + 6] = t_exit - t_entry;
    sd_stamp_buf[blockIdx.x * 8 + 7] = t_entry;
  }
#endif
}

template <typename TO, int TBK>
int launch_t256(const sd_conv_args* a, int vec, hipStream_t stream) {
  const long tiles_m = (a->M + TBM - 1) / TBM;
  const long tiles_n = (a->cout + TBN - 1) / TBN;
  auto kern = conv_gemm_f16_t256_kernel<TO, TBK>;
  static bool attr_set = false;
  if (!attr_set) {
    SD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, TLDS_BYTES));
    attr_set = true;
  }
  {
    SdProfScope prof(SD_PROF_CONV_GEMM, stream, 2.0 * (double)a->M * (double)a->cout * (double)a->taps * (double)a->cin);
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * tiles_n)), dim3(512), TLDS_BYTES, stream, *a, vec);
  }
  SD_CHECK_LAUNCH("conv_gemm_f16_t256_kernel");
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
    const char* e = getenv("SD_F16_KERNEL");
    if (!e) return -1;
    return e[0] == 'r' ? 0 : e[0] == 't' ? 2 : -1;
  }();
  const bool wide = a->cout >= 1024;
  const int choice = forced >= 0 ? forced : (wide ? 2 : 0);
  static const int t256_k = [] {      // SD_T256_K=32|64: K step of the 256x256 kernel (diagnostic)
    const char* e = getenv("SD_T256_K");
    return e ? atoi(e) : SD_T256_K_DEFAULT;
  }();
  // (the 256x256 kernel: no tee_add epilogue, and column statistics only for tiles that span <= 2 segments)
  if (xa && choice == 2 && !(a->tee && a->tee_add) && !(a->colstat && a->T < 128)) {
    if (t256_k == 64) return ya ? launch_t256<_Float16, 64>(a, vec, stream) : launch_t256<float, 64>(a, vec, stream);
    return ya ? launch_t256<_Float16, 32>(a, vec, stream) : launch_t256<float, 32>(a, vec, stream);
  }
  if (xa && ya) return launch<_Float16, _Float16>(a, vec, stream);
  if (xa && !ya) return launch<_Float16, float>(a, vec, stream);
  if (!xa && ya) return launch<float, _Float16>(a, vec, stream);
  return launch<float, float>(a, vec, stream);
}
