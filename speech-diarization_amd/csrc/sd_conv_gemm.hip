// Channel-last conv1d / pointwise conv as an implicit GEMM on the gfx950 matrix
// cores, exact f32 (v_mfma_f32_32x32x2_f32), with the TDNNBlock epilogue
// (bias -> ReLU -> eval-BatchNorm affine) of speechbrain's ECAPA-TDNN fused in.
//
// This is the operator behind every Conv1d the reference reaches through
// EncoderClassifier.encode_batch [REF speech_encode.py:77]: the 80->C k=5 stem,
// the C->C / 3C->3C pointwise convs, the dilated k=3 Res2Net convs, the SE and
// attention 1x1 convs and the final FC.
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves, 2x2), each wave
// 64x64 = 2x2 MFMA tiles of 32x32; K is walked in steps of 32 through a
// double-buffered, register-prefetched LDS stage.  Rows are padded by one
// 16-byte slot (row stride 36 floats) so the ds_read_b128 fragment reads of 16
// lanes land on 16 distinct slots of the 256-byte bank row.
//
// Fragment trick: v_mfma_f32_32x32x2_f32 wants lane (i, h) to hold A[i][k=h].
// Each lane reads FOUR consecutive k (one ds_read_b128) at k0 + 4h and feeds
// element r to MFMA r, so MFMA r sums k in {k0 + r, k0 + 4 + r}; A and B use the
// same permutation, and the sum over k is order independent up to rounding.
#include <atomic>
#include <cstdlib>

#include "sd_common.h"
#include "sd_epilogue.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 32;
constexpr int LDP = BK + 4;  // padded LDS row, floats
constexpr int LDC = BN + 4;  // padded row of the epilogue's C tile in LDS
#ifndef SD_F32_DMA_DEFAULT
#define SD_F32_DMA_DEFAULT 1
#endif
static_assert(BM * LDC <= 2 * (BM + BN) * LDP, "C tile must fit in the operand stage");

#ifdef SD_STAMP
__device__ unsigned long long sd_c32_stamp_buf[8192 * 10];
#endif

#define SD_GLDS16_F32(gptr, lptr)                                                          \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),  \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// one 128x128 output tile.  DMA = true: the operand stage is written by LDS-DMA
// (global_load_lds_dwordx4, no VGPR staging and no ds_write phase).  DMA rows are 128 bytes with no
// padding (a wave instruction lands 8 rows = 1 KB contiguously), so the 16-byte chunk a lane FETCHES is
// permuted at the source, chunk c of row r living at position c ^ ((r >> 1) & 7): the fragment reads of 16
// consecutive rows then hit 16 different bank groups.
// NJ: 32-column MFMA tiles per wave: 2 = the 128x128 tile; 1 = a 128x64 tile (DMA only; conv_gemm_f32_n64_kernel below)
template <bool DMA, int NJ = 2>
__device__ __forceinline__ void conv_tile_f32(const sd_conv_args& p, const int vec, const int tile_m, const int tile_n, float* smem,
                                              const bool mirror = false) {
  static_assert(NJ == 2 || (NJ == 1 && DMA), "the half-width tile exists in the LDS-DMA form only");
  constexpr int TBN = 64 * NJ;                // tile columns
  constexpr int TLDC = TBN + 4;               // padded row of the C tile
  constexpr int NBI = TBN / 32;               // staging instructions per wave for the weight rows
#ifdef SD_STAMP
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
  constexpr int LDS_ROW = DMA ? BK : LDP;     // floats per staged row
  float* As = smem;                           // [2][BM][LDS_ROW]
  float* Bs = smem + 2 * BM * LDS_ROW;        // [2][TBN][LDS_ROW]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = tile_m * BM, n0 = tile_n * TBN;

  // staging role: 8 threads per 32-float row, 4 rows per thread.  Rows past M and output
  // channels past cout are clamped to the last valid one (their results are never stored)
  // and columns past cin re-read column 0 against the zero-filled weight padding, so every
  // load is unconditional: no branches and no selects in the K loop.
  const int c4 = tid & 7;
  const int r0 = tid >> 3;

  int a_seg[4], a_t[4];
  const float* wptr[4];
  const float* aptr[4];
  const int ktot = p.taps * p.cin_pad;
  const float* W = static_cast<const float*>(p.w);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + r0 + 32 * i;
    m = m < p.M ? m : p.M - 1;
    const int seg = (m / p.T) * p.T;
    a_seg[i] = seg;
    a_t[i] = m - seg;
    int n = n0 + r0 + 32 * i;
    n = n < p.cout ? n : p.cout - 1;
    wptr[i] = W + (size_t)n * ktot + (DMA ? 0 : c4 * 4);
  }
  const int nk = p.taps * (p.cin_pad / BK);
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

  f32x4 ra[4], rb[4];
  int ld_tap = 0, ld_c0 = 0;  // position of the next K step to fetch
  set_tap(0);
  auto advance = [&]() {
    ld_c0 += BK;
    if (ld_c0 >= p.cin_pad) {
      ld_c0 = 0;
      ++ld_tap;
      if (ld_tap < p.taps) set_tap(ld_tap);
    }
  };
  auto gload = [&]() {
    const int col = ld_c0 + c4 * 4;
    const int acol = col < p.cin ? col : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const f32x4*>(aptr[i] + acol);
      rb[i] = *reinterpret_cast<const f32x4*>(wptr[i]);
      wptr[i] += BK;
    }
    advance();
  };
  auto lstore = [&](int buf) {
    float* a = As + buf * BM * LDP;
    float* b = Bs + buf * BN * LDP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f32x4*>(a + (r0 + 32 * i) * LDP + c4 * 4) = ra[i];
      *reinterpret_cast<f32x4*>(b + (r0 + 32 * i) * LDP + c4 * 4) = rb[i];
    }
  };
  // LDS-DMA of one K step into stage buf: this wave's instruction i lands rows 32 i + 8 wid .. + 7
  // (lane = 8 (row & 7) + position); the lane fetches chunk  position ^ ((row >> 1) & 7)  of its row
  const int gchunk = (c4 ^ ((r0 >> 1) & 7)) * 4;
  auto gdma_part = [&](int buf, int i0, int i1) {
    const int col = ld_c0 + gchunk;
    const int acol = col < p.cin ? col : 0;
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      SD_GLDS16_F32(aptr[i] + acol, As + buf * BM * BK + (32 * i + 8 * wid) * BK);
      if (i < NBI) {
        SD_GLDS16_F32(wptr[i] + gchunk, Bs + buf * TBN * BK + (32 * i + 8 * wid) * BK);
        wptr[i] += BK;
      }
    }
    if (i1 == 4) advance();
  };
  auto gdma = [&](int buf) { gdma_part(buf, 0, 4); };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_row = lane & 31;
  const int frag_k = (lane >> 5) * 4;

  struct Frag { f32x4 a0, a1, b0, b1; };
  // DMA layout: logical chunk 2 k8 + (lane >> 5) of row r sits at position chunk ^ ((r >> 1) & 7)
  int doff[4];
#pragma unroll
  for (int k8 = 0; k8 < 4; ++k8) doff[k8] = (((2 * k8 + (lane >> 5)) ^ ((frag_row >> 1) & 7)) * 4);
  auto fread = [&](const float* a, const float* b, int k8) {
    Frag f;
    const int o = DMA ? doff[k8] : k8 * 8;
    f.a0 = *reinterpret_cast<const f32x4*>(a + o);
    f.a1 = *reinterpret_cast<const f32x4*>(a + 32 * LDS_ROW + o);
    f.b0 = *reinterpret_cast<const f32x4*>(b + o);
    if (NJ == 2) f.b1 = *reinterpret_cast<const f32x4*>(b + 32 * LDS_ROW + o);
    return f;
  };
  auto mma = [&](const Frag& f) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[r], f.b0[r], acc[0][0], 0, 0, 0);
      if (NJ == 2) acc[0][NJ - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[r], f.b1[r], acc[0][NJ - 1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[r], f.b0[r], acc[1][0], 0, 0, 0);
      if (NJ == 2) acc[1][NJ - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[r], f.b1[r], acc[1][NJ - 1], 0, 0, 0);
    }
  };

  if (DMA) {
    gdma(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    gload();
    lstore(0);
  }
  __syncthreads();

  // One barrier per K step.  The fetch of step kt+1 is issued behind the first MFMA group
  // and written to the other LDS stage ahead of the last group, so the part of a K step in
  // which this wave issues no MFMA is as short as possible (its SIMD partner from the other
  // resident workgroup runs the same program and tends to fall into phase with it).
#ifdef SD_STAMP
  unsigned long long tacc[4] = {0, 0, 0, 0};   // wave 0: MFMA groups + fragment reads, fetch issue, stage write (incl. vmcnt wait), barrier
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
  const unsigned long long t_loop = tprev;
#define C32_TSEG(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); tacc[i] += now_ - tprev; tprev = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define C32_TSEG(i) do { } while (0)
#endif
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    const float* a = As + cur * BM * LDS_ROW + (wm * 64 + frag_row) * LDS_ROW + (DMA ? 0 : frag_k);
    const float* b = Bs + cur * TBN * LDS_ROW + (wn * 32 * NJ + frag_row) * LDS_ROW + (DMA ? 0 : frag_k);
    // DMA pieces go out in two halves, behind the first and the second MFMA group (measured on 1024x1024:
    // all at the start of the step 133.8, all behind group 1 133.3, halves 134.7, quarters 129.8 TFLOP/s)
    Frag f0 = fread(a, b, 0);
    Frag f1 = fread(a, b, 1);
    mma(f0);
    C32_TSEG(0);
    if (more) {
      if (DMA) {                   // the other stage is free since the barrier that ended step kt - 1
        gdma_part(cur ^ 1, 0, 2);
      } else {
        gload();
      }
    }
    C32_TSEG(1);
    f0 = fread(a, b, 2);
    mma(f1);
    if (DMA && more) gdma_part(cur ^ 1, 2, 4);
    f1 = fread(a, b, 3);
    mma(f0);
    C32_TSEG(0);
    if (more && !DMA) lstore(cur ^ 1);
    C32_TSEG(2);
    mma(f1);
    C32_TSEG(0);
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces have landed
    __syncthreads();
    C32_TSEG(3);
    cur ^= 1;
  }
#ifdef SD_STAMP
  const unsigned long long t_epi = __builtin_amdgcn_s_memtime();
#endif

  // ---- epilogue: raw accumulators -> LDS C tile (the main loop's last barrier has retired every
  // read of the operand stage), then sd_store_tile applies bias / activation / BN affine and
  // issues 16-byte row-contiguous stores (sd_epilogue.h)
  float* Cs = smem;
  const int hrow = (lane >> 5) * 4;
#pragma unroll
  for (int ni = 0; ni < NJ; ++ni) {
    const int cl = wn * 32 * NJ + ni * 32 + (lane & 31);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + hrow;
        Cs[rl * TLDC + cl] = acc[mi][ni][r];
      }
    }
  }
#ifdef SD_STAMP
  __builtin_amdgcn_s_waitcnt(0xC07F);
  const unsigned long long t_e1 = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();
#ifdef SD_STAMP
  const unsigned long long t_e2 = __builtin_amdgcn_s_memtime();
#endif
  sd_store_tile<float, BM, TBN, 256, 2, NJ == 2 ? 3 : 2>(p, Cs, TLDC, m0, n0, tid, vec);
  if (NJ == 2 && mirror && tile_m != tile_n) {
    // symmetric product (x = w, the affinity): the tile below the diagonal is this tile transposed, written
    // from the same LDS image.  8 lanes cover one 128-byte line of an output row (32 consecutive m), a wave
    // instruction writes 8 rows; the 4 LDS reads behind a store are 2-way conflicted at most.
    float* const Y = static_cast<float*>(p.y);
    const int r4 = wid * 32 + (lane & 7) * 4;
    const int mrow = m0 + r4;
#pragma unroll 4
    for (int c = lane >> 3; c < BN; c += 8) {
      const int n = n0 + c;
      if (n >= p.cout || mrow >= p.M) continue;
      float* dst = Y + (size_t)n * p.ldo + p.o_col0 + mrow;
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = Cs[(r4 + i) * LDC + c];
      if (vec && mrow + 3 < p.M) {
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (mrow + i < p.M) dst[i] = v[i];
      }
    }
  }
#ifdef SD_STAMP
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t_e3 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0);   // stores issued and acknowledged
  if (tid == 0 && blockIdx.x < 8192) {
    unsigned long long* o = sd_c32_stamp_buf + blockIdx.x * 10;
    for (int i = 0; i < 4; ++i) o[i] = tacc[i];
    o[4] = t_loop - t_begin;
    o[5] = __builtin_amdgcn_s_memtime() - t_epi;
    o[6] = t_e1 - t_epi;    // accumulators -> LDS
    o[7] = t_e2 - t_e1;     // barrier
    o[8] = t_e3 - t_e2;     // parameter loads, LDS reads, arithmetic, store issue
    o[9] = t_begin;         // launch time of the workgroup (for the occupancy timeline)
  }
#endif
}

// One workgroup per tile (the grid may be smaller, SD_PERSIST: the workgroups then walk the tile list).
//
// Workgroups are dealt round-robin over the eight XCDs (one 4 MB L2 each).  Every XCD owns a
// contiguous range of logical tiles, laid out in bands of 8 row tiles, n-major inside a band, so
// the ~64 workgroups an XCD runs at a time form an 8 x 8 patch of tiles: each A row panel and each
// weight column panel it touches is shared by 8 workgroups.  Measured: L2 fill traffic of the C->C
// layers 13.6 -> 8.8 GB per launch (A alone is 1.7 GB; workgroups drift apart in K, so whole panels
// would have to stay resident for more), throughput unchanged within the box-to-box noise (the
// kernel is MFMA bound).  SD_TILE_ORDER=0 (host, diagnostic) keeps the launch order.
template <bool DMA>
__global__ __launch_bounds__(256, 2) void conv_gemm_f32_kernel(const sd_conv_args p, const int vec, const int order, const int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n_tiles = (p.cout + BN - 1) / BN;
  const int q = ntiles >> 3, r = ntiles & 7, xcd = blockIdx.x & 7;
  const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;   // this XCD's logical tiles
  const int peers = ((int)gridDim.x - xcd + 7) >> 3;                          // workgroups dealt to this XCD
  const int m_tiles = ntiles / n_tiles;
  const int start = order == 0 ? (int)blockIdx.x : (int)(blockIdx.x >> 3);
  const int count = order == 0 ? ntiles : q + (xcd < r ? 1 : 0);
  const int step = order == 0 ? (int)gridDim.x : peers;
  for (int i = start; i < count; i += step) {      // (one call site: the tile body is inlined once)
    int tm, tn;
    bool mirror = false;
    if (order == 0) {
      tm = i / n_tiles;
      tn = i - tm * n_tiles;
    } else if (order == 2) {
      // symmetric product (M == cout, x == w): bands of 8 tile rows as below, but a band starts at its own diagonal
      // block (columns 8 b .. n_tiles - 1); entries under the diagonal of that first block are skipped, every
      // other tile is also written transposed.  Band b holds rows(b) * (n_tiles - 8 b) list entries;
      // entries before band b: 8 * (b * n_tiles - 4 b (b - 1))  (all bands before b are full)
      const int wg = first + i;
      const float h = (float)(2 * n_tiles + 8);
      int b = (int)((h - sqrtf(fmaxf(h * h - 8.f * (float)wg, 0.f))) * (1.f / 16.f));
      b = b < 0 ? 0 : b;
      while (b > 0 && 8 * (b * n_tiles - 4 * b * (b - 1)) > wg) --b;
      while (8 * (b + 1) < n_tiles && 8 * ((b + 1) * n_tiles - 4 * (b + 1) * b) <= wg) ++b;
      const int in_band = wg - 8 * (b * n_tiles - 4 * b * (b - 1));
      const int rows = n_tiles - b * 8 < 8 ? n_tiles - b * 8 : 8;
      tm = b * 8 + in_band % rows;
      tn = b * 8 + in_band / rows;
      if (tn < tm) continue;                                                   // uniform
      mirror = true;
    } else {
      const int wg = first + i;
      const int band = wg / (8 * n_tiles);
      const int in_band = wg - band * 8 * n_tiles;
      const int rows = m_tiles - band * 8 < 8 ? m_tiles - band * 8 : 8;
      tm = band * 8 + in_band % rows;
      tn = in_band / rows;
    }
    conv_tile_f32<DMA>(p, vec, tm, tn, smem, mirror);
    __syncthreads();              // the next tile refills the LDS stage the epilogue was reading
  }
}

// Tiles of 16 J rows x 128 columns (J = 5, 6, 7: 80, 96, 112 rows) for launches whose 128-row tiles divide badly over the 256 CUs: a launch
// costs ~15 us + (tile times on the busiest CU), so 408 tiles of 128 rows (a C -> C layer at 32 segments: 2 per CU on 152 CUs) cost what
// 512 do, and 464 tiles of 112 rows (still 2 per CU) cost 0.875 of that.  16-row granularity needs v_mfma_f32_16x16x4_f32: 4 waves as 1 x 4,
// each wave all 16 J rows x 32 columns = J x 2 accumulator tiles; a lane reads four consecutive k with one 16-byte access (chunk fq for
// k 0..15, chunk fq + 4 for k 16..31) and feeds element r to MFMA r, the same permutation on both operands.  Staging (LDS-DMA, two
// stages, the 128x128 kernel's swizzle), epilogue (sd_store_tile over an LDS C tile) and column statistics (units of 16 J rows) as there.
template <int J>
__global__ __launch_bounds__(256, 2) void conv_gemm_f32_vh_kernel(const sd_conv_args p, const int vec) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int VM = 16 * J;
  constexpr int NAI = (VM + 31) / 32;         // staging instruction slots for the activation rows
  constexpr int VLDC = BN + 4;
  static_assert(VM * VLDC <= 2 * (VM + BN) * BK, "C tile must fit in the operand stages");
  float* const As = smem;                     // [2][VM][BK]
  float* const Bs = smem + 2 * VM * BK;       // [2][BN][BK]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n_tiles = (p.cout + BN - 1) / BN;
  int wg;
  {                                           // workgroups b, b + 8, ... (one XCD) take consecutive tiles, column tile fastest
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int tile_m = wg / n_tiles, tile_n = wg - tile_m * n_tiles;
  const int m0 = tile_m * VM, n0 = tile_n * BN;

  const int c4 = tid & 7, r0 = tid >> 3;
  const int gchunk = (c4 ^ ((r0 >> 1) & 7)) * 4;
  const int ktot = p.taps * p.cin_pad;
  const int nk = p.taps * (p.cin_pad / BK);
  const int half = p.taps / 2;
  const float* X = static_cast<const float*>(p.x) + p.a_col0;
  int a_seg[NAI], a_t[NAI];
  const float* aptr[NAI];
  const float* wptr[4];
#pragma unroll
  for (int i = 0; i < NAI; ++i) {
    int m = m0 + r0 + 32 * i;
    m = m < p.M ? m : p.M - 1;
    const int seg = (m / p.T) * p.T;
    a_seg[i] = seg;
    a_t[i] = m - seg;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int n = n0 + r0 + 32 * i;
    n = n < p.cout ? n : p.cout - 1;
    wptr[i] = static_cast<const float*>(p.w) + (size_t)n * ktot + gchunk;
  }
  auto set_tap = [&](int tap) {
    const int delta = (tap - half) * p.dil;
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
      int tt = a_t[i] + delta;
      tt = tt < 0 ? -tt : tt;
      tt = tt >= p.T ? 2 * (p.T - 1) - tt : tt;
      aptr[i] = X + (size_t)(a_seg[i] + tt) * p.lda;
    }
  };
  int ld_tap = 0, ld_c0 = 0;
  set_tap(0);
  auto issue = [&](int buf) {
    const int col = ld_c0 + gchunk;
    const int acol = col < p.cin ? col : 0;
#pragma unroll
    for (int i = 0; i < NAI; ++i)
      if (32 * i + 8 * wid < VM)              // wave-uniform: this wave's 8 rows of slot i exist
        SD_GLDS16_F32(aptr[i] + acol, As + buf * VM * BK + (32 * i + 8 * wid) * BK);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      SD_GLDS16_F32(wptr[i], Bs + buf * BN * BK + (32 * i + 8 * wid) * BK);
      wptr[i] += BK;
    }
    ld_c0 += BK;
    if (ld_c0 >= p.cin_pad) {
      ld_c0 = 0;
      ++ld_tap;
      if (ld_tap < p.taps) set_tap(ld_tap);
    }
  };

  f32x4 acc[J][2];
#pragma unroll
  for (int i = 0; i < J; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  // row 16 i + fr (and 32 wid + 16 j + fr) has the swizzle key (fr >> 1) & 7 whatever i, j
  const int so0 = ((fq ^ ((fr >> 1) & 7)) * 4), so1 = (((fq + 4) ^ ((fr >> 1) & 7)) * 4);
  const float* const a_base = As + fr * BK;
  const float* const b_base = Bs + (32 * wid + fr) * BK;

  issue(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const float* a = a_base + cur * VM * BK;
    const float* b = b_base + cur * BN * BK;
    f32x4 av[2][J], bv[2][2];
#pragma unroll
    for (int i = 0; i < J; ++i) av[0][i] = *reinterpret_cast<const f32x4*>(a + 16 * i * BK + so0);
#pragma unroll
    for (int j = 0; j < 2; ++j) bv[0][j] = *reinterpret_cast<const f32x4*>(b + 16 * j * BK + so0);
    if (kt + 1 < nk) issue(cur ^ 1);          // the other stage is free since the barrier that ended step kt - 1
#pragma unroll
    for (int i = 0; i < J; ++i) av[1][i] = *reinterpret_cast<const f32x4*>(a + 16 * i * BK + so1);
#pragma unroll
    for (int j = 0; j < 2; ++j) bv[1][j] = *reinterpret_cast<const f32x4*>(b + 16 * j * BK + so1);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < J; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][i][r], bv[h][j][r], acc[i][j], 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
  // acc[i][j][r] = C[16 i + 4 fq + r][32 wid + 16 j + fr]
  float* Cs = smem;
#pragma unroll
  for (int i = 0; i < J; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cs[(16 * i + 4 * fq + r) * VLDC + 32 * wid + 16 * j + fr] = acc[i][j][r];
  __syncthreads();
  sd_store_tile<float, VM, BN, 256, 2, (J >= 6 ? 3 : 2)>(p, Cs, VLDC, m0, n0, tid, vec);
}

// 128x64 tiles (each wave 64 x 32) for launches in which whole 128x128 tiles leave CUs idle in the last round: a CU works through its
// resident workgroups at the matrix pipe's rate whatever their number, so what counts is how evenly the work divides over the 256
// CUs, and half tiles divide it twice as finely (the MFA conv at 16 segments: 624 tiles = 3 on the busiest CU, 576 us; 1248 half
// tiles = 5 halves, 509 us).  48 KB of LDS: three workgroups per CU.  Column statistics with two parts per tile (T >= 128).
// Workgroups b, b + 8, ... (one XCD) take consecutive tiles, column tile fastest (shared A panel).
__global__ __launch_bounds__(256, 3) void conv_gemm_f32_n64_kernel(const sd_conv_args p, const int vec) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n_tiles = (p.cout + 63) / 64;
  const int nwg = gridDim.x, b = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  const int tm = wg / n_tiles;
  conv_tile_f32<true, 1>(p, vec, tm, wg - tm * n_tiles, smem);
}


// ------------------------------------------------------------------------------------------
// 256x256 tile for the wide outputs (cout >= 1024: C -> C, 3C -> 3C) of large launches: the structure of the f16
// kernel conv_gemm_f16_t256_kernel (sd_conv_gemm_f16.hip) with exact-f32 operands.  One workgroup of 8 waves per CU
// (2 (M) x 4 (N), each 128 x 64 = 4 x 2 tiles of v_mfma_f32_32x32x2_f32, 128 accumulator registers); K step 32 floats =
// 128-byte rows; the 160 KB of LDS hold THREE stages of A (activations, streamed from HBM) and TWO of B (weights), filled
// by LDS-DMA: waves 0-3 fetch the weights of step k + 1, waves 4-7 the activations of step k + 2 (one step in flight
// across the barrier, vmcnt(8)); a K step is four k-groups of 8 (6 fragment reads -> 32 MFMAs of 64 cycles), the reads
// of group g + 1 sit in front of the MFMAs of group g and the barrier that opens step k + 1 in front of the LAST group's
// MFMAs of step k, so a wave reaches every barrier with 2048 cycles of matrix work queued.
// Why: exact-f32 MFMA is not power-limited (tools/micro/mfma_rate.hip: 155-156 TFLOP/s at 2.39 GHz on random data), and
// the 128x128 kernel's K loop runs at 0.89 of that with its barrier + vmcnt(0) per step and two independent workgroups per
// CU (a tile costs 0.1215 us per unit of K against 0.108 at the pipe's rate, plus 5.7 us).
// Measured and not kept: a register epilogue as in the f16 kernel (operands swapped, 16-byte stores of 4 channels per lane,
// column statistics by DPP + one cross-row shuffle): 138.1 against 139.4 TFLOP/s on 1024 x 1024 — its stores cover 32
// bytes per row and instruction where the LDS-staged epilogue writes whole rows; the same ring structure at 128 x 128 (4
// waves, 80 KB, two workgroups per CU) in place of the two-stage kernel below: Res2Net convs 98.8 vs 100.0, attention TDNN
// 124.8 vs 124.1 TFLOP/s, a 128-segment forward 8.21 vs 8.33 ms — the gain of this kernel is its tile (weights fetched once
// per 256 rows, half the DMA pieces per flop), not the ring.
constexpr int WBM = 256, WBN = 256, WBK = 32;
constexpr int W_ROW = 128;                            // bytes per staged row
constexpr int W_A_STAGE = WBM * W_ROW;                // 32 KB
constexpr int W_B_STAGE = WBN * W_ROW;
constexpr int W_B_BASE = 3 * W_A_STAGE;
constexpr int W_LDS_BYTES = W_B_BASE + 2 * W_B_STAGE;
static_assert(W_LDS_BYTES == 160 * 1024, "ring fills the LDS exactly");
static_assert((WBM / 2) * WBN * 4 <= W_LDS_BYTES, "half C tile must fit in the ring");

__global__ __launch_bounds__(512, 2) void conv_gemm_f32_t256_kernel(const sd_conv_args p, const int vec) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;

  const int n_tiles = (p.cout + WBN - 1) / WBN;
  int wg;
  {                                          // XCD-aware: workgroups b, b + 8, ... (one XCD) take consecutive tiles
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int tile_n = wg % n_tiles;
  const int tile_m = wg / n_tiles;
  const int m0 = tile_m * WBM, n0 = tile_n * WBN;

  // staging role: within its group of 256 threads, thread (r0 = lt / 8, ps = lt % 8) fills physical 16-byte slot ps of rows
  // r0 + 32 i (i < 8) of ITS operand: waves 0-3 the weights, waves 4-7 the activations.  The slot holds logical k chunk
  // ps ^ ((row >> 1) & 7), and (row >> 1) & 7 does not depend on i.
  const bool bload = __builtin_amdgcn_readfirstlane(wid) < 4;
  const int lt = tid & 255;
  const int r0 = lt >> 3;
  const int ls4 = ((lt & 7) ^ ((r0 >> 1) & 7)) * 4;
  const int ktot = p.taps * p.cin_pad;
  const int nk = p.taps * (p.cin_pad / WBK);
  const int half = p.taps / 2;
  const float* ptr[8];
  const float* X = static_cast<const float*>(p.x) + p.a_col0;
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
    const float* W = static_cast<const float*>(p.w);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int n = n0 + r0 + 32 * i;
      n = n < p.cout ? n : p.cout - 1;
      ptr[i] = W + (size_t)n * ktot + ls4;
    }
  } else {
    set_tap(0);
  }
  int ld_tap = 0, ld_c0 = 0;
  char* const dst = smem_raw + ((wid & 3) * 8) * W_ROW;
  auto issue_b = [&](int st) {
    char* base = dst + W_B_BASE + st * W_B_STAGE;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      SD_GLDS16_F32(ptr[i], base + i * 32 * W_ROW);
      ptr[i] += WBK;
    }
  };
  auto issue_a = [&](int st) {
    char* base = dst + st * W_A_STAGE;
    const int col = ld_c0 + ls4;
    const int acol = col < p.cin ? col : 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) SD_GLDS16_F32(ptr[i] + acol, base + i * 32 * W_ROW);
    ld_c0 += WBK;
    if (ld_c0 >= p.cin_pad) {
      ld_c0 = 0;
      ++ld_tap;
      if (ld_tap < p.taps) set_tap(ld_tap);
    }
  };

  f32x16 acc[4][2];                          // [32-row tile][32-channel tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int fn = lane & 31, fh = lane >> 5;
  const int ab_sw = (fn >> 1) & 7;
  const char* const a_base = smem_raw + (wm * 128 + fn) * W_ROW;
  const char* const b_base = smem_raw + W_B_BASE + (wn * 64 + fn) * W_ROW;
  // k-group g of a step (8 k): the lane reads 4 consecutive k at 8 g + 4 fh (logical chunk 2 g + fh) with one 16-byte
  // access and feeds element r to MFMA r (same k permutation on both operands)
  const int so0 = ((0 + fh) ^ ab_sw) << 4, so1 = ((2 + fh) ^ ab_sw) << 4, so2 = ((4 + fh) ^ ab_sw) << 4, so3 = ((6 + fh) ^ ab_sw) << 4;

  f32x4 fa[2][4], fb[2][2];
#define W3_READ(buf_, sa_, sb_, so_)                                                                             \
  do {                                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                \
        fa[buf_][i] = *reinterpret_cast<const f32x4*>(a_base + (sa_) * W_A_STAGE + i * 32 * W_ROW + (so_));      \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                \
        fb[buf_][j] = *reinterpret_cast<const f32x4*>(b_base + (sb_) * W_B_STAGE + j * 32 * W_ROW + (so_));      \
  } while (0)
#define W3_MMA(buf_)                                                                                             \
  _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf_][i][r], fb[buf_][j][r], acc[i][j], 0, 0, 0)

  if (bload) {
    issue_b(0);
  } else {
    issue_a(0);
    if (nk > 1) issue_a(1);
  }
  int sa = 0, sb = 0;                        // stages of step kt
  for (int kt = 0; kt < nk; ++kt) {
    // ---- barrier(kt): this step's stages have landed, every wave has finished reading the previous step's
    if (bload || kt + 1 >= nk) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");      // the activations of step kt + 1 stay in flight
    __builtin_amdgcn_s_barrier();
    const int sa2 = sa == 0 ? 2 : sa - 1;                                  // (sa + 2) % 3 = the A stage of step kt - 1
    W3_READ(0, sa, sb, so0);
    if (kt > 0) { W3_MMA(1); }                                             // deferred group 3 of step kt - 1
    if (bload && kt + 1 < nk) issue_b(sb ^ 1);
    W3_READ(1, sa, sb, so1);
    W3_MMA(0);
    if (!bload && kt + 2 < nk) issue_a(sa2);
    W3_READ(0, sa, sb, so2);
    W3_MMA(1);
    W3_READ(1, sa, sb, so3);
    W3_MMA(0);                                                             // (group 3 runs behind the next barrier)
    sa = sa == 2 ? 0 : sa + 1;
    sb ^= 1;
  }
  W3_MMA(1);
#undef W3_READ
#undef W3_MMA

  // ---- epilogue through LDS, one 128-row half at a time (the 256 x 256 f32 C tile does not fit)
  __syncthreads();
  float* Cs = reinterpret_cast<float*>(smem_raw);
#pragma unroll
  for (int hm = 0; hm < 2; ++hm) {
    if (wm == hm) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cl = wn * 64 + j * 32 + fn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int e = 0; e < 16; ++e) Cs[(i * 32 + 8 * (e >> 2) + 4 * fh + (e & 3)) * WBN + cl] = acc[i][j][e];
        }
      }
    }
    __syncthreads();
    sd_store_tile<float, WBM / 2, WBN, 512, 1, 2>(p, Cs, WBN, m0 + hm * (WBM / 2), n0, tid, vec);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// Launches with fewer 128x128 tiles than the chip has CUs: the per-segment layers (SE squeeze/excite,
// global-context bias, final FC: M = B rows) and, at the reference's own batch sizes (16-128
// segments), the narrow Res2Net convs (51 tiles at 32 segments: a 100 k-cycle tile on a fifth of the
// CUs, 21 times per forward).  Here a workgroup owns a 32x32 output tile and its 4 waves split the
// input channels four ways (every tap, a quarter of cin each; exact-f32 32x32x2 MFMA, operands
// straight from global/L2 in 16-byte pieces, same k-permutation as above); the four partial tiles
// are summed through LDS in wave order, so the result does not depend on timing.  Same operator
// contract as the big kernel (taps with reflect padding, bias / activation / affine, tee).
constexpr int SK_T = 32;
constexpr int SK_LD = SK_T + 1;

__global__ __launch_bounds__(256) void skinny_gemm_f32_kernel(const sd_conv_args p) {
  __shared__ float red[4 * SK_T * SK_LD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n_tiles = (p.cout + SK_T - 1) / SK_T;
  const int tile_n = blockIdx.x % n_tiles, tile_m = blockIdx.x / n_tiles;
  const int m0 = tile_m * SK_T, n0 = tile_n * SK_T;
  const int r = lane & 31, h = lane >> 5;
  int m = m0 + r; m = m < p.M ? m : p.M - 1;
  int n = n0 + r; n = n < p.cout ? n : p.cout - 1;
  const int seg = (m / p.T) * p.T, t = m - seg;
  // this wave's share of K: a quarter of cin_pad (a multiple of 8: cin_pad % 32 == 0) of every tap
  const int kq = p.cin_pad / 4;
  const int kb = wid * kq;
  const float* X = static_cast<const float*>(p.x) + p.a_col0 + 4 * h;
  const float* wb = static_cast<const float*>(p.w) + (size_t)n * p.taps * p.cin_pad + 4 * h;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const int half = p.taps / 2;
  for (int tap = 0; tap < p.taps; ++tap) {
    int tt = t + (tap - half) * p.dil;
    tt = tt < 0 ? -tt : tt;
    tt = tt >= p.T ? 2 * (p.T - 1) - tt : tt;
    const float* xa = X + (size_t)(seg + tt) * p.lda;
    const float* wt = wb + (size_t)tap * p.cin_pad;
#pragma unroll 4
    for (int k = kb; k < kb + kq; k += 8) {
      // columns past cin exist only in the zero-padded weights; do not read x there
      const f32x4 a = (k + 4 * h < p.cin) ? *reinterpret_cast<const f32x4*>(xa + k) : z4;
      const f32x4 b = *reinterpret_cast<const f32x4*>(wt + k);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
    }
  }
  float* mine = red + wid * SK_T * SK_LD;
#pragma unroll
  for (int i = 0; i < 16; ++i) mine[((i & 3) + 8 * (i >> 2) + 4 * h) * SK_LD + r] = acc[i];
  __syncthreads();
  // 1024 outputs / 256 threads: thread -> row tid / 8, columns 4 * (tid % 8) .. +3
  const int row = tid >> 3, c0 = (tid & 7) * 4;
  const int mo = m0 + row;
  if (mo >= p.M) return;
  float* Y = static_cast<float*>(p.y);
  float* TEE = static_cast<float*>(p.tee);
  const float* TADD = static_cast<const float*>(p.tee_add);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int no = n0 + c0 + e;
    if (no >= p.cout) break;
    float v = red[row * SK_LD + c0 + e];
#pragma unroll
    for (int w = 1; w < 4; ++w) v += red[w * SK_T * SK_LD + row * SK_LD + c0 + e];
    if (p.bias) v += p.bias_per_seg ? p.bias[(size_t)(mo / p.T) * p.cout + no] : p.bias[no];
    v = sd_apply_act(v, p.act);
    v = v * (p.scale ? p.scale[no] : 1.f) + (p.shift ? p.shift[no] : 0.f);
    v = sd_apply_act(v, p.act2);
    Y[(size_t)mo * p.ldo + p.o_col0 + no] = v;
    if (TEE && no >= p.tee_lo && no < p.tee_hi) {
      if (TADD) v += TADD[(size_t)mo * p.ld_ta + p.ta_col0 + (no - p.tee_lo)];
      TEE[(size_t)mo * p.ldt + (no - p.tee_lo)] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Per-segment layers of small launches (SE squeeze FC, global-context bias, final FC: M = B <= 256 rows, T = 1, K = 1024 / 6144) with
// K split over the GRID.  As 32x32 tiles of the kernel above a 6144 -> 128 layer is 4 workgroups that each pull 1.5 MB of operands
// through one CU (54 us at the ~26 GB/s a CU fetches); here split s of tile t is a workgroup of its own (its 4 waves split the
// chunk again), writes its 32x32 partial sums to scratch [tile][split][32][32], and a second launch adds a tile's partials in split
// order (a fixed order: the result does not depend on timing) and applies the epilogue.  The kernel boundary between the two is the
// hand-off (no tickets, no fences).
__global__ __launch_bounds__(256) void seg_gemm_partial_f32_kernel(const sd_conv_args p, const int nsplit, const int groups, float* __restrict__ part) {
  __shared__ float red[4 * SK_T * SK_LD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n_tiles = (p.cout + SK_T - 1) / SK_T;
  const int split = blockIdx.x % nsplit, tile = blockIdx.x / nsplit;
  const int tile_n = tile % n_tiles, tile_m = tile / n_tiles;
  const int r = lane & 31, h = lane >> 5;
  int m = tile_m * SK_T + r; m = m < p.M ? m : p.M - 1;
  int n = tile_n * SK_T + r; n = n < p.cout ? n : p.cout - 1;
  // this split's chunk: `groups` k-groups of 8 per wave, 4 waves
  const int k0 = (split * 4 + wid) * groups * 8 + 4 * h;
  const float* xa = static_cast<const float*>(p.x) + p.a_col0 + (size_t)m * p.lda;
  const float* wt = static_cast<const float*>(p.w) + (size_t)n * p.cin_pad;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int g = 0; g < groups; ++g) {
    const int k = k0 + 8 * g;
    // past cin only zero-padded weights exist (x is not read there); past cin_pad neither operand does
    const f32x4 a = k < p.cin ? *reinterpret_cast<const f32x4*>(xa + k) : z4;
    const f32x4 b = k < p.cin_pad ? *reinterpret_cast<const f32x4*>(wt + k) : z4;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
  }
  float* mine = red + wid * SK_T * SK_LD;
#pragma unroll
  for (int i = 0; i < 16; ++i) mine[((i & 3) + 8 * (i >> 2) + 4 * h) * SK_LD + r] = acc[i];
  __syncthreads();
  const int row = tid >> 3, c0 = (tid & 7) * 4;
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float t = red[row * SK_LD + c0 + e];
#pragma unroll
    for (int w = 1; w < 4; ++w) t += red[w * SK_T * SK_LD + row * SK_LD + c0 + e];
    v[e] = t;
  }
  *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * SK_T + row) * SK_T + c0) = v;
}

// one workgroup per quarter tile (8 rows x 32 columns, one output per thread): sum over the splits in order, then the epilogue of
// the 32x32 kernel (no tee: the per-segment layers have none)
__global__ __launch_bounds__(256) void seg_gemm_reduce_f32_kernel(const sd_conv_args p, const int nsplit, const float* __restrict__ part) {
  const int tid = threadIdx.x;
  const int n_tiles = (p.cout + SK_T - 1) / SK_T;
  const int tile = blockIdx.x >> 2, q = blockIdx.x & 3;
  const int tile_n = tile % n_tiles, tile_m = tile / n_tiles;
  const int row = q * 8 + (tid >> 5), col = tid & 31;
  const int mo = tile_m * SK_T + row, no = tile_n * SK_T + col;
  const float* src = part + ((size_t)tile * nsplit * SK_T + row) * SK_T + col;
  float v = 0.f;
#pragma unroll 8
  for (int s2 = 0; s2 < nsplit; ++s2) v += src[(size_t)s2 * SK_T * SK_T];
  if (mo >= p.M || no >= p.cout) return;
  if (p.bias) v += p.bias_per_seg ? p.bias[(size_t)(mo / p.T) * p.cout + no] : p.bias[no];
  v = sd_apply_act(v, p.act);
  v = v * (p.scale ? p.scale[no] : 1.f) + (p.shift ? p.shift[no] : 0.f);
  v = sd_apply_act(v, p.act2);
  static_cast<float*>(p.y)[(size_t)mo * p.ldo + p.o_col0 + no] = v;
}

// ------------------------------------------------------------------------------------------
// Time-axis convs of SMALL launches (the reference's own batches: 16-128 segments [REF anti_stick_diarize.py:134,398]): the narrow
// Res2Net convs and the attention TDNN have cout = 128, so a 32-segment launch is 51 tiles of 128x128 on 256 CUs, and as 32x32
// tiles of the kernel above every workgroup re-reads 2 x 32 rows of K from L2 for 32 x 32 outputs (79 MB per Res2Net conv).  Here a
// workgroup owns a 64x64 tile (202 workgroups at 32 segments), 4 waves as 2 x 2 with ONE 32x32 accumulator tile each; a lone
// workgroup per CU has nothing to hide a fetch behind, so the operands come through an LDS ring of S64_ST stages filled by LDS-DMA
// (S64_ST - 1 K steps in flight, counted vmcnt waits, one barrier per step).  Same staging layout (128-byte rows, 16-byte chunk
// c of row r at position c ^ ((r >> 1) & 7)), fragment permutation and epilogue (sd_store_tile) as the 128x128 kernel.
constexpr int S64_T = 64;
constexpr int S64_ST = 4;
constexpr int S64_STAGE = 2 * S64_T * BK;           // floats per stage: 64 A rows + 64 B rows of 32 floats
static_assert(S64_T * (S64_T + 4) <= S64_ST * S64_STAGE, "C tile must fit in the ring");

// TM = 64: the tile above.  TM = 32 (launches of fewer than 128 such tiles, i.e. up to ~16 segments): 32 rows x 64 columns, twice the
// workgroups; waves 0-1 take the first half of every K step's 32 values, waves 2-3 the second, and the two partial tiles are added
// through LDS (first half + second half: a fixed order).
template <int TM>
__global__ __launch_bounds__(256, 2) void conv_gemm_f32_s64_kernel(const sd_conv_args p, const int vec) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int STAGE = (TM + S64_T) * BK;         // floats per stage
  constexpr int NA = TM / 32;                      // staging instructions per wave for the activation rows
  constexpr int PIECES = NA + 2;                   // LDS-DMA pieces per wave and stage
  constexpr int LDCS = S64_T + 4;
#ifdef SD_STAMP
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = TM == 64 ? wid >> 1 : 0, wn = wid & 1, kh = TM == 64 ? 0 : wid >> 1;
  const int n_tiles = (p.cout + S64_T - 1) / S64_T;
  const int tile_m = blockIdx.x / n_tiles, tile_n = blockIdx.x - tile_m * n_tiles;
  const int m0 = tile_m * TM, n0 = tile_n * S64_T;

  // staging role as in conv_tile_f32<true>: thread (r0 = tid / 8, c4 = tid % 8) fetches position c4 of rows r0 (and r0 + 32) of the
  // operands; the wave's instruction i lands rows 32 i + 8 wid .. + 7
  const int c4 = tid & 7, r0 = tid >> 3;
  const int gchunk = (c4 ^ ((r0 >> 1) & 7)) * 4;
  const int ktot = p.taps * p.cin_pad;
  const int nk = p.taps * (p.cin_pad / BK);
  const int half = p.taps / 2;
  const float* X = static_cast<const float*>(p.x) + p.a_col0;
  int a_seg[NA], a_t[NA];
  const float* wptr[2];
  const float* aptr[NA];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (i < NA) {
      int m = m0 + r0 + 32 * i;
      m = m < p.M ? m : p.M - 1;
      const int seg = (m / p.T) * p.T;
      a_seg[i < NA ? i : 0] = seg;
      a_t[i < NA ? i : 0] = m - seg;
    }
    int n = n0 + r0 + 32 * i;
    n = n < p.cout ? n : p.cout - 1;
    wptr[i] = static_cast<const float*>(p.w) + (size_t)n * ktot + gchunk;
  }
  auto set_tap = [&](int tap) {
    const int delta = (tap - half) * p.dil;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int tt = a_t[i] + delta;
      tt = tt < 0 ? -tt : tt;
      tt = tt >= p.T ? 2 * (p.T - 1) - tt : tt;
      aptr[i] = X + (size_t)(a_seg[i] + tt) * p.lda;
    }
  };
  int ld_tap = 0, ld_c0 = 0;
  set_tap(0);
  auto issue = [&](int st) {                       // the next K step -> ring stage st
    float* As = smem + st * STAGE;
    float* Bs = As + TM * BK;
    const int col = ld_c0 + gchunk;
    const int acol = col < p.cin ? col : 0;        // columns past cin meet zero weights
#pragma unroll
    for (int i = 0; i < NA; ++i) SD_GLDS16_F32(aptr[i] + acol, As + (32 * i + 8 * wid) * BK);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      SD_GLDS16_F32(wptr[i], Bs + (32 * i + 8 * wid) * BK);
      wptr[i] += BK;
    }
    ld_c0 += BK;
    if (ld_c0 >= p.cin_pad) {
      ld_c0 = 0;
      ++ld_tap;
      if (ld_tap < p.taps) set_tap(ld_tap);
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int frag_row = lane & 31;
  constexpr int NK8 = TM == 64 ? 4 : 2;            // k groups of 8 per wave and step
  int doff[NK8];
#pragma unroll
  for (int k8 = 0; k8 < NK8; ++k8) doff[k8] = ((2 * (k8 + NK8 * kh) + (lane >> 5)) ^ ((frag_row >> 1) & 7)) * 4;
  const int a_off = (wm * 32 + frag_row) * BK;
  const int b_off = TM * BK + (wn * 32 + frag_row) * BK;

#pragma unroll
  for (int s = 0; s < S64_ST - 1; ++s)
    if (s < nk) issue(s);
  int cur = 0;                                     // stage of step kt
  // Stamped (build_native.py --variant stamp "-DSD_STAMP", tools/stamp_s64.py) at 32 segments, 202 lone workgroups: a K step takes 1570-1700
  // cycles for 1024 of MFMA, and that did not move with the fragment reads of step kt + 1 issued in front of the MFMAs of kt, with two
  // accumulator chains, with eight stages (112 KB in flight) or with the pieces issued one by one between MFMA groups (17.1 us per launch
  // -> 17.1 / 17.2 / 18.3 / 21.4); two workgroups on a CU (64 segments) take 2190 cycles for a step each.  1570 cycles is the step's 16 KB at
  // ~10.4 B per cycle: what one CU gets of data from beyond its XCD's L2 (fbank's load phase: ~11).
#ifdef SD_STAMP
  const unsigned long long t_issued = __builtin_amdgcn_s_memtime();
  unsigned long long t_first = 0;
#endif
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's pieces of step kt have landed: S64_ST - 2 later steps may stay in flight
    if (kt + S64_ST - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (S64_ST - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                               // everyone's have, and step kt - 1's stage has been read by every wave
#ifdef SD_STAMP
    if (kt == 0) t_first = __builtin_amdgcn_s_memtime();
#endif
    if (kt + S64_ST - 1 < nk) issue(cur == 0 ? S64_ST - 1 : cur - 1);
    const float* a = smem + cur * STAGE + a_off;
    const float* b = smem + cur * STAGE + b_off;
    f32x4 fa[NK8], fb[NK8];
#pragma unroll
    for (int k8 = 0; k8 < NK8; ++k8) {
      fa[k8] = *reinterpret_cast<const f32x4*>(a + doff[k8]);
      fb[k8] = *reinterpret_cast<const f32x4*>(b + doff[k8]);
    }
#pragma unroll
    for (int k8 = 0; k8 < NK8; ++k8)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k8][r], fb[k8][r], acc, 0, 0, 0);
    cur = cur == S64_ST - 1 ? 0 : cur + 1;
  }
  __syncthreads();                                 // the ring is free: the C tile goes over it
#ifdef SD_STAMP
  const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif
  float* Cs = smem;
  {
    const int cl = wn * 32 + (lane & 31), hrow = (lane >> 5) * 4;
    float* mine = Cs + kh * TM * LDCS;             // (TM = 32: the second K half's partial tile behind the first's)
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[(wm * 32 + (r & 3) + 8 * (r >> 2) + hrow) * LDCS + cl] = acc[r];
  }
  __syncthreads();
  if (TM == 32) {
    for (int i = tid; i < TM * S64_T; i += 256) {
      const int o = (i >> 6) * LDCS + (i & 63);
      Cs[o] += Cs[TM * LDCS + o];
    }
    __syncthreads();
  }
#ifdef SD_STAMP
  const unsigned long long t_cs = __builtin_amdgcn_s_memtime();
#endif
  sd_store_tile<float, TM, S64_T, 256, 2, 0>(p, Cs, LDCS, m0, n0, tid, vec);
#ifdef SD_STAMP
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t_st = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0);                   // stores acknowledged
  if (tid == 0 && blockIdx.x < 8192) {             // (the 128x128 kernel's buffer: tools/stamp_s64.py)
    unsigned long long* o = sd_c32_stamp_buf + blockIdx.x * 10;
    o[0] = t_issued - t_begin;                     // arguments, addresses, first three stages requested
    o[1] = t_first - t_issued;                     // first stage landed for every wave
    o[2] = t_loop - t_first;                       // K loop
    o[3] = t_cs - t_loop;                          // accumulators -> C tile (+ the halves' add)
    o[4] = t_st - t_cs;                            // parameters, tee_add rows, LDS reads, store issue
    o[5] = __builtin_amdgcn_s_memtime() - t_st;    // stores retired
    o[9] = t_begin;
  }
#endif
}

}  // namespace

#ifdef SD_STAMP
extern "C" int sd_debug_read_c32_stamps(unsigned long long* out, int n) {
  SD_CHECK_HIP(hipDeviceSynchronize());
  SD_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(sd_c32_stamp_buf), (size_t)n * sizeof(unsigned long long)));
  return SD_OK;
}
#endif

namespace {
long tune_env(const char* name, long dflt) {
  const char* e = sd_experiment_env(name);
  return e ? atol(e) : dflt;
}
std::atomic<long> g_skinny_below{tune_env("SD_SKINNY_TILES", 128L)};     // measured at 16 / 32 / 64 / 128 segments: 128 beats 256 and 512
std::atomic<long> g_wide_from{1024L};
// measured (tools/probe_small_shapes.py): Res2Net conv at 64 segments (101 tiles of 128x128) 19 us against 31 on the 128x128 kernel, at 128
// segments (201 tiles) 37 against 34
constexpr long S64_DEFAULT = 128L;
std::atomic<long> g_s64_below{tune_env("SD_S64_TILES", S64_DEFAULT)};
std::atomic<long> g_half_tiles{tune_env("SD_F32_N64", -1L)};     // -1: by the rule; 0 never; 1 whenever possible
std::atomic<long> g_tile_rows{tune_env("SD_F32_TILE_ROWS", -1L)};  // -1: by the rule; 0: 128-row tiles only; 80 / 96 / 112: that height whenever possible
}  // namespace

extern "C" int sd_set_tuning(int key, long value) {
  if (key == SD_TUNE_SKINNY_TILES) {
    g_skinny_below.store(value < 0 ? 128L : value, std::memory_order_relaxed);
    return SD_OK;
  }
  if (key == SD_TUNE_WIDE_TILES) {
    g_wide_from.store(value < 0 ? 1024L : value, std::memory_order_relaxed);
    return SD_OK;
  }
  if (key == SD_TUNE_S64_TILES) {
    g_s64_below.store(value < 0 ? S64_DEFAULT : value, std::memory_order_relaxed);
    return SD_OK;
  }
  if (key == SD_TUNE_TILE_ROWS) {
    g_tile_rows.store(value < 0 ? -1L : value, std::memory_order_relaxed);
    return SD_OK;
  }
  if (key == SD_TUNE_HALF_TILES) {
    g_half_tiles.store(value < 0 ? -1L : (value ? 1L : 0L), std::memory_order_relaxed);
    return SD_OK;
  }
  if (key == SD_TUNE_F16_NARROW_TILES) {
    sd_f16_narrow_tiles().store(value < 0 ? 128L : value, std::memory_order_relaxed);
    return SD_OK;
  }
  if (key == SD_TUNE_T256_LOCKSTEP_TILES) {
    sd_t256_lockstep_tiles().store(value < 0 ? -1L : value, std::memory_order_relaxed);
    return SD_OK;
  }
  return sd_set_error(SD_ERR_ARG, "sd_set_tuning: unknown key %d", key);
}

static int conv1d_cl_f32_impl(const sd_conv_args* a, sd_stream_t stream, bool symmetric, int* stat_rows = nullptr);

extern "C" int sd_conv1d_cl_f32(const sd_conv_args* a, sd_stream_t stream) { return conv1d_cl_f32_impl(a, stream, false); }

// The same for a caller that can take the column statistics in units other than 128 rows (internal, sd_common.h: the ECAPA schedule):
// *stat_rows receives the unit the launch used (128, or the 80 / 96 / 112 rows of the variable-height kernel), for sd_colstat_finish_rows.
int sd_conv1d_cl_f32_rows(const sd_conv_args* a, sd_stream_t stream, int* stat_rows) {
  if (stat_rows) *stat_rows = 128;
  return conv1d_cl_f32_impl(a, stream, false, stat_rows);
}

// x == w, M == cout, no epilogue arithmetic: only the tiles on and above the diagonal are computed, each is stored
// twice (as is and transposed).  Internal (sd_common.h): the affinity's full-matrix call.
int sd_conv1d_cl_f32_symmetric(const sd_conv_args* a, sd_stream_t stream) { return conv1d_cl_f32_impl(a, stream, true); }

static int conv1d_cl_f32_impl(const sd_conv_args* a, sd_stream_t stream, bool symmetric, int* stat_rows) {
  SD_CHECK_ARG(a != nullptr, "sd_conv1d_cl_f32: null args");
  SD_CHECK_ARG(a->w_dtype == SD_DT_F32, "sd_conv1d_cl_f32: w_dtype %d not supported by the f32 operator", a->w_dtype);
  SD_CHECK_ARG(a->x && a->w && a->y, "sd_conv1d_cl_f32: null x/w/y");
  SD_CHECK_ARG(a->x_dtype == SD_DT_F32 && a->y_dtype == SD_DT_F32, "sd_conv1d_cl_f32: x/y must be f32 (use sd_conv1d_cl_f16 for f16 activations)");
  SD_CHECK_ARG(a->M > 0 && a->T > 0 && a->M % a->T == 0, "sd_conv1d_cl_f32: M=%d must be a positive multiple of T=%d", a->M, a->T);
  SD_CHECK_ARG(a->cin > 0 && a->cin % 4 == 0, "sd_conv1d_cl_f32: cin=%d must be a positive multiple of 4", a->cin);
  SD_CHECK_ARG(a->cin_pad >= a->cin && a->cin_pad % BK == 0, "sd_conv1d_cl_f32: cin_pad=%d must be >= cin and a multiple of %d", a->cin_pad, BK);
  SD_CHECK_ARG(a->cout > 0, "sd_conv1d_cl_f32: cout=%d", a->cout);
  SD_CHECK_ARG(a->taps >= 1 && (a->taps & 1), "sd_conv1d_cl_f32: taps=%d must be odd", a->taps);
  SD_CHECK_ARG(a->dil >= 1, "sd_conv1d_cl_f32: dil=%d", a->dil);
  SD_CHECK_ARG((a->taps / 2) * a->dil < a->T, "sd_conv1d_cl_f32: reflect padding %d needs T > pad (T=%d)", (a->taps / 2) * a->dil, a->T);
  SD_CHECK_ARG(a->lda % 4 == 0 && a->a_col0 % 4 == 0 && a->a_col0 + a->cin <= a->lda,
               "sd_conv1d_cl_f32: lda=%d a_col0=%d cin=%d (need multiples of 4, slice inside row)", a->lda, a->a_col0, a->cin);
  SD_CHECK_ARG(a->o_col0 >= 0 && a->o_col0 + a->cout <= a->ldo, "sd_conv1d_cl_f32: output slice outside row (ldo=%d o_col0=%d cout=%d)", a->ldo, a->o_col0, a->cout);
  SD_CHECK_ARG(sd_aligned16(a->x) && sd_aligned16(a->w), "sd_conv1d_cl_f32: x and w must be 16-byte aligned");
  if (a->tee) {
    SD_CHECK_ARG(a->tee_lo >= 0 && a->tee_lo < a->tee_hi && a->tee_hi <= a->cout && a->tee_hi - a->tee_lo <= a->ldt,
                 "sd_conv1d_cl_f32: bad tee range [%d,%d) ldt=%d", a->tee_lo, a->tee_hi, a->ldt);
    if (a->tee_add)
      SD_CHECK_ARG(a->ta_col0 >= 0 && a->ta_col0 + (a->tee_hi - a->tee_lo) <= a->ld_ta, "sd_conv1d_cl_f32: tee_add slice outside row");
  }
  // 16-byte epilogue stores need every touched row slice 16-byte aligned
  int vec = a->cout % 8 == 0 && a->ldo % 4 == 0 && a->o_col0 % 4 == 0 && sd_aligned16(a->y);
  vec = vec && sd_aligned16(a->bias) && sd_aligned16(a->scale) && sd_aligned16(a->shift);   // null is aligned
  if (a->tee) {
    vec = vec && a->tee_lo % 8 == 0 && a->tee_hi % 8 == 0 && a->ldt % 4 == 0 && sd_aligned16(a->tee);
    if (a->tee_add) vec = vec && a->ld_ta % 4 == 0 && a->ta_col0 % 4 == 0 && sd_aligned16(a->tee_add);
  }
  if (a->colstat) {
    const bool simple = (a->act == SD_ACT_RELU || a->act == SD_ACT_NONE) && a->act2 == SD_ACT_NONE && !a->bias_per_seg;
    if (!(vec && simple && a->T >= 64 && a->cout % 256 == 0 && !a->tee))
      return sd_set_error(SD_ERR_UNSUPPORTED, "sd_conv1d_cl_f32: colstat needs T >= 64, cout %% 256 == 0, relu/identity, per-channel bias, "
                          "aligned slices and no tee (T=%d cout=%d act=%d/%d)", a->T, a->cout, a->act, a->act2);
  }
  const long tiles_m = (a->M + BM - 1) / BM;
  const long tiles_n = (a->cout + BN - 1) / BN;
  SD_CHECK_ARG(tiles_m * tiles_n < (1L << 31), "sd_conv1d_cl_f32: grid too large");
  // fewer 128x128 tiles than half the CUs: 32x32 tiles with in-workgroup split-K (sd_set_tuning / SD_SKINNY_TILES)
  if (symmetric)
    SD_CHECK_ARG(a->M == a->cout && a->T == 1 && a->taps == 1 && !a->bias && !a->scale && !a->shift && !a->tee && !a->colstat &&
                 a->act == SD_ACT_NONE && a->act2 == SD_ACT_NONE && a->cout <= a->ldo - a->o_col0,
                 "sd_conv1d_cl_f32_symmetric: needs a square plain product (M=%d cout=%d)", a->M, a->cout);
  // time-axis convs of small launches: 64x64 tiles through a 4-stage LDS-DMA ring (SD_TUNE_S64_TILES)
  if (!a->colstat && !symmetric && a->T > 1 && a->M >= S64_T && tiles_m * tiles_n < g_s64_below.load(std::memory_order_relaxed)) {
    const long nt64 = (a->cout + S64_T - 1) / S64_T;
    const long g = (long)((a->M + S64_T - 1) / S64_T) * nt64;
    const size_t lds64 = (size_t)S64_ST * S64_STAGE * sizeof(float);
    static const int s32 = [] { const char* e = sd_experiment_env("SD_S64_HALF"); return e ? atoi(e) : -1; }();   // 0 | 1: never / always 32-row tiles
    if (s32 != 0 && (s32 == 1 || g < 128)) {
      SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(conv_gemm_f32_s64_kernel<32>), (int)lds64));
      hipLaunchKernelGGL(conv_gemm_f32_s64_kernel<32>, dim3((unsigned)(((a->M + 31) / 32) * nt64)), dim3(256), lds64, static_cast<hipStream_t>(stream), *a, vec);
    } else {
      SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(conv_gemm_f32_s64_kernel<64>), (int)lds64));
      hipLaunchKernelGGL(conv_gemm_f32_s64_kernel<64>, dim3((unsigned)g), dim3(256), lds64, static_cast<hipStream_t>(stream), *a, vec);
    }
    SD_CHECK_LAUNCH("conv_gemm_f32_s64_kernel");
    return SD_OK;
  }
  const long skinny_below = g_skinny_below.load(std::memory_order_relaxed);
  if (!a->colstat && tiles_m * tiles_n < skinny_below) {
    const long g = (long)((a->M + SK_T - 1) / SK_T) * ((a->cout + SK_T - 1) / SK_T);
    // (not counted in the SD_PROF_CONV_GEMM roofline figures: a different kernel, 0.2 % of the flops)
    hipLaunchKernelGGL(skinny_gemm_f32_kernel, dim3((unsigned)g), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
    SD_CHECK_LAUNCH("skinny_gemm_f32_kernel");
    return SD_OK;
  }
  // wide outputs of large launches: the 256x256 ring kernel (SD_F32_WIDE=0: A/B switch; no tee_add epilogue, column
  // statistics only for tiles that span <= 2 segments, at least four rounds of tiles over the CUs)
  static const bool wide_ok = [] { const char* e = sd_experiment_env("SD_F32_WIDE"); return !(e && e[0] == '0'); }();
  {
    const long t256 = ((a->M + WBM - 1) / WBM) * ((a->cout + WBN - 1) / WBN);
    if (wide_ok && !symmetric && a->cout >= 1024 && t256 >= g_wide_from.load(std::memory_order_relaxed) && t256 > 0 && !(a->tee && a->tee_add) && !(a->colstat && a->T < 128)) {
      SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(conv_gemm_f32_t256_kernel), W_LDS_BYTES));
      {
        SdProfScope prof(SD_PROF_CONV_WIDE, static_cast<hipStream_t>(stream),
                         2.0 * (double)a->M * (double)a->cout * (double)a->taps * (double)a->cin);
        hipLaunchKernelGGL(conv_gemm_f32_t256_kernel, dim3((unsigned)t256), dim3(512), W_LDS_BYTES, static_cast<hipStream_t>(stream), *a, vec);
      }
      SD_CHECK_LAUNCH("conv_gemm_f32_t256_kernel");
      return SD_OK;
    }
  }
  // half-width tiles when they shorten the busiest CU's share (SD_TUNE_HALF_TILES / SD_F32_N64=0|1: never / whenever possible).  Measured
  // (tools/probe_tile_alone.py): a launch takes ~15 us + (tiles on the busiest CU) x 62 us per unit of K = 1024, whether the CU's
  // tiles run side by side or one after the other; a half tile costs 0.52 of a tile
  {
    const long n64 = g_half_tiles.load(std::memory_order_relaxed);
    const long t128 = tiles_m * tiles_n, t64 = tiles_m * ((a->cout + 63) / 64);
    const long c128 = (t128 + 255) / 256, c64 = (t64 + 255) / 256;
    const double cost128 = (double)c128, cost64 = 0.52 * (double)c64;
    const bool can = !symmetric && (!a->colstat || (a->T >= 128 && a->cout % 64 == 0)) && t64 < (1L << 31);
    // tiles of 80 / 96 / 112 rows (SD_TUNE_TILE_ROWS): a tile of 16 J rows costs J / 8 of a 128-row tile
    {
      const long rows = g_tile_rows.load(std::memory_order_relaxed);
      int best = 0;
      double best_cost = 0.97 * (n64 != 0 && can && cost64 < cost128 ? cost64 : cost128);
      // (column statistics in units of 16 J rows: only for a caller that asked for the unit)
      // (T > 1: the time-axis convs.  The 16x16x4 MFMA sums four products per step where the 32x32x2 one sums two, so its results differ
      // from the other kernels' in the last bit; plain products such as the affinity's row blocks (T = 1) keep the bits of the 128x128 kernel)
      if (!symmetric && a->T > 1 && (!a->colstat || stat_rows) && rows != 0)
        for (int j = 5; j <= 7; ++j) {
          if (a->colstat && a->T < (j == 5 ? 80 : 8 * j)) continue;      // a tile may span two (J = 5) / three segments
          const long tj = ((a->M + 16 * j - 1) / (16 * j)) * tiles_n;
          const double cj = (double)((tj + 255) / 256) * (double)j / 8.0 * 1.02;
          if (rows == 16 * j || (rows < 0 && cj < best_cost)) { best = j; best_cost = cj; }
        }
      if (best) {
        const long tj = ((a->M + 16 * best - 1) / (16 * best)) * tiles_n;
        const size_t ldsv = (size_t)2 * (16 * best + BN) * BK * sizeof(float);
        const void* fn = best == 5 ? reinterpret_cast<const void*>(conv_gemm_f32_vh_kernel<5>)
                       : best == 6 ? reinterpret_cast<const void*>(conv_gemm_f32_vh_kernel<6>) : reinterpret_cast<const void*>(conv_gemm_f32_vh_kernel<7>);
        SD_CHECK_HIP(sd_func_max_lds(fn, (int)ldsv));
        {
          SdProfScope prof(SD_PROF_CONV_GEMM, static_cast<hipStream_t>(stream), 2.0 * (double)a->M * (double)a->cout * (double)a->taps * (double)a->cin);
          const dim3 grid((unsigned)tj), block(256);
          hipStream_t hs = static_cast<hipStream_t>(stream);
          if (best == 5) hipLaunchKernelGGL(conv_gemm_f32_vh_kernel<5>, grid, block, ldsv, hs, *a, vec);
          else if (best == 6) hipLaunchKernelGGL(conv_gemm_f32_vh_kernel<6>, grid, block, ldsv, hs, *a, vec);
          else hipLaunchKernelGGL(conv_gemm_f32_vh_kernel<7>, grid, block, ldsv, hs, *a, vec);
        }
        SD_CHECK_LAUNCH("conv_gemm_f32_vh_kernel");
        if (stat_rows) *stat_rows = 16 * best;
        return SD_OK;
      }
    }
    if (can && n64 != 0 && (n64 == 1 || cost64 < 0.97 * cost128)) {
      const size_t lds64 = (size_t)2 * (BM + 64) * BK * sizeof(float);
      SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(conv_gemm_f32_n64_kernel), (int)lds64));
      {
        SdProfScope prof(SD_PROF_CONV_GEMM, static_cast<hipStream_t>(stream), 2.0 * (double)a->M * (double)a->cout * (double)a->taps * (double)a->cin);
        hipLaunchKernelGGL(conv_gemm_f32_n64_kernel, dim3((unsigned)t64), dim3(256), lds64, static_cast<hipStream_t>(stream), *a, vec);
      }
      SD_CHECK_LAUNCH("conv_gemm_f32_n64_kernel");
      return SD_OK;
    }
  }
  // SD_F32_DMA=0|1 (diagnostic): operand staging through registers or by LDS-DMA
  static const int dma = [] {
    const char* e = sd_experiment_env("SD_F32_DMA");
    return e ? atoi(e) : SD_F32_DMA_DEFAULT;
  }();
  const size_t lds = (size_t)2 * (BM + BN) * LDP * sizeof(float);   // the C tile of the epilogue needs BM * LDC <= this
  SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(conv_gemm_f32_kernel<false>), (int)lds));
  SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(conv_gemm_f32_kernel<true>), (int)lds));
  static const int order = [] {
    const char* e = sd_experiment_env("SD_TILE_ORDER");
    return e ? atoi(e) : 1;
  }();
  // SD_PERSIST=<n> (diagnostic): at most n workgroups walk the tile list instead of one workgroup per
  // tile.  Measured with n = 2 per CU: 2.3 % SLOWER (the dispatcher's dynamic placement beats a static
  // share of the tiles; dispatch gaps are not what separates the kernel from its K-loop rate).
  static const int persist = [] {
    const char* e = sd_experiment_env("SD_PERSIST");
    return e ? atoi(e) : 0;
  }();
  {
    SdProfScope prof(SD_PROF_CONV_GEMM, static_cast<hipStream_t>(stream),
                     2.0 * (double)a->M * (double)a->cout * (double)a->taps * (double)a->cin);
    long ntiles = tiles_m * tiles_n;
    int ord = order;
    if (symmetric && tiles_n >= 2) {                      // the list of band entries (see the kernel)
      ntiles = 0;
      for (long b = 0; 8 * b < tiles_n; ++b) ntiles += (tiles_n - 8 * b < 8 ? tiles_n - 8 * b : 8) * (tiles_n - 8 * b);
      ord = 2;
    }
    const long grid = (persist > 0 && ntiles > persist) ? persist : ntiles;
    if (dma)
      hipLaunchKernelGGL(conv_gemm_f32_kernel<true>, dim3((unsigned)grid), dim3(256), lds,
                         static_cast<hipStream_t>(stream), *a, vec, ord, (int)ntiles);
    else
      hipLaunchKernelGGL(conv_gemm_f32_kernel<false>, dim3((unsigned)grid), dim3(256), lds,
                         static_cast<hipStream_t>(stream), *a, vec, ord, (int)ntiles);
  }
  SD_CHECK_LAUNCH("conv_gemm_f32_kernel");
  return SD_OK;
}

// a split = 4 waves x `groups` k-groups of 8: 256 values of K for the long layers, 128 below 2048
static int seg_gemm_shape(int M, int cin_pad, int cout, int* groups, int* nsplit, long* tiles) {
  if (M <= 0 || M > 256 || cin_pad < 512 || cin_pad % 32 != 0 || cout <= 0) return 0;
  *groups = cin_pad >= 2048 ? 8 : 4;
  *nsplit = (int)(((long)cin_pad + 32 * *groups - 1) / (32 * *groups));
  *tiles = (long)((M + SK_T - 1) / SK_T) * (((long)cout + SK_T - 1) / SK_T);
  return *nsplit >= 2;
}

extern "C" size_t sd_seg_gemm_scratch_bytes(int M, int cin_pad, int cout) {
  int groups, nsplit; long tiles;
  if (!seg_gemm_shape(M, cin_pad, cout, &groups, &nsplit, &tiles)) return 0;
  return (size_t)tiles * nsplit * SK_T * SK_T * sizeof(float);
}

// Per-segment layer (T == 1) with caller-provided scratch: K split over the grid when the launch is small and K long (the two kernels
// above), otherwise sd_conv1d_cl_f32: the ECAPA schedule's SE squeeze FC, global-context bias and final FC.
extern "C" int sd_seg_gemm_f32(const sd_conv_args* a, void* scratch, size_t scratch_bytes, sd_stream_t stream) {
  static const bool on = [] { const char* e = sd_experiment_env("SD_SEG_SPLITK"); return !(e && e[0] == '0'); }();
  if (!on || !a || !scratch || a->T != 1 || a->taps != 1 || a->tee || a->colstat || a->w_dtype != SD_DT_F32 || a->x_dtype != SD_DT_F32 ||
      a->y_dtype != SD_DT_F32 || a->M <= 0 || a->M > 256 || a->cin_pad < 512 || a->cin_pad % 32 != 0 || a->cin % 4 != 0 || a->lda % 4 != 0 ||
      a->a_col0 % 4 != 0 || !sd_aligned16(a->x) || !sd_aligned16(a->w) || !sd_aligned16(scratch))
    return sd_conv1d_cl_f32(a, stream);
  int groups, nsplit; long tiles;
  if (!seg_gemm_shape(a->M, a->cin_pad, a->cout, &groups, &nsplit, &tiles) || (size_t)tiles * nsplit * SK_T * SK_T * sizeof(float) > scratch_bytes ||
      !a->x || !a->w || !a->y || a->cin <= 0 || a->cin > a->cin_pad || a->o_col0 < 0 || a->o_col0 + a->cout > a->ldo || a->a_col0 < 0 || a->a_col0 + a->cin > a->lda)
    return sd_conv1d_cl_f32(a, stream);
  float* part = static_cast<float*>(scratch);
  SdProfScope prof(SD_PROF_SEG_SPLITK, static_cast<hipStream_t>(stream), 2.0 * (double)a->M * (double)a->cout * (double)a->cin);
  hipLaunchKernelGGL(seg_gemm_partial_f32_kernel, dim3((unsigned)(tiles * nsplit)), dim3(256), 0, static_cast<hipStream_t>(stream), *a, nsplit, groups, part);
  SD_CHECK_LAUNCH("seg_gemm_partial_f32_kernel");
  hipLaunchKernelGGL(seg_gemm_reduce_f32_kernel, dim3((unsigned)(tiles * 4)), dim3(256), 0, static_cast<hipStream_t>(stream), *a, nsplit, part);
  SD_CHECK_LAUNCH("seg_gemm_reduce_f32_kernel");
  return SD_OK;
}
