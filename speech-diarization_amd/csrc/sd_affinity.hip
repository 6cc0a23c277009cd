// N x N cosine affinity, upper triangle + mirror, in two arithmetic forms of one kernel: exact f32 (rows of f32 values,
// v_mfma_f32_16x16x4_f32) and split16x3 (SD_DT_SPLIT16 rows, three v_mfma_f32_16x16x32_f16 products per value pair: f32-level accuracy
// at the f16 rate).  The notes below were written for the split form:
// N x N cosine affinity from SD_DT_SPLIT16 rows: the write-bound form of SURVEY 8(d)'s affinity row
// (4 N^2 bytes out, 3 x 2 N^2 D f16 flops on the upper triangle only) [REF anti_stick_diarize.py:176-177, diar_diag.py:215-219].
//
// Why a kernel of its own (round 3 stamps of the 256x256 ring kernel on this shape): with D = 192 a tile is six K
// steps and 512 KB of stores; one workgroup per CU (the ring is the whole LDS) leaves the CU idle on the matrix side while the
// stores drain at the store path's ~13 B/clk and idle on the store side while the next tile's K loop runs: 3.0 TB/s.  Here a
// tile is 128 x 128, its two-stage LDS ring is 64 KB and the kernel keeps its accumulators in 64 registers, so TWO workgroups
// share a CU and one's stores drain under the other's K loop.
//
// * 4 waves as 2 x 2, each 64 x 64 = 4 x 4 tiles of v_mfma_f32_16x16x32_f16; a K step is one packed group of 32 values,
//   [hi x 32 | lo x 32] = one 128-byte row piece per matrix row, three products hi.hi + lo.hi + hi.lo per step.
// * Both operands are rows of the SAME packed matrix; LDS-DMA (global_load_lds_dwordx4) lands 8 rows per wave instruction,
//   the 16-byte slots swizzled on the source side (slot ^ ((row >> 1) & 7)) so that the fragment reads are conflict-free.
// * Tiles on and above the diagonal only, walked in 8 x 8 super-tiles: the 64 workgroups resident on an XCD at a time share
//   16 row panels (1.5 MB of its 4 MB L2) instead of 65.
// * Epilogue from registers.  A lane holds K[i0 + 4 q .. + 3][j0 + r] (q = lane / 16, r = lane % 16): the mirrored block
//   leaves as 16-byte stores as it lies, the direct block after a 4 x 4 transpose inside each lane quad (DPP); both are 16
//   rows x 64 contiguous bytes per wave instruction.  K[i][j] and K[j][i] are the same accumulator: the matrix is exactly
//   symmetric.  Diagonal tiles store every element once (the upper triangle as is, the lower from the transposed accumulator of
//   the upper), element by element.
#include <hip/hip_runtime.h>

#include "sd_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int AF_T = 128;                    // tile edge
constexpr int AF_ROW = 128;                  // bytes of one packed group of one row
constexpr int AF_OPER = AF_T * AF_ROW;       // 16 KB: one operand of one K step
constexpr int AF_STAGE = 2 * AF_OPER;        // rows of the tile's row block, then of its column block
constexpr int AF_LDS = 2 * AF_STAGE;         // 64 KB: two workgroups per CU
constexpr int AF_SUP = 8;                    // super-tile edge in tiles

#define AF_GLDS16(gptr, lptr)                                                              \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),  \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

__device__ __forceinline__ float af_dpp_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float af_dpp_xor2(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
}

// 4 x 4 transpose across a lane quad: in: lane b of the quad holds v[c] = M[c][b]; out: lane b holds M[b][c]
__device__ __forceinline__ f32x4 af_quad_transpose(f32x4 v, int b) {
  const bool o1 = b & 1, o2 = b & 2;
  {
    const float g0 = af_dpp_xor1(o1 ? v[0] : v[1]);
    const float g1 = af_dpp_xor1(o1 ? v[2] : v[3]);
    if (o1) { v[0] = g0; v[2] = g1; } else { v[1] = g0; v[3] = g1; }
  }
  {
    const float g0 = af_dpp_xor2(o2 ? v[0] : v[2]);
    const float g1 = af_dpp_xor2(o2 ? v[1] : v[3]);
    if (o2) { v[0] = g0; v[1] = g1; } else { v[2] = g0; v[3] = g1; }
  }
  return v;
}

// a <-> b on one lane bit: odd 16-lane rows of a trade with the even rows of b / the upper 32 lanes of a with the lower 32 of b
__device__ __forceinline__ void af_swap16(float& a, float& b) {
  const auto w = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  const unsigned w0 = w[0], w1 = w[1];       // (a bit_cast applied to the vector ELEMENT reads element 0 for both: hipcc 7.2)
  a = __builtin_bit_cast(float, w0);
  b = __builtin_bit_cast(float, w1);
}
__device__ __forceinline__ void af_swap32(float& a, float& b) {
  const auto w = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  const unsigned w0 = w[0], w1 = w[1];       // (a bit_cast applied to the vector ELEMENT reads element 0 for both: hipcc 7.2)
  a = __builtin_bit_cast(float, w0);
  b = __builtin_bit_cast(float, w1);
}

__device__ __forceinline__ void af_store4(float* p, f32x4 v) {
#ifdef SD_DIAG_NO_STORE          // timing-only diagnostic build: results are not written
  if (v[0] != 12345.678f) return;
#endif
#ifdef SD_AFFINITY_PLAIN_STORE
  *reinterpret_cast<f32x4*>(p) = v;
#else
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));       // 10 GB of output must not wash the row panels out of L2
#endif
}

#ifdef SD_STAMP
// diagnostic build only (build_native.py --variant stamp "-DSD_STAMP"): cycle counters of every 16th workgroup, read by tools/stamp_affinity.py
__device__ unsigned long long sd_affinity_stamp_buf[8192 * 8];
#endif

// super-tile st (row-major over the upper triangle of the nsup x nsup super-grid) and slot `within` of its 8 x 8 tiles -> tile;
// false below the diagonal or past the edge
__device__ __forceinline__ bool af_tile_of(int st, int within, int nsup, int nt, int& tile_m, int& tile_n) {
  const float b = 2.f * (float)nsup + 1.f;
  int si = (int)((b - sqrtf(b * b - 8.f * (float)st)) * 0.5f);       // super-row i starts at i nsup - i (i - 1) / 2
  si = si < 0 ? 0 : (si > nsup - 1 ? nsup - 1 : si);
  while (si > 0 && si * nsup - si * (si - 1) / 2 > st) --si;
  while (si + 1 < nsup && (si + 1) * nsup - (si + 1) * si / 2 <= st) ++si;
  const int sj = si + (st - (si * nsup - si * (si - 1) / 2));
  tile_m = si * AF_SUP + (within >> 3);
  tile_n = sj * AF_SUP + (within & 7);
  return tile_m <= tile_n && tile_n < nt;
}

// one K step: the three products of one packed group from LDS stage rows a (tile rows) and b (tile columns)
__device__ __forceinline__ void af_step(f32x4 (&acc)[4][4], const char* a, const char* b, int so_hi, int so_lo) {
  h8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ah[i] = *reinterpret_cast<const h8*>(a + i * 16 * AF_ROW + so_hi);
#pragma unroll
  for (int j = 0; j < 4; ++j) bh[j] = *reinterpret_cast<const h8*>(b + j * 16 * AF_ROW + so_hi);
#pragma unroll
  for (int i = 0; i < 4; ++i) al[i] = *reinterpret_cast<const h8*>(a + i * 16 * AF_ROW + so_lo);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) bl[j] = *reinterpret_cast<const h8*>(b + j * 16 * AF_ROW + so_lo);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
}

// the same K step on EXACT f32 operands (v_mfma_f32_16x16x4_f32, the accumulator layout of the f16 form): a stage row is 32 f32 values in
// the same 128 bytes, a lane reads four consecutive k (one ds_read_b128) at chunk fq (+ 4 for the second half) and feeds element r to
// MFMA r, which therefore sums k in {4 fq + r} over the four lane groups; both operands use the same permutation
__device__ __forceinline__ void af_step_f32(f32x4 (&acc)[4][4], const char* a, const char* b, int so_hi, int so_lo) {
#pragma unroll
  for (int c2 = 0; c2 < 2; ++c2) {
    const int so = c2 ? so_lo : so_hi;
    f32x4 av[4], bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const f32x4*>(a + i * 16 * AF_ROW + so);
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(b + j * 16 * AF_ROW + so);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][r], bv[j][r], acc[i][j], 0, 0, 0);
  }
}

// acc[i][j][r] = K[rbase + 16 i + 4 fq + r][cbase + 16 j + fr] / alpha.  Off the diagonal both copies of the block leave as 4 rows x 256
// contiguous bytes per wave instruction (diagnostic builds with the same bytes as 16 x 64 / 8 x 128 / 4 x 256 / 2 x 512 bytes per
// instruction: 3.27 / 2.57 / 2.24 / 2.28 ms for the 50 k matrix), shuffled in registers:
// * direct block, rows along i: a 4 x 4 transpose inside each lane quad turns a lane's 4 rows x 1 column into 1 row x 4 columns
//   (16 bytes), then the four column blocks j trade places with the four 16-lane rows (v_permlane16_swap on lane bit 4,
//   v_permlane32_swap on bit 5): register r then holds rows 4 r .. 4 r + 3 of the wave's 64 columns;
// * mirrored block, rows along j: a lane already holds 16 bytes of its output row; the four row blocks i trade places with the quad
//   lanes (the same quad transpose, per component): register m then holds output rows m, 4 + m, 8 + m, 12 + m, 256 bytes each.
__device__ __forceinline__ void af_epilogue(const f32x4 (&acc)[4][4], bool diag, int rbase, int cbase, int fr, int fq, int N, float* __restrict__ out,
                                            long ldo, float alpha) {
  const int qb = fr & 3, qa = fr >> 2;
  if (!diag) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {                         // direct: rows rbase + 16 i + [0, 16), columns cbase + [0, 64)
      float t[4][4];                                      // [column block j, then row group r][component]
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v = acc[i][j];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= alpha;
        v = af_quad_transpose(v, qb);                     // row 16 i + 4 fq + qb, columns 16 j + 4 qa .. + 3
#pragma unroll
        for (int c = 0; c < 4; ++c) t[j][c] = v[c];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        af_swap16(t[0][c], t[1][c]);
        af_swap16(t[2][c], t[3][c]);
        af_swap32(t[0][c], t[2][c]);
        af_swap32(t[1][c], t[3][c]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {                       // t[r]: row 16 i + 4 r + qb, columns 16 fq + 4 qa .. + 3
        const int ti = rbase + 16 * i + 4 * r + qb, tj = cbase + 16 * fq + 4 * qa;
        if (ti < N && tj < N) af_store4(out + (size_t)ti * ldo + tj, f32x4{t[r][0], t[r][1], t[r][2], t[r][3]});
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {                         // mirrored: rows cbase + 16 j + [0, 16), columns rbase + [0, 64)
      f32x4 w[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 m = af_quad_transpose(f32x4{acc[0][j][c] * alpha, acc[1][j][c] * alpha, acc[2][j][c] * alpha, acc[3][j][c] * alpha}, qb);
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r][c] = m[r];       // w[r]: row 16 j + 4 qa + r, columns 16 qb + 4 fq .. + 3
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mr = cbase + 16 * j + 4 * qa + r, mc = rbase + 16 * qb + 4 * fq;
        if (mr < N && mc < N) af_store4(out + (size_t)mr * ldo + mc, w[r]);
      }
    }
  } else {
    // diagonal tile: element by element, every output written once.  Upper triangle (gj >= gi) as is; the strictly lower part from
    // the transposed accumulator of its mirror image, so that K[i][j] and K[j][i] are the same bits here too
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int gj = cbase + 16 * j + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gi = rbase + 16 * i + 4 * fq + r;
          const float v = acc[i][j][r] * alpha;
          if (gi < N && gj < N && gj >= gi) {
            out[(size_t)gi * ldo + gj] = v;
            if (gj > gi) out[(size_t)gj * ldo + gi] = v;
          }
        }
      }
  }
}

// EXACT = false: rows of SD_DT_SPLIT16 halves, three f16 products per value pair.  EXACT = true: rows of f32 values (zero padded to whole
// groups of 32), exact f32 MFMA.  Either way a row of one group is 128 bytes and `ldx` is the row stride in BYTES.
template <bool EXACT>
__global__ __launch_bounds__(256, 2) void affinity_sym_kernel(const char* __restrict__ xs, const long ldx, const int N,
                                                              const int groups, float* __restrict__ out, const long ldo,
                                                              const float alpha, const int nt, const int nsup) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef SD_STAMP
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime(), r_entry = __builtin_amdgcn_s_memrealtime();
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;

  // workgroup -> tile: XCD blockIdx % 8 walks a contiguous eighth of the super-tiles, 64 consecutive workgroups each
  int wg;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  int tile_m, tile_n;
  if (!af_tile_of(wg >> 6, wg & 63, nsup, nt, tile_m, tile_n)) return;       // workgroup-uniform: below the diagonal or past the edge

  // staging role: thread (r0 = tid / 8, ps = tid % 8) fills physical slot ps of rows r0 + 32 i (i < 4) of both operands; the slot holds
  // logical 16-byte chunk ps ^ ((row >> 1) & 7), and (row >> 1) & 7 does not depend on i
  const int r0 = tid >> 3;
  const int ls8 = ((tid & 7) ^ ((r0 >> 1) & 7)) * 16;     // bytes
  const char* pa[4];
  const char* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#ifdef SD_DIAG_SAME_PANELS       // timing-only diagnostic build: every tile loads rows 0 .. 255 (all L2 hits)
    int m = r0 + 32 * i;
    int n = 128 + r0 + 32 * i;
#else
    int m = tile_m * AF_T + r0 + 32 * i;
    int n = tile_n * AF_T + r0 + 32 * i;
#endif
    m = m < N ? m : N - 1;
    n = n < N ? n : N - 1;
    pa[i] = xs + (size_t)m * ldx + ls8;                   // (ldx, ls8 in bytes)
    pb[i] = xs + (size_t)n * ldx + ls8;
  }
  char* const dst = smem + (wid * 8) * AF_ROW;
  auto issue = [&](int stage) {                          // the next K step
    char* base = dst + stage * AF_STAGE;
#ifdef SD_DIAG_NO_LOAD           // timing-only diagnostic build: operands are whatever the LDS holds
    if (N != 12345) return;
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      AF_GLDS16(pa[i], base + i * 32 * AF_ROW);
      pa[i] += AF_ROW;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      AF_GLDS16(pb[i], base + AF_OPER + i * 32 * AF_ROW);
      pb[i] += AF_ROW;
    }
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const int so_hi = (fq ^ sw) << 4, so_lo = ((4 + fq) ^ sw) << 4;
  const char* const a_base = smem + (wm * 64 + fr) * AF_ROW;
  const char* const b_base = smem + AF_OPER + (wn * 64 + fr) * AF_ROW;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
#ifdef SD_STAMP
  const unsigned long long t_loop0 = __builtin_amdgcn_s_memtime();
#endif
  issue(0);
  for (int kt = 0; kt < groups; ++kt) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // step kt has landed; every wave is done with the stage of step kt - 1
    if (kt + 1 < groups) issue((kt + 1) & 1);
    if constexpr (EXACT) af_step_f32(acc, a_base + (kt & 1) * AF_STAGE, b_base + (kt & 1) * AF_STAGE, so_hi, so_lo);
    else af_step(acc, a_base + (kt & 1) * AF_STAGE, b_base + (kt & 1) * AF_STAGE, so_hi, so_lo);
  }
#ifdef SD_STAMP
  const unsigned long long t_loop1 = __builtin_amdgcn_s_memtime();
#endif
  af_epilogue(acc, tile_m == tile_n, tile_m * AF_T + wm * 64, tile_n * AF_T + wn * 64, fr, fq, N, out, ldo, alpha);
#ifdef SD_STAMP
  const unsigned long long t_issued = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t_exit = __builtin_amdgcn_s_memtime();
  if (tid == 0 && (blockIdx.x & 15) == 0 && (blockIdx.x >> 4) < 8192) {
    unsigned long long* o = sd_affinity_stamp_buf + (blockIdx.x >> 4) * 8;
    o[0] = t_loop0 - t_entry;
    o[1] = t_loop1 - t_loop0;
    o[2] = t_issued - t_loop1;
    o[3] = t_exit - t_issued;
    o[4] = t_exit - t_entry;
    o[5] = __builtin_amdgcn_s_memrealtime() - r_entry;
    o[6] = tile_m != tile_n;
  }
#endif
}

}  // namespace

#ifdef SD_STAMP
extern "C" int sd_debug_read_affinity_stamps(unsigned long long* out, int n) {
  SD_CHECK_HIP(hipDeviceSynchronize());
  SD_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(sd_affinity_stamp_buf), (size_t)n * sizeof(unsigned long long)));
  return SD_OK;
}
#endif

namespace {
template <bool EXACT>
int launch_affinity(const void* xs, long ldx_bytes, int N, int groups, float* out, long ldo, float alpha, sd_stream_t stream, const char* who) {
  if (N % 4 != 0 || ldo % 4 != 0 || !sd_aligned16(out) || !sd_aligned16(xs) || ldx_bytes % 16 != 0 || ldx_bytes < (long)AF_ROW * groups || groups <= 0)
    return sd_set_error(SD_ERR_ARG, "%s: N=%d ldo=%ld row bytes=%ld groups=%d", who, N, ldo, ldx_bytes, groups);
  const int nt = (N + AF_T - 1) / AF_T;
  const int nsup = (nt + AF_SUP - 1) / AF_SUP;
  const long nwg = (long)nsup * (nsup + 1) / 2 * (AF_SUP * AF_SUP);
  if (nwg >= (1L << 31)) return sd_set_error(SD_ERR_ARG, "%s: N=%d needs %ld workgroups", who, N, nwg);
  auto kern = affinity_sym_kernel<EXACT>;
  SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(kern), AF_LDS));
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), AF_LDS, static_cast<hipStream_t>(stream),
                     static_cast<const char*>(xs), ldx_bytes, N, groups, out, ldo, alpha, nt, nsup);
  SD_CHECK_LAUNCH(EXACT ? "affinity_sym_kernel<exact f32>" : "affinity_sym_kernel<split16x3>");
  return SD_OK;
}
}  // namespace

// xs: SD_DT_SPLIT16 [N][ldx halfs] (ldx = 2 x padded D, groups = padded D / 32), both operands; out f32 [N][ldo], N % 4 == 0,
// ldo % 4 == 0, out 16-byte aligned; every entry = alpha x (row i . row j)
int sd_affinity_sym_split16(const void* xs, int ldx, int N, int groups, float* out, long ldo, float alpha, sd_stream_t stream) {
  return launch_affinity<false>(xs, 2L * ldx, N, groups, out, ldo, alpha, stream, "sd_affinity_sym_split16");
}

// xn: f32 [N][ldx floats], columns past D zero up to groups x 32; exact f32 products, K[i][j] and K[j][i] the same bits
int sd_affinity_sym_f32(const float* xn, int ldx, int N, int groups, float* out, long ldo, sd_stream_t stream) {
  return launch_affinity<true>(xn, 4L * ldx, N, groups, out, ldo, 1.0f, stream, "sd_affinity_sym_f32");
}
