// The Res2Net chain of one SE-Res2Net block as ONE kernel (f16 activations, f32 accumulation):
//   y_1 = TDNN_1(c_1),   y_j = TDNN_j(c_j + y_{j-1})  (j = 2..n),   TDNN(u) = BN(ReLU(conv_{k=3, dilation d}(u) + bias))
// where c_j is channel chunk j of the tdnn1 output r [B*T][ld] and y_j is written over it — the seven dependent
// 128 -> 128 convs speechbrain's Res2NetBlock runs inside EncoderClassifier.encode_batch [REF speech_encode.py:77]
// (SURVEY.md Appendix A.3).  As seven launches of the 128x128 GEMM kernel each conv moved ~1 GB through HBM for
// 66 GFLOP (input chunk, next chunk, y, the "tee" copy), ran the matrix pipe at 0.24 and was all prologue / epilogue.
//
// One workgroup = one segment.  The segment's current chain input u_j [T][128] f16 (51 KB at T = 201) lives in LDS for
// the whole chain; a conv reads it as the MFMA B operand with the rows gathered at reflect(t + (tap - 1) d) — the
// whole segment is resident, so the dilation needs no halo — and the weights (96 KB per conv, L2-resident, shared by
// every workgroup) stream from global memory straight into the A-operand registers.
//   * v_mfma_f32_32x32x16_f16 with the WEIGHTS as the A operand: accumulator rows are output channels, columns are
//     time rows, so a lane ends up with 4 consecutive channels of one time row per 8-channel group: 8-byte LDS
//     accesses in the epilogue.
//   * FOUR MFMA waves (one per SIMD), each 32 output channels over ALL 32-row time tiles: a conv's weights enter the CU
//     once — in MFMA-fragment order from a repacked copy (chain_pack_kernel: 1 KB contiguous per wave instruction; read
//     as strided rows of the [cout][3][128] layout they arrived at ~18 B/clk per CU, 5.3 k cycles per conv) — every
//     activation fragment is read by 4 waves (0.66 MB of LDS reads per conv), and with 7 waves per workgroup a wave may
//     hold 256 registers: fragment tt of K step s + 1 is read right behind MFMA tt of step s (two fragment sets).
//   * Three LDS buffers: u_j (being read), the next chunk c_{j+1} (turned into u_{j+1} = c_{j+1} + y_j in place by
//     the epilogue, same rounding as the unfused path: f16(y_f32 + c)), and a staging copy of y_j.
//   * Three more waves do every long-latency memory operation: one DMAs chunk j + 2 into the buffer conv j has finished
//     with (global_load_lds, a whole K loop ahead of its use), two copy the staged y_j to HBM in full 256-byte rows
//     and never wait for their stores; all three share the DMA of c_1 at the start.  The MFMA waves issue only L2-hit
//     weight / parameter loads and no stores, so none of their waits ever sits behind an HBM round trip (vmcnt is in
//     order; a wait after a store waits for its acknowledgement).
//   * Two workgroup barriers per conv: A (K loop done, c_{j+1} landed) and B (u_{j+1} and the y_j staging complete).
// Measured (MI355X, 5000 segments of T = 201, tools/time_chain.py, tools/stamp_res2.py): 1.33 ms per block against 1.3 ms
// for the seven launches it replaces and 1.45-1.59 ms for the first version of this kernel (8 waves x 16 channels on
// v_mfma_f32_16x16x32_f16, every wave reading every activation fragment, 168 registers: no read-ahead).  Per conv and
// workgroup (stamps, cycles): K loop 9.4 k — 10.3 k with the reads issued as a block, 7.2 k with no memory operation at
// all, i.e. the matrix pipe itself issues one 32x32x16 MFMA per 43 cycles here, not 32 — epilogue 6.8 k (VALU 3.4 k,
// one wave per SIMD hides nothing), start of the chain 4.4 k.  The same MFMA shape on 8 waves (4 channel groups x 2
// interleaved halves of the tiles, two per SIMD, 168 registers) was slower: 1.53 ms, K loop 10.4 k, 14 spills.
// Epilogue (j) and K loop (j + 1) cannot overlap inside a segment (a conv needs every row and channel of u_{j+1});
// two segments per workgroup could, which two segments' buffers (6 x 51 KB) do not allow.  HBM floor of ANY schedule:
// 3.6 GB per block, 0.7 ms at 5 TB/s.
// LDS image of a buffer: rows of 256 bytes, 16-byte chunk q of row t stored at chunk q ^ (t & 15): the fragment reads
// (16 lanes = 16 consecutive rows, one chunk) are bank-conflict free, and the DMA fills it by permuting its per-lane
// SOURCE address.
#include <cstdlib>

#include "sd_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int RC_MAXN = 7;        // convs per chain (res2net_scale - 1)
constexpr int RC_NT = 7;          // most 32-row time tiles of a segment (the kernel is instantiated for 1..7)
constexpr int RC_CH = 128;        // channels per chunk
constexpr int RC_ROWB = RC_CH * 2;
constexpr int RC_CPU = 4;         // 1 KB pieces the copy-out wave keeps in flight
constexpr int RC_LDS_MAX = 160 * 1024;
constexpr int RC_PACK_HALFS = RC_CH * 3 * RC_CH;   // 96 KB per conv

struct ChainArgs {
  _Float16* r;
  int ld, T, n, dil;
  const _Float16* wpk;              // the convs' weights in MFMA-fragment order (chain_pack_kernel), RC_PACK_HALFS per conv
  const _Float16* w[RC_MAXN];
  const float* bias[RC_MAXN];
  const float* scale[RC_MAXN];
  const float* shift[RC_MAXN];
};

#define RC_GLDS16(gptr, lptr)                                                              \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),  \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

__device__ __forceinline__ void rc_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

#ifdef SD_STAMP
// diagnostic build only: per-workgroup cycle counters of MFMA wave 0 [K loops, barrier A, epilogues, barrier B, total],
// read by tools/stamp_res2.py
__device__ unsigned long long sd_res2_stamp_buf[4096 * 32];
#define RC_T(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tprev; tprev = now_; } while (0)
#define RC_ARRIVE(j_, slot_) do { if ((j_) == 3 && lane == 0 && blockIdx.x < 4096) sd_res2_stamp_buf[blockIdx.x * 32 + (slot_)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RC_ARRIVE(j_, slot_) do { } while (0)
#define RC_T(i) do { } while (0)
#endif

// K order per tap: pairs of steps over 32 channels; lane (row, half h) takes channels 32 p + 16 h + 8 e .. + 7 in step
// 2 p + e, so its two weight fragments of a pair are 32 contiguous bytes and a fragment's LDS chunk is 4 p + 2 h + e.
constexpr int RC_THREADS = 448;  // 4 MFMA waves + a DMA wave + 2 copy-out waves
constexpr int RC_PF = 3;         // step pairs the weight fragments are fetched ahead (~2.7 k cycles: a loaded L2 hit with one wave per SIMD to hide it)

template <int NT>                 // 32-row time tiles
__global__ __launch_bounds__(RC_THREADS) void res2net_chain_f16_kernel(const ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef float f32x16v __attribute__((ext_vector_type(16)));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T;
  const int rows4 = (T + 3) & ~3;
  const int BUF = rows4 * RC_ROWB;
  char* const buf_y = smem + 2 * BUF;
  _Float16* const R = a.r + (size_t)blockIdx.x * T * a.ld;
  const int n = a.n;

  if (wid == 4) {
    // ------------------------------------------------------------------ DMA wave: chunk j + 2 -> the buffer conv j has finished with
    const int lrow = lane >> 4, lq = lane & 15;
    auto dma_chunk = [&](int chunk, char* buf) {
      for (int i = 0; i < rows4; i += 4) {
        int row = i + lrow;
        row = row < T ? row : T - 1;
        RC_GLDS16(R + (size_t)row * a.ld + chunk * RC_CH + ((lq ^ (row & 15)) << 3), buf + i * RC_ROWB);
      }
    };
    // u_1 = c_1 -> buffer 1: a third of the rows each by this wave and the two copy-out waves (one wave issues a 1 KB
    // piece every ~85 cycles: the chain's start waited 10 k cycles for one wave to issue c_1 and c_2); c_2 follows S
    for (int i = 0; i < rows4; i += 12) {
      int row = i + lrow;
      row = row < T ? row : T - 1;
      RC_GLDS16(R + (size_t)row * a.ld + 1 * RC_CH + ((lq ^ (row & 15)) << 3), smem + BUF + i * RC_ROWB);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    rc_barrier();                                      // S
    if (n >= 2) dma_chunk(2, smem);                    // c_2 -> buffer 0 (needed by the epilogue of conv 1)
    for (int j = 1; j <= n; ++j) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // c_{j+1} has landed
      rc_barrier();                                    // A(j)
      if (j + 2 <= n) dma_chunk(j + 2, smem + (j & 1) * BUF);
      rc_barrier();                                    // B(j)
    }
    return;
  }
  if (wid >= 5) {
    // ------------------------------------------------------------------ copy-out waves: staged y_j -> r[:, chunk j] in whole 256-byte rows.
    // Their stores are never waited for: only their LDS reads must be done before the next epilogue refills the
    // staging buffer, which rc_barrier's lgkmcnt(0) ensures.
    const int lrow = lane >> 4, lq = lane & 15;
    const int part = wid - 5;
    for (int i = 4 * (part + 1); i < rows4; i += 12) {  // this wave's third of c_1 (see the DMA wave)
      int row = i + lrow;
      row = row < T ? row : T - 1;
      RC_GLDS16(R + (size_t)row * a.ld + 1 * RC_CH + ((lq ^ (row & 15)) << 3), smem + BUF + i * RC_ROWB);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    rc_barrier();                                      // S
    for (int j = 1; j <= n; ++j) {
      rc_barrier();                                    // A(j)
      rc_barrier();                                    // B(j): y_j is staged
      for (int i0 = 4 * RC_CPU * part; i0 < rows4; i0 += 8 * RC_CPU) {
        h8 v[RC_CPU];
#pragma unroll
        for (int k = 0; k < RC_CPU; ++k) {
          const int i = i0 + 4 * k < rows4 ? i0 + 4 * k : rows4 - 4;
          v[k] = *reinterpret_cast<const h8*>(buf_y + i * RC_ROWB + lane * 16);
        }
#pragma unroll
        for (int k = 0; k < RC_CPU; ++k) {
          const int row = i0 + 4 * k + lrow;
          if (i0 + 4 * k < rows4 && row < T)
            *reinterpret_cast<h8*>(R + (size_t)row * a.ld + j * RC_CH + ((lq ^ (row & 15)) << 3)) = v[k];
        }
      }
    }
    return;
  }

  // -------------------------------------------------------------------- MFMA waves: wave cg owns channels 32 cg .. + 31, all rows
  const int fn = lane & 31, fh = lane >> 5;
  const int cg = wid;
  constexpr int NTW = NT;
  // weights in fragment order: [conv][wave][pair 0..11][half e][lane][8]: a wave's fragment is 1 KB contiguous (read
  // as strided rows of the [cout][3][128] layout, 16 bytes at 32 places per instruction, the fragments arrived at
  // ~18 B/clk per CU: 5.3 k cycles per conv, as long as the matrix work itself)
  const size_t w_off = (size_t)cg * 12 * 2 * 512 + lane * 8;
  h8 wa[RC_PF + 1][2];                                // weight fragment pairs in flight (rotating, 12 % (RC_PF + 1) == 0)
  static_assert(12 % (RC_PF + 1) == 0, "the fragment ring must line up across convs");
#ifdef SD_STAMP
  unsigned long long tacc[5] = {0, 0, 0, 0, 0};
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
  rc_barrier();                                        // S
#ifdef SD_STAMP
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
  tacc[4] = tprev - t_entry;
#endif
  for (int j = 1; j <= n; ++j) {
    RC_ARRIVE(j, 11 + (wid == 0 ? 0 : 20));
    const char* cur = smem + (j & 1) * BUF;
    char* nxt = smem + ((j + 1) & 1) * BUF;
    const _Float16* Wj = a.wpk + (size_t)(j - 1) * RC_PACK_HALFS + w_off;
    f32x16v acc[NTW];
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;

    if (j == 1) {
#pragma unroll
      for (int pp = 0; pp < RC_PF; ++pp) {
        wa[pp][0] = *reinterpret_cast<const h8*>(Wj + pp * 1024);
        wa[pp][1] = *reinterpret_cast<const h8*>(Wj + pp * 1024 + 512);
      }
    }
    int base[NTW];                                     // LDS address of chunk 2 fh of this lane's gathered row, per tile
    auto set_tap = [&](int tap) {
      const int delta = (tap - 1) * a.dil;
      int fnq = fn;
      asm volatile("" : "+v"(fnq));                    // opaque per tap: gather addresses are not invariants of the chain loop
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) {
        int tr = tt * 32 + fnq;
        tr = (tr < T ? tr : T - 1) + delta;
        tr = tr < 0 ? -tr : tr;
        tr = tr >= T ? 2 * (T - 1) - tr : tr;
        base[tt] = tr * RC_ROWB + (((2 * fh) ^ (tr & 15)) << 4);
      }
    };
    h8 xb[2][NTW];
    set_tap(0);
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) xb[0][tt] = *reinterpret_cast<const h8*>(cur + base[tt]);
#pragma unroll
    for (int s = 0; s < 24; ++s) {                     // step s: tap s / 8, pair (s / 2) % 4, half e = s % 2
      if ((s & 1) == 0) {                              // weights: RC_PF pairs ahead; past this conv's last pair: the next conv's first
        const int pp = s >> 1;
        const int qq = (pp + RC_PF) % 12;
#ifndef RC_DIAG_NOW
        const _Float16* Wn = pp + RC_PF < 12 || j >= n ? Wj : Wj + RC_PACK_HALFS;
        wa[(pp + RC_PF) % (RC_PF + 1)][0] = *reinterpret_cast<const h8*>(Wn + qq * 1024);
        wa[(pp + RC_PF) % (RC_PF + 1)][1] = *reinterpret_cast<const h8*>(Wn + qq * 1024 + 512);
#else
        (void)qq;
#endif
      }
      // MFMA tt of this step, then the read of fragment tt of step s + 1 (reads issued as a block in front of the MFMAs
      // are matrix-pipe idle time)
      if (s + 1 < 24 && ((s + 1) & 7) == 0) set_tap((s + 1) >> 3);
      const int sn = s + 1;
      const int x = ((4 * ((sn >> 1) & 3)) + (sn & 1)) << 4;
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) {
        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[(s >> 1) % (RC_PF + 1)][s & 1], xb[s & 1][tt], acc[tt], 0, 0, 0);
#ifndef RC_DIAG_NOLDS
        if (s + 1 < 24) xb[sn & 1][tt] = *reinterpret_cast<const h8*>(cur + (base[tt] ^ x));
#endif
      }
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA ...
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);     // ... the address of the next read ...
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // ... and the read
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // per-channel parameters of this lane's 16 channels (32 w + 8 g + 4 fh + r): fetched here so the wait at the barrier covers their latency
    f32x4v pb[4], ps[4], ph[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * cg + 8 * g + 4 * fh;
#ifdef RC_DIAG_NOPARAM
      pb[g] = f32x4v{0.1f, 0.f, 0.2f, 0.f}; ps[g] = f32x4v{1.f, 0.9f, 1.f, 1.1f}; ph[g] = f32x4v{0.f, 0.1f, 0.f, 0.f}; (void)c;
      continue;
#endif
      pb[g] = a.bias[j - 1] ? *reinterpret_cast<const f32x4v*>(a.bias[j - 1] + c) : f32x4v{0.f, 0.f, 0.f, 0.f};
      ps[g] = a.scale[j - 1] ? *reinterpret_cast<const f32x4v*>(a.scale[j - 1] + c) : f32x4v{1.f, 1.f, 1.f, 1.f};
      ph[g] = a.shift[j - 1] ? *reinterpret_cast<const f32x4v*>(a.shift[j - 1] + c) : f32x4v{0.f, 0.f, 0.f, 0.f};
    }
    RC_T(0);
    RC_ARRIVE(j, wid);
    rc_barrier();                                      // A(j)
    RC_ARRIVE(j, 10 + (wid == 0 ? 0 : 20));
    RC_T(1);
    const bool more = j < n;
    int fne = fn;
    asm volatile("" : "+v"(fne));                      // opaque per conv (as in the K loop)
    // one wave per SIMD: nothing hides an LDS round trip, so the next tile's c_{j+1} values are read while this tile is
    // computed, and only the last tile (the one that can hold rows past T) predicates its stores
    auto tile_row = [&](int tt, int& rowo, int& sw) {
      int t = tt * 32 + fne;
      t = t < T ? t : T - 1;
      rowo = t * RC_ROWB + 8 * fh;
      sw = t & 15;
    };
    h4 cn[2][4];
    int rowo, sw;
    tile_row(0, rowo, sw);
    if (more) {
#pragma unroll
      for (int g = 0; g < 4; ++g) cn[0][g] = *reinterpret_cast<const h4*>(nxt + rowo + (((4 * cg + g) ^ sw) << 4));
    }
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) {
      int rown = rowo, swn = sw;
      if (tt + 1 < NTW) {
        tile_row(tt + 1, rown, swn);
        if (more) {
#pragma unroll
          for (int g = 0; g < 4; ++g) cn[(tt + 1) & 1][g] = *reinterpret_cast<const h4*>(nxt + rown + (((4 * cg + g) ^ swn) << 4));
        }
      }
      const bool live = tt + 1 < NT || tt * 32 + fn < T;      // (T > 32 (NT - 1): compile-time true except for the last tile)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int eo = rowo + (((4 * cg + g) ^ sw) << 4);
        h4 y, u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = sd_max_keep_nan(acc[tt][4 * g + r] + pb[g][r], 0.f) * ps[g][r] + ph[g][r];
          y[r] = (_Float16)v;
          u[r] = (_Float16)(v + (more ? (float)cn[tt & 1][g][r] : 0.f));
        }
        if (live) {
          *reinterpret_cast<h4*>(buf_y + eo) = y;
          if (more) *reinterpret_cast<h4*>(nxt + eo) = u;
        }
      }
      rowo = rown; sw = swn;
    }
    RC_T(2);
    rc_barrier();                                      // B(j)
    RC_T(3);
  }
#ifdef SD_STAMP
  if (tid == 0 && blockIdx.x < 4096) {
    for (int i = 0; i < 5; ++i) sd_res2_stamp_buf[blockIdx.x * 32 + 20 + i] = tacc[i];
    sd_res2_stamp_buf[blockIdx.x * 32 + 25] = __builtin_amdgcn_s_memtime() - t_entry;
  }
#endif
}

// [cout][3][128] weights of the chain's convs -> fragment order of res2net_chain_f16_kernel: thread = (conv, wave, pair,
// half, lane); lane (row fn, half-row fh) of channel group w takes W[32 w + fn][tap][32 p + 16 fh + 8 e .. + 7], pair = 4 tap + p
__global__ __launch_bounds__(256) void chain_pack_kernel(const ChainArgs a, _Float16* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;      // n * 6144 threads
  if (idx >= a.n * 6144) return;
  const int lane = idx & 63, e = (idx >> 6) & 1, pp = (idx >> 7) % 12, w = (idx / (128 * 12)) & 3, j = idx / 6144;
  const int fn = lane & 31, fh = lane >> 5;
  const _Float16* src = a.w[j] + ((size_t)(32 * w + fn) * 3 + (pp >> 2)) * RC_CH + 32 * (pp & 3) + 16 * fh + 8 * e;
  *reinterpret_cast<h8*>(out + (size_t)idx * 8) = *reinterpret_cast<const h8*>(src);
}

}  // namespace

#ifdef SD_STAMP
extern "C" int sd_debug_read_res2_stamps(unsigned long long* out, int n) {
  SD_CHECK_HIP(hipDeviceSynchronize());
  SD_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(sd_res2_stamp_buf), (size_t)n * sizeof(unsigned long long)));
  return SD_OK;
}
#endif

extern "C" int sd_res2net_chain_supported(int T, int chunk, int n, int taps, int dil) {
  if (chunk != RC_CH || taps != 3 || n < 1 || n > RC_MAXN) return 0;
  if (T < 2 || dil < 1 || dil >= T) return 0;
  if (((T + 31) >> 5) > RC_NT) return 0;
  return 3 * ((T + 3) & ~3) * RC_ROWB <= RC_LDS_MAX;
}

extern "C" size_t sd_res2net_chain_workspace_bytes(int n) {
  return n < 1 || n > RC_MAXN ? 0 : (size_t)n * RC_PACK_HALFS * sizeof(_Float16);
}

extern "C" int sd_res2net_chain_f16(void* r, int ld, int B, int T, const sd_layer* layers, int n, void* ws, size_t ws_bytes, sd_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SD_CHECK_ARG(r != nullptr && layers != nullptr && ws != nullptr, "sd_res2net_chain_f16: null pointer");
  SD_CHECK_ARG(sd_aligned16(ws) && ws_bytes >= sd_res2net_chain_workspace_bytes(n), "sd_res2net_chain_f16: workspace %zu < %zu bytes (or not 16-byte aligned)",
               ws_bytes, sd_res2net_chain_workspace_bytes(n));
  SD_CHECK_ARG(B >= 0 && ld >= (n + 1) * RC_CH && ld % 8 == 0 && sd_aligned16(r), "sd_res2net_chain_f16: B=%d ld=%d n=%d (need ld >= (n + 1) * 128, ld %% 8 == 0, 16-byte aligned r)", B, ld, n);
  if (n < 1 || n > RC_MAXN || !sd_res2net_chain_supported(T, layers[0].cout, n, layers[0].taps, layers[0].dil))
    return sd_set_error(SD_ERR_UNSUPPORTED, "sd_res2net_chain_f16: needs 1..7 convs of 128 -> 128, k = 3, dilation < T, T <= 212 (T=%d n=%d cout=%d taps=%d dil=%d)",
                        T, n, layers[0].cout, layers[0].taps, layers[0].dil);
  ChainArgs a = {};
  a.r = static_cast<_Float16*>(r);
  a.ld = ld; a.T = T; a.n = n; a.dil = layers[0].dil;
  for (int j = 0; j < n; ++j) {
    const sd_layer& l = layers[j];
    SD_CHECK_ARG(l.w && l.w_dtype == SD_DT_F16 && l.cin == RC_CH && l.cin_pad == RC_CH && l.cout == RC_CH && l.taps == 3 && l.dil == a.dil,
                 "sd_res2net_chain_f16: layer %d must be f16 128 -> 128, k = 3, dilation %d (cin=%d cin_pad=%d cout=%d taps=%d dil=%d dtype=%d)",
                 j, a.dil, l.cin, l.cin_pad, l.cout, l.taps, l.dil, l.w_dtype);
    SD_CHECK_ARG(sd_aligned16(l.w) && sd_aligned16(l.bias) && sd_aligned16(l.scale) && sd_aligned16(l.shift), "sd_res2net_chain_f16: layer %d parameters must be 16-byte aligned", j);
    a.w[j] = static_cast<const _Float16*>(l.w);
    a.bias[j] = l.bias; a.scale[j] = l.scale; a.shift[j] = l.shift;
  }
  if (B == 0) return SD_OK;
  a.wpk = static_cast<const _Float16*>(ws);
  hipLaunchKernelGGL(chain_pack_kernel, dim3((unsigned)(n * 24)), dim3(256), 0, stream, a, static_cast<_Float16*>(ws));
  SD_CHECK_LAUNCH("chain_pack_kernel");
  const int lds = 3 * ((T + 3) & ~3) * RC_ROWB;
  {
    void (*kern)(const ChainArgs) = nullptr;
    switch ((T + 31) >> 5) {                                   // 32-row time tiles
      case 1: kern = res2net_chain_f16_kernel<1>; break;
      case 2: kern = res2net_chain_f16_kernel<2>; break;
      case 3: kern = res2net_chain_f16_kernel<3>; break;
      case 4: kern = res2net_chain_f16_kernel<4>; break;
      case 5: kern = res2net_chain_f16_kernel<5>; break;
      case 6: kern = res2net_chain_f16_kernel<6>; break;
      default: kern = res2net_chain_f16_kernel<7>; break;
    }
    SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(kern), RC_LDS_MAX));
    {
      SdProfScope prof(SD_PROF_CONV_GEMM, stream, 2.0 * (double)B * T * RC_CH * 3 * RC_CH * n);
      hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(RC_THREADS), lds, stream, a);
    }
    SD_CHECK_LAUNCH("res2net_chain_f16_kernel");
    return SD_OK;
  }
}

