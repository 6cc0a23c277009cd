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
//   * v_mfma_f32_16x16x32_f16 with the WEIGHTS as the A operand: accumulator rows are output channels, columns are
//     time rows, so a lane ends up with 4 consecutive channels of one time row: 8-byte LDS accesses in the epilogue.
//   * 8 MFMA waves = 8 channel groups of 16, each over ALL time tiles: per K step of 32 a wave loads one weight
//     fragment and <= 14 activation fragments for as many MFMAs, so a conv's 96 KB of weights enter the CU once
//     (with 4 channel groups x 2 time halves they entered twice, 192 KB per conv at ~18 B/clk = 10.7 k cycles, twice
//     the matrix time); the price is every wave reading every activation fragment (1 KB of LDS per MFMA).
//   * Three LDS buffers: u_j (being read), the next chunk c_{j+1} (turned into u_{j+1} = c_{j+1} + y_j in place by
//     the epilogue, same rounding as the unfused path: f16(y_f32 + c)), and a staging copy of y_j.
//   * Two more waves do every long-latency memory operation: one DMAs chunk j + 2 into the buffer conv j has finished
//     with (global_load_lds, a whole K loop ahead of its use), one copies the staged y_j to HBM in full 256-byte rows
//     and never waits for its stores.  The MFMA waves issue only L2-hit weight / parameter loads and no stores, so
//     none of their waits ever sits behind an HBM round trip (vmcnt is in order; a wait after a store waits for its
//     acknowledgement: with one wave doing both, the chain ran at a quarter of the matrix rate).
//   * Two workgroup barriers per conv: A (K loop done, c_{j+1} landed) and B (u_{j+1} and the y_j staging complete).
// Measured (MI355X, 5000 segments of T = 201, tools/time_chain.py, tools/stamp_res2.py): 1.47-1.59 ms per block (run to
// run) against 1.3 ms for the seven launches it replaces: not faster by itself, but 21 fewer launches per forward, half
// the HBM traffic (3.6 GB per block, 0.7 ms at 5 TB/s is the floor of ANY schedule) and tdnn1 freed of its tee epilogue
// (the f16 step as a whole gained 1 %).  Per conv a workgroup spends ~13 k cycles where the matrix pipe needs 5 k
// (stamps): K loop 10 k on the older waves / 13 k on the younger ones (with every wave reading every activation
// fragment the LDS pipe is as loaded as the matrix pipe), epilogue 3.6 k, barrier skew.  What would change the picture
// is two segments per workgroup (weights and barriers amortised), which two segments' buffers (6 x 51 KB) do not allow.
// LDS image of a buffer: rows of 256 bytes, 16-byte chunk q of row t stored at chunk q ^ (t & 15): the fragment reads
// (16 lanes = 16 consecutive rows, one chunk) and the epilogue's 8-byte accesses are bank-conflict free, and the DMA
// fills it by permuting its per-lane SOURCE address.
#include "sd_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int RC_MAXN = 7;        // convs per chain (res2net_scale - 1)
constexpr int RC_NT = 14;         // most 16-row time tiles of a segment (the kernel is instantiated for 1..14)
constexpr int RC_CH = 128;        // channels per chunk
constexpr int RC_ROWB = RC_CH * 2;
constexpr int RC_THREADS = 704;   // 8 MFMA waves + a DMA wave + 2 copy-out waves
constexpr int RC_CPU = 4;         // 1 KB pieces the copy-out wave keeps in flight
constexpr int RC_PF = 2;          // K steps the weight fragments are fetched ahead
constexpr int RC_LDS_MAX = 160 * 1024;

struct ChainArgs {
  _Float16* r;
  int ld, T, n, dil;
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

// NT = ceil(T / 16) time tiles: the tile loops carry no run-time bounds (a branch per tile cut the K loop into
// read -> wait -> MFMA blocks, 2.5x the time of the pipe); the last tile's rows past T re-read clamped rows and are
// dropped in the epilogue.
template <int NT>
__global__ __launch_bounds__(RC_THREADS) void res2net_chain_f16_kernel(const ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T;
  const int rows4 = (T + 3) & ~3;
  const int BUF = rows4 * RC_ROWB;
  char* const buf_y = smem + 2 * BUF;
  _Float16* const R = a.r + (size_t)blockIdx.x * T * a.ld;
  const int n = a.n;

  if (wid == 8) {
    // ------------------------------------------------------------------ DMA wave: chunk j + 2 -> the buffer conv j has finished with
    const int lrow = lane >> 4, lq = lane & 15;
    auto dma_chunk = [&](int chunk, char* buf) {       // r[:, chunk] -> LDS image (4 rows per instruction)
      for (int i = 0; i < rows4; i += 4) {
        int row = i + lrow;
        row = row < T ? row : T - 1;
        RC_GLDS16(R + (size_t)row * a.ld + chunk * RC_CH + ((lq ^ (row & 15)) << 3), buf + i * RC_ROWB);
      }
    };
    dma_chunk(1, smem + BUF);                          // u_1 = c_1 -> buffer 1
    if (n >= 2) dma_chunk(2, smem);                    // c_2 -> buffer 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    rc_barrier();                                      // S
    for (int j = 1; j <= n; ++j) {
      RC_ARRIVE(j, 8);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // c_{j+1} has landed
      RC_ARRIVE(j, 18);
      rc_barrier();                                    // A(j): every wave has finished reading u_j
      if (j + 2 <= n) dma_chunk(j + 2, smem + (j & 1) * BUF);
      rc_barrier();                                    // B(j)
    }
    return;
  }
  if (wid >= 9) {
    // ------------------------------------------------------------------ copy-out wave: staged y_j -> r[:, chunk j] in whole 256-byte rows.
    // Its stores are never waited for (a wait behind a store is an HBM round trip): only its LDS reads must be done
    // before the next epilogue refills the staging buffer, which rc_barrier's lgkmcnt(0) ensures.
    const int lrow = lane >> 4, lq = lane & 15;
    const int part = wid - 9;                          // each copy wave takes every other group of RC_CPU pieces
    rc_barrier();                                      // S
    for (int j = 1; j <= n; ++j) {
      if (part == 0) RC_ARRIVE(j, 9);
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

  // -------------------------------------------------------------------- MFMA waves: wave w owns channels 16 w .. 16 w + 15, all rows
  const int fr = lane & 15, fq = lane >> 4;
  // epilogue addressing: channels 16 w + 4 fq + (0..3) of row t -> 8 bytes at chunk 2 w + (fq >> 1)
  const int ep_chunk0 = 2 * wid + (fq >> 1), ep_sub = 8 * (fq & 1);
  const int ch0 = 16 * wid + 4 * fq;

  const size_t w_off = ((size_t)(16 * wid + fr) * 3) * RC_CH + 8 * fq;       // this lane's row of the packed [cout][3][128] weights
  h8 wa[RC_PF + 1];                                    // weight fragments in flight (rotating, 12 % (RC_PF + 1) == 0)
  static_assert(12 % (RC_PF + 1) == 0, "the fragment ring must line up across convs");
#ifdef SD_STAMP
  unsigned long long tacc[5] = {0, 0, 0, 0, 0};
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
  rc_barrier();                                        // S
#ifdef SD_STAMP
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
  tacc[4] = tprev - t_entry;                           // prologue: waiting for c_1 / c_2
#endif
  for (int j = 1; j <= n; ++j) {
    RC_ARRIVE(j, 11 + (wid == 0 ? 0 : 20));            // wave 0 enters K loop j
    const char* cur = smem + (j & 1) * BUF;
    char* nxt = smem + ((j + 1) & 1) * BUF;
    const _Float16* Wj = a.w[j - 1] + w_off;
    f32x4v acc[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[tt][r] = 0.f;

    if (j == 1) {
#pragma unroll
      for (int q = 0; q < RC_PF; ++q) wa[q] = *reinterpret_cast<const h8*>(Wj + (q >> 2) * RC_CH + (q & 3) * 32);
    }
    // K steps of 32: tap = q / 4, channels 32 (q % 4) ...  All NT activation fragments of a step are read before its MFMAs
    // (scheduling barriers keep hipcc from re-serialising them into read -> wait -> MFMA with one read in flight, which
    // ran the chain at 2x the time of the matrix pipe); the SIMD's other wave covers the first read's latency.
    int base[NT];                                      // LDS address of this lane's k chunk fq of the gathered row, per tile
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      {                                                // weights: RC_PF steps ahead; past this conv's last step: the next conv's first
        const int qq = (q + RC_PF) % 12;
        const _Float16* Wn = q + RC_PF < 12 || j >= n ? Wj : a.w[j] + w_off;
        wa[(q + RC_PF) % (RC_PF + 1)] = *reinterpret_cast<const h8*>(Wn + (qq >> 2) * RC_CH + (qq & 3) * 32);
      }
      if ((q & 3) == 0) {
        const int delta = ((q >> 2) - 1) * a.dil;
        int frq = fr;
        asm volatile("" : "+v"(frq));                  // opaque per tap: a conv's gather addresses are not loop invariants
                                                       // to be hoisted out of the chain loop and spilled
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          int tr = tt * 16 + frq;
          tr = (tr < T ? tr : T - 1) + delta;
          tr = tr < 0 ? -tr : tr;
          tr = tr >= T ? 2 * (T - 1) - tr : tr;
          base[tt] = tr * RC_ROWB + ((fq ^ (tr & 15)) << 4);
        }
      }
      h8 xb[NT];
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)                  // chunk 4 s + fq of the row sits at (4 s + fq) ^ sw = (fq ^ sw) ^ 4 s
        xb[tt] = *reinterpret_cast<const h8*>(cur + (base[tt] ^ ((q & 3) << 6)));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[q % (RC_PF + 1)], xb[tt], acc[tt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // per-channel parameters of this lane's 4 channels: fetched here so the wait at the barrier covers their latency
    const f32x4v pb = a.bias[j - 1] ? *reinterpret_cast<const f32x4v*>(a.bias[j - 1] + ch0) : f32x4v{0.f, 0.f, 0.f, 0.f};
    const f32x4v ps = a.scale[j - 1] ? *reinterpret_cast<const f32x4v*>(a.scale[j - 1] + ch0) : f32x4v{1.f, 1.f, 1.f, 1.f};
    const f32x4v ph = a.shift[j - 1] ? *reinterpret_cast<const f32x4v*>(a.shift[j - 1] + ch0) : f32x4v{0.f, 0.f, 0.f, 0.f};
    RC_T(0);
    RC_ARRIVE(j, wid);
    rc_barrier();                                      // A(j)
    RC_ARRIVE(j, 10 + (wid == 0 ? 0 : 20));
    RC_T(1);
    const bool more = j < n;
    int fre = fr;
    asm volatile("" : "+v"(fre));                      // opaque per conv (as in the K loop): not hoisted out of the chain loop
    constexpr int EC = 4;                              // tiles per epilogue chunk (register budget: 168 with 11 waves)
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += EC) {
      int eo[EC];                                      // this lane's 8 bytes of each tile
      h4 cn[EC];
#pragma unroll
      for (int i = 0; i < EC; ++i) {
        if (t0 + i < NT) {
          int t = (t0 + i) * 16 + fre;
          t = t < T ? t : T - 1;
          eo[i] = t * RC_ROWB + ((ep_chunk0 ^ (t & 15)) << 4) + ep_sub;
          if (more) cn[i] = *reinterpret_cast<const h4*>(nxt + eo[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < EC; ++i) {
        if (t0 + i < NT) {
          const bool live = (t0 + i) * 16 + fr < T;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[t0 + i][r] + pb[r], 0.f) * ps[r] + ph[r];
          h4 y, u;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            y[r] = (_Float16)v[r];
            u[r] = (_Float16)(v[r] + (more ? (float)cn[i][r] : 0.f));
          }
          if (live) {
            *reinterpret_cast<h4*>(buf_y + eo[i]) = y;
            if (more) *reinterpret_cast<h4*>(nxt + eo[i]) = u;
          }
        }
      }
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
  if (((T + 15) >> 4) > RC_NT) return 0;
  return 3 * ((T + 3) & ~3) * RC_ROWB <= RC_LDS_MAX;
}

extern "C" int sd_res2net_chain_f16(void* r, int ld, int B, int T, const sd_layer* layers, int n, sd_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SD_CHECK_ARG(r != nullptr && layers != nullptr, "sd_res2net_chain_f16: null pointer");
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
  const int lds = 3 * ((T + 3) & ~3) * RC_ROWB;
  const int nt = (T + 15) >> 4;                                // 16-row time tiles
  void (*kern)(const ChainArgs) = nullptr;
  switch (nt) {
#define RC_CASE(N_) case N_: kern = res2net_chain_f16_kernel<N_>; break;
    RC_CASE(1) RC_CASE(2) RC_CASE(3) RC_CASE(4) RC_CASE(5) RC_CASE(6) RC_CASE(7)
    RC_CASE(8) RC_CASE(9) RC_CASE(10) RC_CASE(11) RC_CASE(12) RC_CASE(13)
#undef RC_CASE
    default: kern = res2net_chain_f16_kernel<14>; break;
  }
  SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(kern), RC_LDS_MAX));
  {
    SdProfScope prof(SD_PROF_CONV_GEMM, stream, 2.0 * (double)B * T * RC_CH * 3 * RC_CH * n);
    hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(RC_THREADS), lds, stream, a);
  }
  SD_CHECK_LAUNCH("res2net_chain_f16_kernel");
  return SD_OK;
}
