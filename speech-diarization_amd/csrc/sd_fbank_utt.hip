// Log-mel filterbank in ONE launch: one workgroup per utterance, factored DFT, floor + mean removal in LDS.
//
// Same arithmetic contract as sd_fbank.hip (torchaudio MelSpectrogram + log + mean removal [REF speech_encode.py:17-36], or
// the speechbrain Fbank + sentence mean-norm front end of EncoderClassifier.encode_batch [REF speech_encode.py:77]; SURVEY.md
// Appendix A.1 / A.2); what changes is the shape of the work.  The folded-DFT kernel there streams a 364 KB basis through LDS
// for every 128 frames (one L2 -> LDS round trip and one barrier per k step: matrix pipe 25 % busy) and needs a second launch
// for the utterance-level top_db floor and the mean over T, which re-reads and re-writes the whole output.  Here:
//
//  * One 256-thread workgroup owns one utterance.  Its padded signal (n + 400 samples) is staged ONCE into LDS, already
//    clamped, scaled by 2^10 and split into two f16 halves per sample (hi = f16(v), lo = f16(v - hi): one dword), at word
//    i + i / 160 (frame stride 161 words: the 32 frames of a wave tile fall on 32 distinct banks).  The log-mel rows of the
//    whole utterance are collected in LDS too (over the dead signal), so the utterance maximum, the top_db floor and the mean
//    over T are applied before the ONLY write of the output: algorithmic traffic (128 000 B read + 64 320 B written per 2 s
//    segment), no atomics, no second pass.
//  * The 400-point DFT is factored 400 = 16 x 25 (n = 25 n1 + n2, k = k1 + 16 k2):
//        Y[k1, n2] = sum_{n1 < 16} w[n] x[n] e^{-2 pi i k1 n / 400}                       (stage 1: K = 16 = one MFMA k step)
//        X[k1 + 16 k2] = sum_{n2 < 25} e^{-2 pi i n2 k2 / 25} Y[k1, n2]                     (stage 2: one 25-point DFT per k1)
//    and, x being real, Y[16 - k1, n2] = e^{-2 pi i n2 / 25} conj(Y[k1, n2]), so only k1 = 0..8 are computed and output 24 - k2 of
//    k1's 25-point DFT is conj(X[16 - k1 + 16 k2]): nine 25-point DFTs give all 201 bins.  Window and twiddles are folded into
//    the stage-1 matrices (one 32 x 16 matrix per n2, 50 KB in all, read from L2); the stage-2 matrix is the same for every k1
//    (64 x 64, 16 KB, LDS resident).  No basis streaming, no barrier inside a tile: 2 x 75 + 216 + 162 MFMAs of 32 cycles per
//    32-frame tile (stage 1 runs twice, see the register note below) against 546 + 126.
//  * Every product runs on the f16 matrix cores at f32 accuracy as in sd_fbank.hip: operands split hi + lo, three MFMAs
//    hi.hi + hi.lo + lo.hi with f32 accumulation (2^-21 relative per stage; the stage-1 sums are split again for stage 2),
//    power spectrum as split bf16 against split bf16 mel weights.  The MFMA flushes f16 subnormals, so every LOW half that matters
//    must be a normal number (>= 2^-14) and every value < 2^16.  Scales, all powers of two (exact): the samples by the 2^k that puts
//    the UTTERANCE's peak into [2^13, 2^14) (low halves normal down to 100 dB below the peak, at every signal level), both
//    matrices by 2^5, the stage-1 sums by 2^-8 before their split (< 2^15); the mel weights and one multiply in front of the log
//    take 2^(-2 (k + 2)) back.
//  * A wave (one per SIMD, up to 512 registers) owns a 32-frame tile: the stage-1 sums of its frames stay in registers, the
//    accumulator rows of stage 1 are stage 2's contraction index, and the two lane halves trade the problems k1 / k1 + 5 with
//    v_permlane32_swap so that each half supplies one n2 of a pair.  The split sums of all nine problems are 250 registers per lane,
//    more than the 256 architectural VGPRs leave room for: the problems run in two passes (k1 in {0,1,2,5,6,7}, then {3,4,8}), each
//    with its own run of stage 1 on its own table.
//
// Measured (MI355X, 10 000 segments of 2 s, same box): 2.10 ms against 2.66-2.70 ms for the folded kernel + finalize pass, i.e.
// 0.92 TB/s of algorithmic bytes = 0.115 of the 8 TB/s HBM figure, as ONE launch; max abs error against the float64 oracle
// 1.4e-4 (ln) / 3.9e-4 (dB), the same at every signal level.  What bounds it (in-kernel stamps, tools/stamp_fbank_utt.py): one wave
// per SIMD issues ~20 000 instructions per utterance at ~6 cycles each (VALU 4 cycles for a lone wave, the split / pack / swap work
// around each MFMA, waits that nothing else covers); the matrix pipe is ~30 % busy.  Two findings on the way: (1) fully unrolled
// (both passes as separate code, staging per group) the kernel was 91 KB of instructions, more than the 64 KB instruction cache
// two CUs share: the first tile round of every utterance ran at half the speed of the second until the passes became one
// run-time loop over two tables (49 KB); (2) the stage-2 matrix kept in registers (64, accumulator file) cost a
// v_accvgpr_read per register and use -- more issue slots than the 16 LDS reads per problem it saved.
//
// Takes utterances of up to 216 frames (the padded signal must fit the CU's 160 KB of LDS); longer ones run on the folded
// kernel + finalize pass of sd_fbank.hip.
#include "sd_fbank_internal.h"
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

namespace {

constexpr int NFFT = 400;
constexpr int HOP = 160;
constexpr int NFREQ = 201;
constexpr int NN2 = 25;          // n = 25 n1 + n2
constexpr int NPROB = 9;         // k1 = 0..8
constexpr int FT = 32;           // frames per wave tile
constexpr int MELP = 81;         // log-mel row stride in LDS
constexpr int UTT_MAX_T = 216;
constexpr int UTT_MAX_GROUPS = (((UTT_MAX_T * HOP - 1) + NFFT + 3) / 4 + 255) / 256;     // groups of 4 samples per thread (n <= 160 UTT_MAX_T - 1)
constexpr int A2_BYTES = 16 * 1024;
constexpr int SCRATCH_FLOATS = 1536;       // [0..7] wave maxima of the log-mel rows, [8..11] of |x|, [16..95] column means, [128..128 + 12 * 96) column partial sums, the last 64: spare words
constexpr int LDS_LIMIT = 160 * 1024;
constexpr float XMAX = 16.f;
constexpr double A1SCALE = 32.0;         // stage-1 matrices (|w cos| <= 1): the LOW halves of entries >= 2^-8 stay normal f16 numbers (the MFMA flushes f16 subnormals)
constexpr double A2SCALE = 32.0;         // stage-2 matrix, likewise
constexpr float YSCALE = 1.0f / 256.0f;  // stage-1 sums (< 2^23 for a peak < 2^14) -> < 2^15 before their split
constexpr double PSCALE = 1.0 / 16777216.0;    // (2^10 A1SCALE YSCALE A2SCALE)^-2 = 2^-24: the mel weights assume samples scaled by 2^10; the kernel corrects for its 2^k
constexpr int NKK = 3;             // problems pairs per pass: pass 0 takes kk = 0..2, pass 1 kk = 3..4 (+ one empty slot)
constexpr size_t A1_PASS_BYTES = (size_t)NN2 * 2 * 1024;
constexpr size_t A1_BYTES = 2 * A1_PASS_BYTES;
constexpr size_t MELW_BYTES = (size_t)NPROB * 2 * 3 * 2 * 1024;

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct UttArgs {
  const float* wav; int B; int n; int T;
  const long long* starts; long long n_total;
  const _Float16* a1; const _Float16* a2; const __bf16* melw;
  int n_mels; int pad_mode; int log_mode; float log_eps; float top_db; int use_floor; int mean_norm;
  float* out; int ld_out;
  unsigned inv_mels;
};

#ifdef SD_STAMP
// diagnostic build only (build_native.py --variant stamp "-DSD_STAMP"): cycle counters per phase, waves of the first 1024 workgroups,
// read by tools/stamp_fbank_utt.py; never in a product or timed build
__device__ unsigned long long g_utt_stamps[1024][4][16];
#define UTT_STAMP_FIRST 4096       // (not the first blocks: every CU starts those at the same moment)
#define UTT_STAMP(i) do { if (blockIdx.x >= UTT_STAMP_FIRST && blockIdx.x < UTT_STAMP_FIRST + 1024 && lane == 0) g_utt_stamps[blockIdx.x - UTT_STAMP_FIRST][wid][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define UTT_STAMP(i) do { } while (0)
#endif

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {       // f(integral_constant<int, I>) for I = I .. N - 1: indices are compile-time constants
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__host__ __device__ constexpr int img_words(int n) { return (n + NFFT) + (n + NFFT) / HOP + 1; }

// accumulator register r of lane half h <-> row of a 32x32 MFMA tile
__host__ __device__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {      // upper 32 lanes of a <-> lower 32 lanes of b
  const auto w = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  const unsigned w0 = w[0], w1 = w[1];
  a = w0; b = w1;
}

__device__ __forceinline__ unsigned pk_rtz(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
}
__device__ __forceinline__ float f16lo(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xFFFFu)); }
__device__ __forceinline__ float f16hi(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }

__device__ __forceinline__ h8v frag_of(unsigned a, unsigned b, unsigned c, unsigned d) {
  const u32x4 v = {a, b, c, d};
  return __builtin_bit_cast(h8v, v);
}

// one stage-2 problem (k1) of a tile: 25-point DFT of Y[k1, .] -> power spectrum -> mel accumulators
// bh / bl: the split Y values as B fragments of the four k steps (register g of pair slot 4 s + i)
__device__ __forceinline__ void utt_problem(const char* a2l, const unsigned (&uh)[16], const unsigned (&ul)[16],
                                            f32x16 (&mel)[3], bf8v (&wcur)[2][6]) {
  f32x16 c2[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) c2[u][r] = 0.f;
  // the stage-2 matrix fragments come from LDS (a2l = fragment 0 + 16 lane), step (s, u)'s pair requested in front of the previous step's MFMAs
  h8v ah = *reinterpret_cast<const h8v*>(a2l), al = *reinterpret_cast<const h8v*>(a2l + 1024);
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int s = st >> 1, u = st & 1;
    const h8v bh = frag_of(uh[4 * s], uh[4 * s + 1], uh[4 * s + 2], uh[4 * s + 3]);
    const h8v bl = frag_of(ul[4 * s], ul[4 * s + 1], ul[4 * s + 2], ul[4 * s + 3]);
    h8v nh = ah, nl = al;
    if (st + 1 < 8) {
      const int s1 = (st + 1) >> 1, u1 = (st + 1) & 1;
      nh = *reinterpret_cast<const h8v*>(a2l + ((u1 * 4 + s1) * 2 + 0) * 1024);
      nl = *reinterpret_cast<const h8v*>(a2l + ((u1 * 4 + s1) * 2 + 1) * 1024);
    }
    c2[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c2[u], 0, 0, 0);
    c2[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c2[u], 0, 0, 0);
    c2[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c2[u], 0, 0, 0);
    ah = nh; al = nl;
  }
  // |X|^2 of the 16 outputs this lane holds per row tile (registers 2 q, 2 q + 1 = re, im of output q + 8 h + 16 u): they are the
  // mel product's B fragment as they lie (k = 8 h + q), split into two bf16 halves
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    bf8v p1, p2;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float pw = c2[u][2 * q] * c2[u][2 * q] + c2[u][2 * q + 1] * c2[u][2 * q + 1];
      p1[q] = (__bf16)pw;
      p2[q] = (__bf16)(pw - (float)p1[q]);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      mel[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wcur[u][2 * t], p1, mel[t], 0, 0, 0);
      mel[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wcur[u][2 * t], p2, mel[t], 0, 0, 0);
      mel[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wcur[u][2 * t + 1], p1, mel[t], 0, 0, 0);
    }
  }
}

// Table fragments (1 KB: 16 bytes per lane) are read as uniform base + 32-bit lane offset: the base (+ the fragment's compile-time
// offset) stays in scalar registers.  With a 64-bit per-lane pointer per fragment hipcc computed all ~160 addresses in front of the
// tile loop and spilled them to scratch memory.
// (Explicitly global: behind the per-round asm barrier the optimiser no longer knows the address space and would emit flat loads.)
typedef const __attribute__((address_space(1))) char* gptr_t;
template <typename V>
__device__ __forceinline__ V utt_frag(gptr_t base, unsigned lane16, int frag) {
  return *reinterpret_cast<const __attribute__((address_space(1))) V*>(base + (size_t)frag * 1024 + lane16);
}

__device__ __forceinline__ void utt_wload(gptr_t melw, unsigned lane16, int k1, bf8v (&w)[2][6]) {
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int i = 0; i < 6; ++i) w[u][i] = utt_frag<bf8v>(melw, lane16, (k1 * 2 + u) * 6 + i);
}

// ---- stage 1 of one tile for the problem pairs of ONE pass (its matrices put the pass's rows into accumulator registers 0..5:
// register 2 i + part = re / im of kk = 3 pass + i, k1 = kk in lane half 0 and kk + 5 in half 1): for each n2 one 32x32x16 product
// (rows: k1 x re/im, k: n1, columns: frames) as three MFMAs on split operands; the sums are split into two f16 halves (re, im packed) as
// they come out.  Both passes run the SAME code on different tables (the unrolled body is 9 KB; the whole kernel has to stay inside the
// 64 KB instruction cache, see the kernel).  x0: word of x[200 h] of the lane's frame; x[200 h + m] lies at x0 + m + [m >= (h ? 120 : 160)]
__device__ __forceinline__ void utt_stage1(const unsigned* x0, int h, gptr_t a1b, unsigned lane16, unsigned (&yh)[NN2 + 1][NKK], unsigned (&yl)[NN2 + 1][NKK]) {
  const unsigned* const xm = x0 + h;
  // n2 in pairs (2 m, 2 m + 1; 25 is padding): the 16 signal words of pair m + 1 and the 4 matrix fragments of pair m + 2 are requested
  // in front of pair m's six MFMAs (one wave per SIMD: nothing else hides an LDS or L2 round trip)
  constexpr int NP = (NN2 + 1) / 2;
  auto xword = [&](int e, int n2) -> unsigned {
    const int m = 25 * e + n2;                    // n = 200 h + m
    return m < 120 ? x0[m] : (m < 160 ? xm[m] : x0[m + 1]);
  };
  unsigned d[2][2][8];                            // [buffer][n2 parity][e]
  h8v af[3][4];                                   // [ring slot][hi(2m), lo(2m), hi(2m+1), lo(2m+1)]
  auto xload = [&](int buf, int m) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      d[buf][0][e] = xword(e, 2 * m);
      d[buf][1][e] = 2 * m + 1 < NN2 ? xword(e, 2 * m + 1) : 0u;
    }
  };
  auto aload = [&](int slot, int m) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int frag = 4 * m + i;
      af[slot][i] = utt_frag<h8v>(a1b, lane16, frag < 2 * NN2 ? frag : 2 * NN2 - 1);
    }
  };
  aload(0, 0);
  aload(1, 1);
  xload(0, 0);
  f32x16 c1[2][2];                                // [pair parity][n2 parity]
  static_for<0, NP + 1>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    if constexpr (m < NP) {
      if constexpr (m + 2 < NP) aload((m + 2) % 3, m + 2);
      if constexpr (m + 1 < NP) xload((m + 1) & 1, m + 1);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (2 * m + q < NN2) {
          const unsigned* dd = d[m & 1][q];
          const h8v xh = frag_of(__builtin_amdgcn_perm(dd[1], dd[0], 0x05040100u), __builtin_amdgcn_perm(dd[3], dd[2], 0x05040100u),
                                 __builtin_amdgcn_perm(dd[5], dd[4], 0x05040100u), __builtin_amdgcn_perm(dd[7], dd[6], 0x05040100u));
          const h8v xl = frag_of(__builtin_amdgcn_perm(dd[1], dd[0], 0x07060302u), __builtin_amdgcn_perm(dd[3], dd[2], 0x07060302u),
                                 __builtin_amdgcn_perm(dd[5], dd[4], 0x07060302u), __builtin_amdgcn_perm(dd[7], dd[6], 0x07060302u));
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.f;
          z = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[m % 3][2 * q], xh, z, 0, 0, 0);
          z = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[m % 3][2 * q], xl, z, 0, 0, 0);
          c1[m & 1][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[m % 3][2 * q + 1], xh, z, 0, 0, 0);
        }
      }
    }
    if constexpr (m > 0) {                        // split the previous pair's sums while this pair's MFMAs run
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int n2 = 2 * (m - 1) + q;
        if (n2 < NN2) {
          const f32x16& c = c1[(m - 1) & 1][q];
#pragma unroll
          for (int i = 0; i < NKK; ++i) {
            const float re = YSCALE * c[2 * i], im = YSCALE * c[2 * i + 1];
            const unsigned hi = pk_rtz(re, im);
            yh[n2][i] = hi;
            yl[n2][i] = pk_rtz(re - f16lo(hi), im - f16hi(hi));
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);            // keep the pairs in this order: requests, MFMAs, the previous pair's split
  });
#pragma unroll
  for (int i = 0; i < NKK; ++i) { yh[NN2][i] = 0u; yl[NN2][i] = 0u; }     // n2 = 25: padding of the last pair
}

// ---- stage 2 + mel for the problem pairs (kk, kk + 5), kk = 3 pass + i.  Lane half 0 holds k1 = kk, half 1 holds k1 = kk + 5 of every n2;
// trading the upper lanes of n2 = 2 g with the lower lanes of n2 = 2 g + 1 (in place) gives two registers in which BOTH halves belong to one
// problem, half 0 with n2 = 2 g and half 1 with n2 = 2 g + 1: element pair g & 3 of k step g >> 2 of that problem's B fragment.
// `pass` is a run-time value (wave-uniform): kk = 5 and k1 = 9 do not exist and are skipped.
__device__ __forceinline__ void utt_pairs(gptr_t melw, unsigned lane16, const char* a2l, int pass, bool active, unsigned (&yh)[NN2 + 1][NKK],
                                          unsigned (&yl)[NN2 + 1][NKK], f32x16 (&mel)[3]) {
  bf8v wa[2][6], wb[2][6];
  const gptr_t wbase = melw + (size_t)(3 * pass) * 12 * 1024;      // problem k1's 12 fragments at melw + k1 * 12 KB
  if (active) utt_wload(wbase, lane16, 0, wa);
  static_for<0, NKK>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const int kk = 3 * pass + i;
    if (kk < 5 && active) {
      unsigned uh[16], ul[16], vh[16], vl[16];
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        if (g < 13) {
          unsigned a = yh[2 * g][i], bb = yh[2 * g + 1][i];
          swap32(a, bb);
          uh[g] = a; vh[g] = bb;
          unsigned c = yl[2 * g][i], dd = yl[2 * g + 1][i];
          swap32(c, dd);
          ul[g] = c; vl[g] = dd;
        } else {
          uh[g] = 0u; vh[g] = 0u; ul[g] = 0u; vl[g] = 0u;
        }
      }
      // the mel weights of the next problem are requested one problem ahead (k1 = 9 does not exist: its slot re-reads k1 = 8)
      utt_wload(wbase, lane16, kk < 4 ? i + 5 : i + 4, wb);
      utt_problem(a2l, uh, ul, mel, wa);                // k1 = kk
      if constexpr (i + 1 < NKK) utt_wload(wbase, lane16, kk < 4 ? i + 1 : i, wa);
      if (kk < 4) utt_problem(a2l, vh, vl, mel, wb);    // k1 = kk + 5
    }
  });
}

__global__ __launch_bounds__(256, 1) void fbank_utt_kernel(const UttArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  char* const a2s = smem_raw;                                                        // stage-2 matrix fragments (16 KB)
  float* const scratch = reinterpret_cast<float*>(smem_raw + A2_BYTES);
  unsigned* const img = reinterpret_cast<unsigned*>(smem_raw + A2_BYTES + SCRATCH_FLOATS * 4);    // split samples; later the log-mel rows
  float* const lm = reinterpret_cast<float*>(img);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.x;

  int kexp = 10;
  UTT_STAMP(0);
  // ---- stage-2 matrix -> LDS; wave maxima reset
#pragma unroll
  for (int i = 0; i < A2_BYTES / (256 * 16); ++i)
    *reinterpret_cast<u32x4*>(a2s + (i * 256 + tid) * 16) = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.a2) + (i * 256 + tid) * 16);
  if (tid < 8) scratch[tid] = -INFINITY;

  // ---- the padded signal (sample i - 200 of the utterance at extended index i) -> LDS: clamped to +-16, scaled by the power of two
  // 2^k that puts the utterance's peak into [2^13, 2^14) (exact; |x| <= 16 gives k >= 9, a silent utterance takes 2^10), split into
  // two f16 halves.  With the scale chosen per utterance the low halves stay clear of the f16 subnormals at EVERY signal level
  // (a fixed 2^10 lost bits below ~1e-4 of full scale); the mel energies take 2^(2 (10 - k)) back before the log.
  // A thread owns groups of four consecutive indices that lie inside the utterance and the signal (16-byte loads, all of them in
  // flight at once: ~128 KB per workgroup, the CU has nothing else to run) and keeps them in registers across the peak reduction,
  // so the image is written once.  Everything else -- the padding at both ends, and whatever hangs over an end of the signal when a
  // window does -- goes element by element through a small loop (raw value to LDS, scaled in place after the reduction).
  const int L = p.n + NFFT;
  {
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // (a window may start at any sample)
    const long long start = p.starts ? p.starts[b] : (long long)b * p.n;
    auto group_inner = [&](int g) -> bool {
      const int i = 4 * g;                                                       // extended indices i .. i + 3 = samples i - 200 ..
      const long long gi = start + i - NFFT / 2;
      return i >= NFFT / 2 && i + 3 < NFFT / 2 + p.n && gi >= 0 && gi + 3 < p.n_total;
    };
    f32x4 v[UTT_MAX_GROUPS];
    float amax = 0.f;
#pragma unroll
    for (int u = 0; u < UTT_MAX_GROUPS; ++u) {
      const int g = tid + 256 * u;
      const bool in = group_inner(g);
      const f32x4 x = *reinterpret_cast<const f32x4u*>(p.wav + (in ? start + 4 * g - NFFT / 2 : 0));
#pragma unroll
      for (int c = 0; c < 4; ++c) v[u][c] = in ? x[c] : 0.f;
    }
    // element by element: i in [0, 204) and [196 + n, L) -- or all of [0, L) when the window hangs over an end of the signal
    const bool hang = start < 0 || start + p.n > p.n_total;
    auto edge_range = [&](int lo, int hi, auto&& fn) {
      for (int i = lo + tid; i < hi; i += 256)
        if (!group_inner(i >> 2)) fn(i);
    };
    auto edges = [&](auto&& fn) {
      if (hang) { edge_range(0, L, fn); }
      else { edge_range(0, NFFT / 2 + 4 < L ? NFFT / 2 + 4 : L, fn); edge_range(NFFT / 2 + p.n - 4 > NFFT / 2 + 4 ? NFFT / 2 + p.n - 4 : NFFT / 2 + 4, L, fn); }
    };
    edges([&](int i) {
      int sidx = i - NFFT / 2;
      bool ok = true;
      if (p.pad_mode == SD_PAD_REFLECT) {
        sidx = sidx < 0 ? -sidx : sidx;
        sidx = sidx >= p.n ? 2 * (p.n - 1) - sidx : sidx;
      } else {
        ok = sidx >= 0 && sidx < p.n;
      }
      sidx = sidx < 0 ? 0 : (sidx >= p.n ? p.n - 1 : sidx);
      long long gi = start + sidx;                    // a window may hang over either end of the signal: zeros there
      ok = ok && gi >= 0 && gi < p.n_total;
      gi = gi < 0 ? 0 : (gi >= p.n_total ? p.n_total - 1 : gi);
      float x = p.wav[gi];
      x = ok ? __builtin_amdgcn_fmed3f(x, -XMAX, XMAX) : 0.f;
      amax = fmaxf(amax, fabsf(x));
      img[i + i / HOP] = __float_as_uint(x);
    });
#pragma unroll
    for (int u = 0; u < UTT_MAX_GROUPS; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        v[u][c] = __builtin_amdgcn_fmed3f(v[u][c], -XMAX, XMAX);
        amax = fmaxf(amax, fabsf(v[u][c]));                                      // (groups that were not loaded hold zeros)
      }
    amax = sd_wave_max(amax);
    if (lane == 0) scratch[8 + wid] = amax;
    UTT_STAMP(1);
    __syncthreads();
    const float mx = fmaxf(fmaxf(scratch[8], scratch[9]), fmaxf(scratch[10], scratch[11]));
    const int e = (int)((__float_as_uint(mx) >> 23) & 0xFFu) - 126;       // mx = m 2^e, m in [0.5, 1) (normal numbers)
    if (mx >= 1e-30f) kexp = 14 - e;                                       // mx 2^k in [2^13, 2^14); k <= 14 + 99
    const float xs = __uint_as_float((unsigned)(kexp + 127) << 23);
    auto split = [&](float x) -> unsigned {
      const _Float16 hi = (_Float16)x;
      const _Float16 lo = (_Float16)(x - (float)hi);
      return (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
    };
#pragma unroll
    for (int u = 0; u < UTT_MAX_GROUPS; ++u) {
      const int g = tid + 256 * u;
      if (group_inner(g)) {
        unsigned* const dst = img + 4 * g + (4 * g) / HOP;               // (160 is a multiple of 4: a group never straddles a skew step)
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[c] = split(xs * v[u][c]);
      }
    }
    edges([&](int i) {
      unsigned* const w = img + i + i / HOP;
      *w = split(xs * __uint_as_float(*w));
    });
  }
  const float pback = __uint_as_float((unsigned)(10 - kexp + 127) << 23);    // 2^(10 - k), applied twice to a mel energy
  UTT_STAMP(2);
  __syncthreads();
  UTT_STAMP(3);

  const int ntiles = (p.T + FT - 1) / FT;
  const int j = lane & 31, h = lane >> 5;
  const unsigned lane16 = (unsigned)lane * 16u;
#pragma unroll 1
  for (int round = 0; round < 2; ++round) {
    // (opaque to the optimiser per round: keeps the ~160 fragment addresses base + constant from being computed in front of the loop)
    gptr_t a1b = (gptr_t)reinterpret_cast<const char*>(p.a1);
    gptr_t melw = (gptr_t)reinterpret_cast<const char*>(p.melw);
    asm volatile("" : "+s"(a1b), "+s"(melw));
    const int tile = wid + 4 * round;
    const bool active = tile < ntiles;                // wave-uniform
    const int f0 = tile * FT;
    int nvalid = p.T - f0; nvalid = nvalid > FT ? FT : nvalid;
    const int f = f0 + (j < nvalid ? j : 0);
    const unsigned* const x0 = img + 161 * f + 201 * h;      // word of x[200 h] of this lane's frame
    f32x16 mel[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mel[t][r] = 0.f;
    // The nine problems run in two passes over stage 1 (k1 = kk + 5 h: kk = 0..2, then kk = 3..4): the split stage-1 sums of ONE
    // pass (150 registers) fit the 256 architectural VGPRs beside the fragments, those of all five kk (250) do not.  Price: stage 1's
    // 75 MFMAs twice.  The two passes are ONE piece of code (a run-time loop; the pass selects its tables): the unrolled body is
    // ~20 KB, and with the two passes as separate code (and a fully unrolled staging phase) the kernel was 91 KB -- more than the 64 KB
    // instruction cache a CU pair shares, so that every utterance streamed its instructions from L2 and the first tile round of an
    // utterance ran at half the speed of the second (in-kernel stamps).
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      unsigned yh[NN2 + 1][NKK], yl[NN2 + 1][NKK];
      if (active) utt_stage1(x0, h, a1b + (size_t)pass * A1_PASS_BYTES, lane16, yh, yl);
      UTT_STAMP(4 + 5 * round + 2 * pass);
      // the log-mel rows of the first four tiles overwrite signal words that only those tiles' stage 1 reads
      if (round == 0 && pass == 1) __syncthreads();
      utt_pairs(melw, lane16, a2s + lane16, pass, active, yh, yl, mel);
      UTT_STAMP(5 + 5 * round + 2 * pass);
    }
    if (active) {
      // ---- log, utterance maximum, rows -> LDS
      const bool ln = p.log_mode == SD_LOG_LN_EPS;
      const float lscale = ln ? 0.6931471805599453f : 3.0102999566398120f;      // v_log_f32 is log2
      float vmax = -INFINITY;
      // (no branches: values of frames / mel rows that do not exist go to a spare word and count as -inf)
      float* const mrow = lm + (size_t)(f0 + j) * MELP;
      float* const spare = scratch + SCRATCH_FLOATS - 64 + lane;
      const bool jok = j < nvalid;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = 32 * t + acc_row(r, 0) + 4 * h;
          const bool ok = jok && m < p.n_mels;
          const float v = mel[t][r] * pback * pback;
          const float lv = lscale * __builtin_amdgcn_logf(ln ? v + p.log_eps : fmaxf(v, p.log_eps));
          *(ok ? mrow + m : spare) = lv;
          vmax = fmaxf(vmax, ok ? lv : -INFINITY);
        }
      vmax = sd_wave_max(vmax);
      if (lane == 0) scratch[round * 4 + wid] = vmax;
      UTT_STAMP(8 + 5 * round);
    }
  }
  __syncthreads();
  UTT_STAMP(14);

  // ---- top_db floor relative to the utterance maximum, mean over T per mel bin, the one write of the output
  float thr = -INFINITY;
  if (p.use_floor) {
    float mx = scratch[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) mx = fmaxf(mx, scratch[i]);
    thr = mx - p.top_db;
  }
  float* const mean = scratch + 16;
  float* const part = scratch + 128;
  if (p.mean_norm) {
    // column sums: thread = (row group, 4 consecutive mels), row groups stride RG; two rows per step keep two chains of LDS reads in flight
    const int per_row = (p.n_mels + 3) >> 2;
    int RG = 256 / per_row; RG = RG > 12 ? 12 : RG;
    const int rg = tid / per_row, c0 = (tid - rg * per_row) * 4;
    if (rg < RG) {
      float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
      int t = rg;
      for (; t + RG < p.T; t += 2 * RG) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c0 + c < p.n_mels) { s0[c] += fmaxf(lm[t * MELP + c0 + c], thr); s1[c] += fmaxf(lm[(t + RG) * MELP + c0 + c], thr); }
      }
      if (t < p.T) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c0 + c < p.n_mels) s0[c] += fmaxf(lm[t * MELP + c0 + c], thr);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c0 + c < p.n_mels) part[rg * 96 + c0 + c] = s0[c] + s1[c];
    }
    __syncthreads();
    if (tid < p.n_mels) {
      float sum = 0.f;
      for (int k = 0; k < RG; ++k) sum += part[k * 96 + tid];
      mean[tid] = sum / (float)p.T;
    }
    __syncthreads();
  }
  float* const orow = p.out + (size_t)b * p.T * p.ld_out;
  if ((p.n_mels & 3) == 0 && (p.ld_out & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15u) == 0) {
    const int per_row = p.n_mels >> 2;
    const int total = p.T * per_row;
    const unsigned inv = p.inv_mels;                  // ceil(2^32 / (n_mels / 4))
#pragma unroll 4
    for (int e = tid; e < total; e += 256) {
      const int t = (int)__umulhi((unsigned)e, inv);
      const int c = (e - t * per_row) * 4;
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaxf(lm[t * MELP + c + i], thr) - (p.mean_norm ? mean[c + i] : 0.f);
      *reinterpret_cast<f32x4*>(orow + (size_t)t * p.ld_out + c) = v;
    }
  } else {
    const int total = p.T * p.n_mels;
    for (int e = tid; e < total; e += 256) {
      const int t = e / p.n_mels, c = e - t * p.n_mels;
      orow[(size_t)t * p.ld_out + c] = fmaxf(lm[t * MELP + c], thr) - (p.mean_norm ? mean[c] : 0.f);
    }
  }
  UTT_STAMP(15);
}

size_t utt_lds_bytes(int n, int T) {
  const size_t words = (size_t)img_words(n) > (size_t)T * MELP ? (size_t)img_words(n) : (size_t)T * MELP;
  return (size_t)A2_BYTES + SCRATCH_FLOATS * 4 + words * 4;
}

unsigned short bf16_bits(float v) {        // round to nearest even (finite inputs)
  unsigned u;
  std::memcpy(&u, &v, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
float bf16_value(unsigned short b) {
  const unsigned u = (unsigned)b << 16;
  float v;
  std::memcpy(&v, &u, 4);
  return v;
}

}  // namespace

int sd_fbank_utt_create_tables(sd_fbank_plan* plan, const float* window, const float* mel_fb) {
  const int n_mels = plan->n_mels;
  // stage 1, one table per pass: A operand of v_mfma_f32_32x32x16_f16: lane l holds row l & 31, k = n1 = 8 (l >> 5) + e.  Row rho is
  // accumulator register r = (rho & 3) + 4 (rho >> 3) of lane half (rho >> 2) & 1: r = 2 i + part, kk = 3 pass + i, k1 = kk + 5 half
  // (rows r >= 6, kk = 5 and k1 = 9: zero)
  std::vector<_Float16> a1(A1_BYTES / 2);
  {
    size_t o = 0;
    for (int pass = 0; pass < 2; ++pass)
      for (int n2 = 0; n2 < NN2; ++n2)
        for (int part = 0; part < 2; ++part)
          for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e, ++o) {
              const int rho = l & 31, hh = (rho >> 2) & 1, r = (rho & 3) + 4 * (rho >> 3);
              const int kk = 3 * pass + (r >> 1), im = r & 1, k1 = kk + 5 * hh;
              const int n = 25 * (8 * (l >> 5) + e) + n2;
              double v = 0.0;
              if (r < 2 * NKK && kk < 5 && k1 < NPROB) {
                const double ang = 2.0 * M_PI * (double)(((long)k1 * n) % NFFT) / (double)NFFT;
                v = A1SCALE * (double)window[n] * (im ? -std::sin(ang) : std::cos(ang));
              }
              const _Float16 hi = (_Float16)(float)v;
              a1[o] = part ? (_Float16)(float)(v - (double)(float)hi) : hi;
            }
  }
  // stage 2: rows (output idx = q + 8 half + 16 u, re / im), k = 16 s + 8 (l >> 5) + e: pair slot g = 4 s + (e >> 1), n2 = 2 g + (l >> 5),
  // (e & 1) = re / im of Y
  std::vector<_Float16> a2(A2_BYTES / 2);
  {
    size_t o = 0;
    for (int u = 0; u < 2; ++u)
      for (int s = 0; s < 4; ++s)
        for (int part = 0; part < 2; ++part)
          for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e, ++o) {
              const int rho = l & 31, hh = (rho >> 2) & 1, r = (rho & 3) + 4 * (rho >> 3);
              const int idx = (r >> 1) + 8 * hh + 16 * u, oim = r & 1;
              const int n2 = 2 * (4 * s + (e >> 1)) + (l >> 5), yim = e & 1;
              double v = 0.0;
              if (idx < NN2 && n2 < NN2) {
                const double th = 2.0 * M_PI * (double)((n2 * idx) % NN2) / (double)NN2;
                // (c - i s)(yr + i yi): re = c yr + s yi, im = c yi - s yr
                v = A2SCALE * (oim == 0 ? (yim == 0 ? std::cos(th) : std::sin(th)) : (yim == 0 ? -std::sin(th) : std::cos(th)));
              }
              const _Float16 hi = (_Float16)(float)v;
              a2[o] = part ? (_Float16)(float)(v - (double)(float)hi) : hi;
            }
  }
  // mel weights against the power fragments: lane l holds row (mel) 32 t + (l & 31), k = 8 (l >> 5) + e <-> output idx = k + 16 u of
  // problem k1: bin k1 + 16 idx for idx <= 12, bin 16 - k1 + 16 (24 - idx) for 13 <= idx <= 24 and 1 <= k1 <= 7 (k1 = 0, 8: duplicates)
  std::vector<unsigned short> mw(MELW_BYTES / 2);
  {
    size_t o = 0;
    for (int k1 = 0; k1 < NPROB; ++k1)
      for (int u = 0; u < 2; ++u)
        for (int t = 0; t < 3; ++t)
          for (int part = 0; part < 2; ++part)
            for (int l = 0; l < 64; ++l)
              for (int e = 0; e < 8; ++e, ++o) {
                const int m = 32 * t + (l & 31), idx = e + 8 * (l >> 5) + 16 * u;
                int bin = -1;
                if (idx <= 12) bin = k1 + 16 * idx;
                else if (idx <= 24 && k1 >= 1 && k1 <= 7) bin = 16 - k1 + 16 * (24 - idx);
                float w = 0.f;
                if (bin >= 0 && bin < NFREQ && m < n_mels) w = (float)((double)mel_fb[(size_t)bin * n_mels + m] * PSCALE);
                const unsigned short w1 = bf16_bits(w);
                mw[o] = part == 0 ? w1 : bf16_bits(w - bf16_value(w1));
              }
  }
  plan->utt_a1_dev = plan->utt_a2_dev = plan->utt_melw_dev = nullptr;
  hipError_t e = hipMalloc(&plan->utt_a1_dev, A1_BYTES);
  if (e == hipSuccess) e = hipMalloc(&plan->utt_a2_dev, A2_BYTES);
  if (e == hipSuccess) e = hipMalloc(&plan->utt_melw_dev, MELW_BYTES);
  if (e == hipSuccess) e = hipMemcpy(plan->utt_a1_dev, a1.data(), A1_BYTES, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(plan->utt_a2_dev, a2.data(), A2_BYTES, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(plan->utt_melw_dev, mw.data(), MELW_BYTES, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    sd_fbank_utt_destroy_tables(plan);
    return sd_set_error(SD_ERR_HIP, "sd_fbank_plan_create: device table upload failed: %s", hipGetErrorString(e));
  }
  return SD_OK;
}

void sd_fbank_utt_destroy_tables(sd_fbank_plan* plan) {
  if (plan->utt_a1_dev) (void)hipFree(plan->utt_a1_dev);
  if (plan->utt_a2_dev) (void)hipFree(plan->utt_a2_dev);
  if (plan->utt_melw_dev) (void)hipFree(plan->utt_melw_dev);
  plan->utt_a1_dev = plan->utt_a2_dev = plan->utt_melw_dev = nullptr;
}

#ifdef SD_STAMP
extern "C" int sd_debug_read_utt_stamps(unsigned long long* host, int n) {
  SD_CHECK_ARG(host && n == 1024 * 4 * 16, "sd_debug_read_utt_stamps: need 65536 entries");
  SD_CHECK_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_utt_stamps), sizeof(unsigned long long) * n));
  return SD_OK;
}
#endif

bool sd_fbank_utt_supported(const sd_fbank_plan* plan, int n) {
  static const bool on = [] { const char* e = sd_experiment_env("SD_FBANK_UTT"); return !(e && atoi(e) == 0); }();   // A/B switch: 0 = folded kernel, 32 = this one, 16 (default) = sd_fbank_utt16.hip
  if (!on || !plan->utt_a1_dev) return false;
  const int T = 1 + n / HOP;
  return T <= UTT_MAX_T && utt_lds_bytes(n, T) <= (size_t)LDS_LIMIT;
}

int sd_fbank_utt_launch(const sd_fbank_plan* plan, const float* wav_dev, long long n_total, const long long* starts_dev, int B, int n,
                        int mean_norm, float* out_dev, int ld_out, hipStream_t stream) {
  const int T = 1 + n / HOP;
  UttArgs a;
  a.wav = wav_dev; a.B = B; a.n = n; a.T = T;
  a.starts = starts_dev; a.n_total = n_total;
  a.a1 = static_cast<const _Float16*>(plan->utt_a1_dev); a.a2 = static_cast<const _Float16*>(plan->utt_a2_dev);
  a.melw = static_cast<const __bf16*>(plan->utt_melw_dev);
  a.n_mels = plan->n_mels; a.pad_mode = plan->pad_mode; a.log_mode = plan->log_mode; a.log_eps = plan->log_eps;
  a.top_db = plan->top_db;
  a.use_floor = plan->log_mode == SD_LOG_DB_TOPDB && plan->top_db >= 0.f;
  a.mean_norm = mean_norm;
  a.out = out_dev; a.ld_out = ld_out;
  const int per_row = plan->n_mels % 4 == 0 ? plan->n_mels / 4 : plan->n_mels;
  a.inv_mels = (unsigned)((((unsigned long long)1 << 32) + per_row - 1) / per_row);
  const size_t lds = utt_lds_bytes(n, T);
  SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(fbank_utt_kernel), LDS_LIMIT));
  {
    // algorithmic bytes: waveform read once + log-mel written once (192 320 B per 2 s segment)
    SdProfScope prof(SD_PROF_FBANK, stream, (double)B * ((double)n * 4.0 + (double)T * plan->n_mels * 4.0));
    hipLaunchKernelGGL(fbank_utt_kernel, dim3((unsigned)B), dim3(256), lds, stream, a);
  }
  SD_CHECK_LAUNCH("fbank_utt_kernel");
  return SD_OK;
}
