// Log-mel filterbank in ONE launch: one workgroup per utterance, factored DFT on the f16 matrix cores, floor + mean removal in LDS.
//
// Same arithmetic contract as sd_fbank.hip (torchaudio MelSpectrogram + log + mean removal [REF speech_encode.py:17-36], or the
// speechbrain Fbank + sentence mean-norm front end of EncoderClassifier.encode_batch [REF speech_encode.py:77]; SURVEY.md Appendix
// A.1 / A.2); what changes is the shape of the work.  The folded-DFT kernel there streams a 364 KB basis through LDS for every 128
// frames (one L2 -> LDS round trip and one barrier per k step: matrix pipe 25 % busy) and needs a second launch for the
// utterance-level top_db floor and the mean over T, which re-reads and re-writes the whole output.  Here:
//
//  * One 512-thread workgroup owns one utterance of up to 32 100 samples (201 frames: the padded signal must fit the CU's 160 KB of LDS
//    beside the resident matrix and the table ring; longer utterances run on sd_fbank.hip's two launches).  Its padded signal (n + 400 samples) is staged ONCE into LDS,
//    clamped to +-16, scaled by the power of two 2^k that puts the UTTERANCE's peak into [2^13, 2^14) and split into two f16 halves per
//    sample (hi = f16(v), lo = f16(v - hi): one dword), at word i + i / 160 (frame stride 161 words: the 16 frames of a wave tile fall on
//    16 distinct banks).  The log-mel rows of the whole utterance are collected in LDS too (over the dead signal), so the utterance
//    maximum, the top_db floor and the mean over T are applied before the ONLY write of the output: algorithmic traffic (128 000 B read +
//    64 320 B written per 2 s segment), no atomics, no second pass.
//  * 400 = 16 x 25, n = 25 n1 + n2.  Stage 1, per n2: Y[k1, n2] = sum_{n1 < 16} w[n] x[n] e^{-2 pi i k1 n / 400} (window and twiddles folded
//    into one matrix per n2).  x is real, so Y[16 - k1, n2] = e^{-2 pi i n2 / 25} conj Y[k1, n2]: k1 = 1..7 are complex, Y[0, n2] is real and
//    Y[8, n2] e^{+i pi n2 / 25} =: rho[n2] is real: SIXTEEN real rows per n2 = the 16 rows of one v_mfma_f32_16x16x32_f16.  Its K = 32 carries
//    the HIGH halves of the 16 samples in k < 16 and their LOW halves in k >= 16 (lane groups 0, 1 / 2, 3 of the B operand, one v_perm
//    selector per lane), against [A1h | A1h] and [A1l | 0]: two MFMAs = A1h xh + A1h xl + A1l xh, f32-level accuracy (2^-21 per stage).
//    Rows by lane group g = lane >> 4 (a lane holds accumulator rows 4 g .. 4 g + 3 of its frame): g = 0: k1 = 1, 2; g = 1: k1 = 3, 4;
//    g = 2: k1 = 5, 6; g = 3: k1 = 7 and the two reals (Y[0], rho): a "pair" of problems per group, 100 packed registers per lane for
//    the split sums of all 25 n2 -- which is what lets TWO waves share a SIMD (a first version with 32-frame tiles on 32x32x16 MFMAs
//    kept a frame's sums on two lanes, 250 registers: one wave per SIMD, two passes over stage 1, ~6 cycles per instruction, 2.10 ms
//    per 10 000 segments against 1.6 here).
//  * Stage 2, per problem: a 25-point DFT over n2.  Its contraction index n2 lives in separate registers of ONE lane group; a 4 x 4
//    transpose between four registers (n2 = 4 m .. 4 m + 3) and the four lane groups (v_permlane16_swap + v_permlane32_swap, in place)
//    turns that into "register = pair, lane group g = n2 mod 4", i.e. the B fragment of a K = 32 step: k = 8 g + 2 m' + part <-> n2 =
//    4 (4 s + m') + g.  Complex problems share one 64 x 64 matrix, LDS resident (outputs idx 0..12 are bins k1 + 16 idx, idx 13..24 the
//    conjugates of bins 16 - k1 + 16 (24 - idx)); the pair of reals has its own (rows 0..12: bins 16 k2, rows 13..25: bins 8 + 16 k2 with
//    the phase e^{-i pi n2 / 25} folded in).  The stage-1 sums are split hi + lo again (scaled by 2^-8 first: < 2^15), three MFMAs per step.
//  * A lane then holds re / im of two outputs per 16-row tile: eight powers per problem = the B fragment of ONE K = 32 step of the mel
//    product (80 mels = five 16-row tiles; power spectrum as split bf16 against split bf16 weights, 2^-16).
//  * The f16 MFMA flushes subnormal operands, so every LOW half that matters must be a normal number (>= 2^-14) and every value < 2^16:
//    hence the per-utterance 2^k (low halves normal down to 100 dB below the utterance's peak, at every signal level), both matrices
//    x 2^5, the 2^-8 on the stage-1 sums; the mel weights and one multiply in front of the log take 2^(-2 (k + 2)) back.
//  * The tables every wave needs in the same order (stage-1 matrices 50 KB, the real pair's matrix 16 KB, mel weights 80 KB per tile
//    round) pass through a 16 KB LDS ring once per round (see the kernel): read per wave from L2 they were the bottleneck.
// Per 16-frame tile: 50 + 192 + 120 MFMAs of 16 cycles.  Measured: DESIGN.md section 4.
#include "sd_fbank_internal.h"
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

namespace {

constexpr int NFFT = 400;
constexpr int HOP = 160;
constexpr int NFREQ = 201;
constexpr int NN2 = 25;
constexpr int NBLK = 7;                 // blocks of four n2 (28 >= 25)
constexpr int FT = 16;                  // frames per wave tile
constexpr int MELP = 81;
constexpr int U16_MAX_T = 208;          // 13 tiles: two rounds of the 8 waves
constexpr int U16_THREADS = 512;
constexpr int U16_MAX_GROUPS = (((U16_MAX_T * HOP - 1) + NFFT + 3) / 4 + U16_THREADS - 1) / U16_THREADS;
constexpr int A2_BYTES = 16 * 1024;
constexpr int RING_BYTES = 16 * 1024;   // two chunks of eight 1 KB fragments: the streamed tables, shared by the eight waves
constexpr int SCRATCH_FLOATS = 64;      // [0..15] wave maxima of the log-mel rows, [16..23] of |x|, [24..31] "this wave saw a NaN sample"
constexpr int LDS_LIMIT = 160 * 1024;
constexpr float XMAX = 16.f;
constexpr double A1SCALE = 32.0, A2SCALE = 32.0;
constexpr float YSCALE = 1.0f / 256.0f;
constexpr double PSCALE = 1.0 / 16777216.0;      // (2^10 * 32 * 2^-8 * 32)^-2
constexpr int NPROB = 8;                // k1 = 1..7 and the pair of reals (k1 = 0, 8)
constexpr int MEL_TILES = 5;
// one device table of 1 KB fragments: [A2: 16, LDS resident] then the STREAM in the order a tile round consumes it, eight fragments
// (= four items of a hi / lo or W1 / W2 pair) per chunk: [A1: 25 n2 x (hi, lo), padded to 56] [A2 of the real pair: 16] [mel: 8 problems x 5
// tiles x (W1, W2)]
constexpr int T_A2 = 0, T_STREAM = 16;
constexpr int S_A1 = 0, S_A2S = 56, S_MEL = 72, S_FRAGS = S_MEL + NPROB * MEL_TILES * 2;      // stream-relative
constexpr int NCHUNK = S_FRAGS / 8;     // 19
static_assert(S_FRAGS % 8 == 0 && S_A2S % 8 == 0 && S_MEL % 8 == 0, "chunks of eight fragments");
constexpr int T_FRAGS = T_STREAM + S_FRAGS;

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) char* gptr_t;

struct U16Args {
  const float* wav; int B; int n; int T;
  const long long* starts; long long n_total;
  const char* tables;
  int n_mels; int pad_mode; int log_mode; float log_eps; float top_db; int use_floor; int mean_norm;
  float* out; int ld_out;
  unsigned inv_mels;
};

#ifdef SD_STAMP
__device__ unsigned long long g_utt16_stamps[1024][8][16];
#define U16_STAMP(i) do { if (blockIdx.x >= 4096 && blockIdx.x < 5120 && lane == 0) g_utt16_stamps[blockIdx.x - 4096][wid][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define U16_STAMP(i) do { } while (0)
#endif

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__host__ __device__ constexpr int img_words(int n) { return (n + NFFT) + (n + NFFT) / HOP + 1; }

__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {      // 16-lane rows 1, 3 of a <-> rows 0, 2 of b
  const auto w = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  const unsigned w0 = w[0], w1 = w[1];
  a = w0; b = w1;
}
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {      // rows 2, 3 of a <-> rows 0, 1 of b
  const auto w = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  const unsigned w0 = w[0], w1 = w[1];
  a = w0; b = w1;
}
// r[i] holds in lane group g the value (i, g)  ->  r[i] holds in lane group g the value (g, i)
__device__ __forceinline__ void transpose4(unsigned (&r)[4]) {
  swap16(r[0], r[1]);
  swap16(r[2], r[3]);
  swap32(r[0], r[2]);
  swap32(r[1], r[3]);
}

__device__ __forceinline__ unsigned pk_rtz(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b)); }
__device__ __forceinline__ float f16lo(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xFFFFu)); }
__device__ __forceinline__ float f16hi(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }
__device__ __forceinline__ h8v frag_of(unsigned a, unsigned b, unsigned c, unsigned d) {
  const u32x4 v = {a, b, c, d};
  return __builtin_bit_cast(h8v, v);
}
template <typename V>
__device__ __forceinline__ V gfrag(gptr_t base, unsigned lane16, int frag) {      // uniform base + 32-bit lane offset: the base stays in scalar registers
  return *reinterpret_cast<const __attribute__((address_space(1))) V*>(base + (size_t)frag * 1024 + lane16);
}

// the 8 powers a lane holds of one problem (rows 4 g + r of tile u: r = 2 q + part -> output idx = 2 g + q + 8 u; power (u, q) = element
// 2 u + q of the mel product's B fragment), split into two bf16 halves (packed: 4 + 4 registers)
__device__ __forceinline__ void u16_powers(const f32x4 (&c2)[4], unsigned (&p1)[4], unsigned (&p2)[4]) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float pa = c2[u][0] * c2[u][0] + c2[u][1] * c2[u][1];
    const float pb = c2[u][2] * c2[u][2] + c2[u][3] * c2[u][3];
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 h;
    h[0] = (__bf16)pa; h[1] = (__bf16)pb;
    bf2 l;
    l[0] = (__bf16)(pa - (float)h[0]); l[1] = (__bf16)(pb - (float)h[1]);
    p1[u] = __builtin_bit_cast(unsigned, h);
    p2[u] = __builtin_bit_cast(unsigned, l);
  }
}

// steps [ST0, ST1) of one stage-2 problem: C2[u] (16 outputs x 16 frames each, u = 0..3) += A2 (64 x 64; the fragment pair of step
// (s, u) = (st >> 2, st & 3) at mat + (u * 2 + s) * 2 KB resp. + item(st) * 2 KB, hi then lo) times the problem's split sums
template <int ST0, int ST1, typename AddrOf>
__device__ __forceinline__ void u16_steps(AddrOf&& addr_of, const unsigned (&yh)[NBLK], const unsigned (&yl)[NBLK], f32x4 (&c2)[4]) {
  static_assert((ST1 - ST0) % 2 == 0 && ST0 % 2 == 0, "steps go in pairs (u, u + 1) of one k step");
  // two steps at a time, their three-MFMA chains interleaved (a dependent MFMA right behind its predecessor waits for the result);
  // the next pair's four matrix fragments are requested in front of this pair's MFMAs
  h8v ah0 = *reinterpret_cast<const h8v*>(addr_of(ST0)), al0 = *reinterpret_cast<const h8v*>(addr_of(ST0) + 1024);
  h8v ah1 = *reinterpret_cast<const h8v*>(addr_of(ST0 + 1)), al1 = *reinterpret_cast<const h8v*>(addr_of(ST0 + 1) + 1024);
#pragma unroll
  for (int st = ST0; st < ST1; st += 2) {
    const int s = st >> 2, u = st & 3;
    const h8v bh = frag_of(yh[4 * s], yh[4 * s + 1], yh[4 * s + 2], s == 0 ? yh[3] : 0u);
    const h8v bl = frag_of(yl[4 * s], yl[4 * s + 1], yl[4 * s + 2], s == 0 ? yl[3] : 0u);
    h8v nh0 = ah0, nl0 = al0, nh1 = ah1, nl1 = al1;
    if (st + 2 < ST1) {
      nh0 = *reinterpret_cast<const h8v*>(addr_of(st + 2)); nl0 = *reinterpret_cast<const h8v*>(addr_of(st + 2) + 1024);
      nh1 = *reinterpret_cast<const h8v*>(addr_of(st + 3)); nl1 = *reinterpret_cast<const h8v*>(addr_of(st + 3) + 1024);
    }
    c2[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bh, c2[u], 0, 0, 0);
    c2[u + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bh, c2[u + 1], 0, 0, 0);
    c2[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bl, c2[u], 0, 0, 0);
    c2[u + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bl, c2[u + 1], 0, 0, 0);
    c2[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bh, c2[u], 0, 0, 0);
    c2[u + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bh, c2[u + 1], 0, 0, 0);
    ah0 = nh0; al0 = nl0; ah1 = nh1; al1 = nl1;
  }
}

// the same for TWO problems that share the matrix (all complex problems do): each fragment pair is read from LDS once and feeds both
template <typename AddrOf>
__device__ __forceinline__ void u16_steps2(AddrOf&& addr_of, const unsigned (&yhA)[NBLK], const unsigned (&ylA)[NBLK], const unsigned (&yhB)[NBLK],
                                           const unsigned (&ylB)[NBLK], f32x4 (&ca)[4], f32x4 (&cb)[4]) {
  h8v ah0 = *reinterpret_cast<const h8v*>(addr_of(0)), al0 = *reinterpret_cast<const h8v*>(addr_of(0) + 1024);
  h8v ah1 = *reinterpret_cast<const h8v*>(addr_of(1)), al1 = *reinterpret_cast<const h8v*>(addr_of(1) + 1024);
#pragma unroll
  for (int st = 0; st < 8; st += 2) {
    const int s = st >> 2, u = st & 3;
    const h8v bhA = frag_of(yhA[4 * s], yhA[4 * s + 1], yhA[4 * s + 2], s == 0 ? yhA[3] : 0u);
    const h8v blA = frag_of(ylA[4 * s], ylA[4 * s + 1], ylA[4 * s + 2], s == 0 ? ylA[3] : 0u);
    const h8v bhB = frag_of(yhB[4 * s], yhB[4 * s + 1], yhB[4 * s + 2], s == 0 ? yhB[3] : 0u);
    const h8v blB = frag_of(ylB[4 * s], ylB[4 * s + 1], ylB[4 * s + 2], s == 0 ? ylB[3] : 0u);
    h8v nh0 = ah0, nl0 = al0, nh1 = ah1, nl1 = al1;
    if (st + 2 < 8) {
      nh0 = *reinterpret_cast<const h8v*>(addr_of(st + 2)); nl0 = *reinterpret_cast<const h8v*>(addr_of(st + 2) + 1024);
      nh1 = *reinterpret_cast<const h8v*>(addr_of(st + 3)); nl1 = *reinterpret_cast<const h8v*>(addr_of(st + 3) + 1024);
    }
    ca[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bhA, ca[u], 0, 0, 0);
    cb[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bhB, cb[u], 0, 0, 0);
    ca[u + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bhA, ca[u + 1], 0, 0, 0);
    cb[u + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bhB, cb[u + 1], 0, 0, 0);
    ca[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, blA, ca[u], 0, 0, 0);
    cb[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, blB, cb[u], 0, 0, 0);
    ca[u + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, blA, ca[u + 1], 0, 0, 0);
    cb[u + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, blB, cb[u + 1], 0, 0, 0);
    ca[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bhA, ca[u], 0, 0, 0);
    cb[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bhB, cb[u], 0, 0, 0);
    ca[u + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bhA, ca[u + 1], 0, 0, 0);
    cb[u + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bhB, cb[u + 1], 0, 0, 0);
    ah0 = nh0; al0 = nl0; ah1 = nh1; al1 = nl1;
  }
}

__global__ __launch_bounds__(U16_THREADS, 2) void fbank_utt16_kernel(const U16Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  char* const a2s = smem_raw;                                                        // stage-2 matrix fragments (16 KB); later the column sums
  char* const ring = smem_raw + A2_BYTES;                                            // streamed table fragments: 2 chunks x 8 KB
  float* const scratch = reinterpret_cast<float*>(smem_raw + A2_BYTES + RING_BYTES);
  unsigned* const img = reinterpret_cast<unsigned*>(smem_raw + A2_BYTES + RING_BYTES + SCRATCH_FLOATS * 4);    // split samples; later the log-mel rows
  float* const lm = reinterpret_cast<float*>(img);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.x;
  const unsigned lane16 = (unsigned)lane * 16u;
  const gptr_t tab = (gptr_t)p.tables;

  int kexp = 10;
  U16_STAMP(0);
  // ---- stage-2 matrix -> LDS; wave maxima reset
#pragma unroll
  for (int i = 0; i < A2_BYTES / (U16_THREADS * 16); ++i)
    *reinterpret_cast<u32x4*>(a2s + (i * U16_THREADS + tid) * 16) =
        *reinterpret_cast<const u32x4*>(p.tables + (size_t)T_A2 * 1024 + (i * U16_THREADS + tid) * 16);
  if (tid < 16) scratch[tid] = -INFINITY;

  // ---- the padded signal -> LDS (clamped, scaled by the utterance's 2^k, split into two f16 halves, word i + i / 160).  A thread owns groups
  // of four consecutive samples inside the utterance and the signal (16-byte loads, all in flight at once, kept in registers across the peak
  // reduction so that the image is written once); the padding at both ends -- and whatever hangs over an end of the signal when a window
  // does -- goes element by element through a small loop (raw value to LDS, scaled in place after the reduction)
  const int L = p.n + NFFT;
  {
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    const long long start = p.starts ? p.starts[b] : (long long)b * p.n;
    auto group_inner = [&](int g) -> bool {
      const int i = 4 * g;
      const long long gi = start + i - NFFT / 2;
      return i >= NFFT / 2 && i + 3 < NFFT / 2 + p.n && gi >= 0 && gi + 3 < p.n_total;
    };
    f32x4 v[U16_MAX_GROUPS];
    float amax = 0.f;
    bool bad = false;           // a NaN sample (v_med3 alone would launder it into -XMAX): the utterance's features become NaN, see the store phase
#pragma unroll
    for (int u = 0; u < U16_MAX_GROUPS; ++u) {
      const int g = tid + U16_THREADS * u;
      const bool in = group_inner(g);
      const f32x4 x = *reinterpret_cast<const f32x4u*>(p.wav + (in ? start + 4 * g - NFFT / 2 : 0));
#pragma unroll
      for (int c = 0; c < 4; ++c) v[u][c] = in ? x[c] : 0.f;
    }
    const bool hang = start < 0 || start + p.n > p.n_total;
    auto edge_range = [&](int lo, int hi, auto&& fn) {
      for (int i = lo + tid; i < hi; i += U16_THREADS)
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
      long long gi = start + sidx;
      ok = ok && gi >= 0 && gi < p.n_total;
      gi = gi < 0 ? 0 : (gi >= p.n_total ? p.n_total - 1 : gi);
      float x = p.wav[gi];
      bad |= ok && x != x;
      x = ok ? __builtin_amdgcn_fmed3f(x, -XMAX, XMAX) : 0.f;
      amax = fmaxf(amax, fabsf(x));
      img[i + i / HOP] = __float_as_uint(x);
    });
#pragma unroll
    for (int u = 0; u < U16_MAX_GROUPS; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bad |= v[u][c] != v[u][c];
        v[u][c] = __builtin_amdgcn_fmed3f(v[u][c], -XMAX, XMAX);
        amax = fmaxf(amax, fabsf(v[u][c]));
      }
    amax = sd_wave_max(amax);
    const bool wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
    if (lane == 0) { scratch[16 + wid] = amax; scratch[24 + wid] = wave_bad ? 1.f : 0.f; }
    U16_STAMP(1);
    __syncthreads();
    float mx = scratch[16];
#pragma unroll
    for (int i = 1; i < 8; ++i) mx = fmaxf(mx, scratch[16 + i]);
    const int e = (int)((__float_as_uint(mx) >> 23) & 0xFFu) - 126;
    if (mx >= 1e-30f) kexp = 14 - e;
    const float xs = __uint_as_float((unsigned)(kexp + 127) << 23);
    auto split = [&](float x) -> unsigned {
      const _Float16 hi = (_Float16)x;
      const _Float16 lo = (_Float16)(x - (float)hi);
      return (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
    };
#pragma unroll
    for (int u = 0; u < U16_MAX_GROUPS; ++u) {
      const int g = tid + U16_THREADS * u;
      if (group_inner(g)) {
        unsigned* const dst = img + 4 * g + (4 * g) / HOP;
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[c] = split(xs * v[u][c]);
      }
    }
    edges([&](int i) {
      unsigned* const w = img + i + i / HOP;
      *w = split(xs * __uint_as_float(*w));
    });
  }
  const float pback = __uint_as_float((unsigned)(10 - kexp + 127) << 23);
  U16_STAMP(2);
  __syncthreads();
  U16_STAMP(3);

  const int ntiles = (p.T + FT - 1) / FT;
  const int j = lane & 15, g = lane >> 4;
  const int hx = g & 1;                               // lane groups 2, 3 read the samples 0, 1 read (and take their low halves)
  const unsigned sel = g < 2 ? 0x05040100u : 0x07060302u;     // v_perm_b32 selector: the low (f16 hi) / high (f16 lo) 16 bits of two words
  // ---- the streamed tables.  Every wave of the workgroup needs the same fragments in the same order (stage-1 matrices, the real pair's
  // stage-2 matrix, the mel weights: 152 KB per tile round).  Read per wave from L2 they were the bottleneck of this kernel (8 waves x
  // 146 KB through the CU's 64 B/clk vector-memory path: ~18 k cycles per round, stamps); here they pass through a 16 KB LDS ring ONCE per
  // round: chunk q = stream fragments 8 q .. 8 q + 7, wave w fetches fragment w by LDS-DMA, a workgroup barrier in front of each chunk's
  // first use (19 per round) says "chunk q has landed, chunk q - 1 is no longer read", and every wave then requests its piece of chunk q + 1.
  // Waves without a tile in a round still fetch and join the barriers.
  const gptr_t stream = tab + (size_t)T_STREAM * 1024 + (size_t)wid * 1024 + lane16;
  auto dma = [&](int q) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stream + (size_t)q * 8192),
                                     (__attribute__((address_space(3))) void*)(ring + ((q & 1) * 8 + wid) * 1024), 16, 0, 0);
  };
  auto chunk_sync = [&](int q) {                      // in front of the first use of chunk q
#ifdef SD_U16_DIAG_NOWAIT      // timing-only diagnostic build (results wrong): what the wait for the chunk's own DMA costs
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#elif defined(SD_U16_DIAG_NOBARRIER)   // timing-only diagnostic build (results wrong): what the lock step costs
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#endif
    if (q + 1 < NCHUNK) dma(q + 1);
  };
  // LDS address (+ 16 lane) of stream-relative item i (= fragments 2 i, 2 i + 1) of the current round
  const char* const ringl = ring + lane16;
  auto item = [&](int i) -> const char* { return ringl + ((((i >> 2) & 1) * 8) + 2 * (i & 3)) * 1024; };

  const int nrounds = ntiles > 8 ? 2 : 1;
#pragma unroll 1
  for (int round = 0; round < nrounds; ++round) {
    // A wave without a tile in this round computes on frame T - 1 and stores nothing: it has to walk the barriers anyway, its SIMD partner
    // has a tile, and the round takes as long as the SIMD that carries two (no branches around the phases: fewer live ranges to merge)
    const int tile = wid + 8 * round;
    const int f0 = tile * FT;
    int nvalid = p.T - f0; nvalid = nvalid > FT ? FT : (nvalid < 0 ? 0 : nvalid);
    int f = f0 + (j < nvalid ? j : 0); f = f < p.T ? f : p.T - 1;
    const unsigned* const x0 = img + 161 * f + 201 * hx;     // word of x[200 hx]; x[200 hx + m] at x0 + m + [m >= (hx ? 120 : 160)]
    const unsigned* const xm = x0 + hx;
    __syncthreads();                                  // nobody reads the ring any more (previous round / the signal is staged)
    dma(0);
    // split stage-1 sums: [n2] x {hi, lo} x {components 0-1, 2-3} of this lane group's pair; n2 = 25..27 padding
    unsigned yh01[4 * NBLK], yl01[4 * NBLK], yh23[4 * NBLK], yl23[4 * NBLK];
    {
      // ---- stage 1: per n2 three MFMAs (hi.hi + hi.lo + lo.hi); the signal words of n2 + 1 are requested ahead, the previous n2's sums
      // are split while this one's MFMAs run
      auto xword = [&](int e, int n2) -> unsigned {
        const int m = 25 * e + n2;
        return m < 120 ? x0[m] : (m < 160 ? xm[m] : x0[m + 1]);
      };
      // n2 in pairs (2 m, 2 m + 1; 25 is padding): the two three-MFMA chains interleave; the signal words of the next pair are requested
      // in front of this pair's MFMAs, the previous pair's sums are split behind them
      constexpr int NP = (NN2 + 1) / 2;
      unsigned d[2][2][8];
      auto xload = [&](int buf, int m) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          d[buf][0][e] = xword(e, 2 * m);
          d[buf][1][e] = 2 * m + 1 < NN2 ? xword(e, 2 * m + 1) : 0u;
        }
      };
      xload(0, 0);
      f32x4 c1[2][2];
      static_for<0, NP + 1>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        if constexpr (m < NP && m % 2 == 0) chunk_sync((S_A1 / 2 + 2 * m) / 4);
        if constexpr (m < NP) {
          constexpr int ia = S_A1 / 2 + 2 * m, ib = (2 * m + 1 < NN2) ? ia + 1 : ia;
          const h8v ah0 = *reinterpret_cast<const h8v*>(item(ia)), al0 = *reinterpret_cast<const h8v*>(item(ia) + 1024);
          const h8v ah1 = *reinterpret_cast<const h8v*>(item(ib)), al1 = *reinterpret_cast<const h8v*>(item(ib) + 1024);
          if constexpr (m + 1 < NP) xload((m + 1) & 1, m + 1);
          const unsigned* d0 = d[m & 1][0];
          const unsigned* d1 = d[m & 1][1];
          // lane groups 0, 1 take the HIGH halves of their 8 samples, groups 2, 3 the LOW halves of the same samples (the selector is per lane):
          // with the matrix fragments [A1h | A1h] and [A1l | 0] over k = 32, two MFMAs give A1h xh + A1h xl + A1l xh
          const h8v xb0 = frag_of(__builtin_amdgcn_perm(d0[1], d0[0], sel), __builtin_amdgcn_perm(d0[3], d0[2], sel),
                                  __builtin_amdgcn_perm(d0[5], d0[4], sel), __builtin_amdgcn_perm(d0[7], d0[6], sel));
          const h8v xb1 = frag_of(__builtin_amdgcn_perm(d1[1], d1[0], sel), __builtin_amdgcn_perm(d1[3], d1[2], sel),
                                  __builtin_amdgcn_perm(d1[5], d1[4], sel), __builtin_amdgcn_perm(d1[7], d1[6], sel));
          f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
          z0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xb0, z0, 0, 0, 0);
          z1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xb1, z1, 0, 0, 0);
          c1[m & 1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, xb0, z0, 0, 0, 0);
          c1[m & 1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, xb1, z1, 0, 0, 0);
        }
        if constexpr (m > 0) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int n2 = 2 * (m - 1) + q;
            if (n2 < NN2) {
              const f32x4& c = c1[(m - 1) & 1][q];
              const float a0 = YSCALE * c[0], a1 = YSCALE * c[1], a2 = YSCALE * c[2], a3 = YSCALE * c[3];
              const unsigned h01 = pk_rtz(a0, a1), h23 = pk_rtz(a2, a3);
              yh01[n2] = h01;
              yl01[n2] = pk_rtz(a0 - f16lo(h01), a1 - f16hi(h01));
              yh23[n2] = h23;
              yl23[n2] = pk_rtz(a2 - f16lo(h23), a3 - f16hi(h23));
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);            // one pair at a time: left alone, the scheduler pulls the reads of many steps forward and spills the sums
      });
#pragma unroll
      for (int i = NN2; i < 4 * NBLK; ++i) { yh01[i] = 0u; yl01[i] = 0u; yh23[i] = 0u; yl23[i] = 0u; }
    }
    U16_STAMP(4 + 4 * round);
    // (the barrier of the next chunk also says: every wave is done with stage 1 -- the log-mel rows of the first eight tiles overwrite
    // signal words that only those tiles' stage 1 reads)
    chunk_sync(S_A2S / 8);
    // ---- registers <-> lane groups: block m (n2 = 4 m + i in register i, pair g in lane group g) -> pair i in register i, n2 = 4 m + g in
    // lane group g.  After it, element m of array P<kind>[i] is pair i's k-step data: P..[i][m] for m = 4 s .. 4 s + 3 = B fragment of step s
    unsigned Ph01[4][NBLK], Pl01[4][NBLK], Ph23[4][NBLK], Pl23[4][NBLK];
    unsigned pw1[NPROB][4], pw2[NPROB][4];
    f32x4 c2s[4];
    {
#pragma unroll
      for (int m = 0; m < NBLK; ++m) {
        unsigned r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = yh01[4 * m + i];
        transpose4(r);
#pragma unroll
        for (int i = 0; i < 4; ++i) Ph01[i][m] = r[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = yl01[4 * m + i];
        transpose4(r);
#pragma unroll
        for (int i = 0; i < 4; ++i) Pl01[i][m] = r[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = yh23[4 * m + i];
        transpose4(r);
#pragma unroll
        for (int i = 0; i < 4; ++i) Ph23[i][m] = r[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = yl23[4 * m + i];
        transpose4(r);
#pragma unroll
        for (int i = 0; i < 4; ++i) Pl23[i][m] = r[i];
      }
      U16_STAMP(5 + 4 * round);
      // ---- stage 2: problem 2 i = components 0-1 of pair i, problem 2 i + 1 = components 2-3; the seven complex ones on the resident matrix
      const char* const a2l = a2s + lane16;
      auto resident = [&](int st) -> const char* { return a2l + (((st & 3) * 2 + (st >> 2)) * 2) * 1024; };
      // (two problems at a time against each fragment pair: problems 2 i, 2 i + 1 = the two of pair i; problem 6 alone)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        f32x4 ca[4], cb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { ca[u] = f32x4{0.f, 0.f, 0.f, 0.f}; cb[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        u16_steps2(resident, Ph01[i], Pl01[i], Ph23[i], Pl23[i], ca, cb);
        u16_powers(ca, pw1[2 * i], pw2[2 * i]);
        u16_powers(cb, pw1[2 * i + 1], pw2[2 * i + 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        f32x4 c2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) c2[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        u16_steps<0, 8>(resident, Ph01[3], Pl01[3], c2);
        u16_powers(c2, pw1[6], pw2[6]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // the pair of reals: its own matrix, streamed (items S_A2S / 2 .. + 7 in step order (s, u))
#pragma unroll
      for (int u = 0; u < 4; ++u) c2s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      u16_steps<0, 4>([&](int st) -> const char* { return item(S_A2S / 2 + st); }, Ph23[3], Pl23[3], c2s);
    }
    chunk_sync(S_A2S / 8 + 1);
    {
      u16_steps<4, 8>([&](int st) -> const char* { return item(S_A2S / 2 + st); }, Ph23[3], Pl23[3], c2s);
      u16_powers(c2s, pw1[NPROB - 1], pw2[NPROB - 1]);
    }
    U16_STAMP(6 + 4 * round);
    // ---- mel: 5 row tiles x 8 problems, one K = 32 step each, split bf16 (W1 P1 + W1 P2 + W2 P1); weights from the ring (items S_MEL / 2 + it)
    f32x4 mel[MEL_TILES];
#pragma unroll
    for (int t = 0; t < MEL_TILES; ++t) mel[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    static_for<0, MEL_TILES * NPROB / 2>([&](auto itc) {
      // items (problem pr = it / 5, tile t = it % 5) two at a time: consecutive items feed different accumulators, their chains interleave
      constexpr int it = 2 * decltype(itc)::value;
      constexpr int t0 = it % MEL_TILES, pr0 = it / MEL_TILES, t1 = (it + 1) % MEL_TILES, pr1 = (it + 1) / MEL_TILES;
      if constexpr (it % 4 == 0) chunk_sync(S_MEL / 8 + it / 4);
      const bf8v a10 = *reinterpret_cast<const bf8v*>(item(S_MEL / 2 + it)), a20 = *reinterpret_cast<const bf8v*>(item(S_MEL / 2 + it) + 1024);
      const bf8v a11 = *reinterpret_cast<const bf8v*>(item(S_MEL / 2 + it + 1)), a21 = *reinterpret_cast<const bf8v*>(item(S_MEL / 2 + it + 1) + 1024);
      const bf8v b10 = __builtin_bit_cast(bf8v, (u32x4{pw1[pr0][0], pw1[pr0][1], pw1[pr0][2], pw1[pr0][3]}));
      const bf8v b20 = __builtin_bit_cast(bf8v, (u32x4{pw2[pr0][0], pw2[pr0][1], pw2[pr0][2], pw2[pr0][3]}));
      const bf8v b11 = __builtin_bit_cast(bf8v, (u32x4{pw1[pr1][0], pw1[pr1][1], pw1[pr1][2], pw1[pr1][3]}));
      const bf8v b21 = __builtin_bit_cast(bf8v, (u32x4{pw2[pr1][0], pw2[pr1][1], pw2[pr1][2], pw2[pr1][3]}));
      mel[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a10, b10, mel[t0], 0, 0, 0);
      mel[t1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a11, b11, mel[t1], 0, 0, 0);
      mel[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a10, b20, mel[t0], 0, 0, 0);
      mel[t1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a11, b21, mel[t1], 0, 0, 0);
      mel[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a20, b10, mel[t0], 0, 0, 0);
      mel[t1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a21, b11, mel[t1], 0, 0, 0);
    });
    {
      // ---- log, utterance maximum, rows -> LDS (no branches: what does not exist goes to a spare word and counts as -inf)
      const bool ln = p.log_mode == SD_LOG_LN_EPS;
      const float lscale = ln ? 0.6931471805599453f : 3.0102999566398120f;      // v_log_f32 is log2
      float vmax = -INFINITY;
      float* const mrow = lm + (size_t)(f0 + j) * MELP;
      float* const spare = scratch + 32 + (lane & 31);
      const bool jok = j < nvalid;
#pragma unroll
      for (int t = 0; t < MEL_TILES; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * t + 4 * g + r;
          const bool ok = jok && m < p.n_mels;
          const float v = mel[t][r] * pback * pback;
          const float lv = lscale * __builtin_amdgcn_logf(ln ? v + p.log_eps : fmaxf(v, p.log_eps));
          *(ok ? mrow + m : spare) = lv;
          vmax = fmaxf(vmax, ok ? lv : -INFINITY);
        }
      vmax = sd_wave_max(vmax);
      if (lane == 0) scratch[round * 8 + wid] = vmax;
    }
    U16_STAMP(7 + 4 * round);
  }
  __syncthreads();
  U16_STAMP(14);

  // ---- top_db floor relative to the utterance maximum, mean over T per mel bin, the one write of the output
  float thr = -INFINITY;
  if (p.use_floor) {
    float mx = scratch[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, scratch[i]);
    thr = mx - p.top_db;
  }
  float* const part = reinterpret_cast<float*>(a2s);          // (the stage-2 matrix is dead) [24][96] partial column sums, then [96] means behind them
  float* const mean = part + 24 * 96;
  if (p.mean_norm) {
    const int per_row = (p.n_mels + 3) >> 2;
    int RG = U16_THREADS / per_row; RG = RG > 24 ? 24 : RG;
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
  {
    // a NaN sample anywhere in the utterance: the reference's arithmetic carries it into the utterance maximum (top_db floor) and the mean over
    // T, i.e. into every value of the utterance's features.  Here the sample was clamped on its way into the f16 image, so the rows are
    // overwritten: bad input shows as NaN features (and a NaN embedding), never as a plausible finite row.
    float nb = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) nb += scratch[24 + i];
    if (nb != 0.f) {
      const int total = p.T * p.n_mels;
      for (int e = tid; e < total; e += U16_THREADS) {
        const int t = e / p.n_mels, c = e - t * p.n_mels;
        orow[(size_t)t * p.ld_out + c] = __int_as_float(0x7FC00000);
      }
      return;
    }
  }
  if ((p.n_mels & 3) == 0 && (p.ld_out & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15u) == 0) {
    const int per_row = p.n_mels >> 2;
    const int total = p.T * per_row;
    const unsigned inv = p.inv_mels;
#pragma unroll 4
    for (int e = tid; e < total; e += U16_THREADS) {
      const int t = (int)__umulhi((unsigned)e, inv);
      const int c = (e - t * per_row) * 4;
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaxf(lm[t * MELP + c + i], thr) - (p.mean_norm ? mean[c + i] : 0.f);
      *reinterpret_cast<f32x4*>(orow + (size_t)t * p.ld_out + c) = v;
    }
  } else {
    const int total = p.T * p.n_mels;
    for (int e = tid; e < total; e += U16_THREADS) {
      const int t = e / p.n_mels, c = e - t * p.n_mels;
      orow[(size_t)t * p.ld_out + c] = fmaxf(lm[t * MELP + c], thr) - (p.mean_norm ? mean[c] : 0.f);
    }
  }
  U16_STAMP(15);
}

size_t u16_lds_bytes(int n, int T) {
  const size_t words = (size_t)img_words(n) > (size_t)T * MELP ? (size_t)img_words(n) : (size_t)T * MELP;
  return (size_t)A2_BYTES + RING_BYTES + SCRATCH_FLOATS * 4 + words * 4;
}

unsigned short bf16_bits(float v) {
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

// what row (lane group G, component c) of the stage-1 product is: k1 and re / im, or one of the two reals
struct RowOf { int k1; int kind; };      // kind 0: re, 1: im, 2: Y[0] (real), 3: rho (Y[8] rotated to the real axis)
RowOf row_of(int G, int c) {
  if (G < 3) return RowOf{1 + 2 * G + (c >> 1), c & 1};
  if (c < 2) return RowOf{7, c};
  return RowOf{c == 2 ? 0 : 8, c};
}

}  // namespace

#ifdef SD_STAMP
extern "C" int sd_debug_read_utt16_stamps(unsigned long long* host, int n) {
  SD_CHECK_ARG(host && n == 1024 * 8 * 16, "sd_debug_read_utt16_stamps: need 131072 entries");
  SD_CHECK_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_utt16_stamps), sizeof(unsigned long long) * n));
  return SD_OK;
}
#endif

int sd_fbank_utt16_create_tables(sd_fbank_plan* plan, const float* window, const float* mel_fb) {
  const int n_mels = plan->n_mels;
  std::vector<unsigned short> tab((size_t)T_FRAGS * 512);
  auto put_f16 = [&](size_t frag_hi, size_t frag_lo, int l, int e, double v) {
    const _Float16 hi = (_Float16)(float)v;
    const _Float16 lo = (_Float16)(float)(v - (double)(float)hi);
    tab[frag_hi * 512 + l * 8 + e] = __builtin_bit_cast(unsigned short, hi);
    tab[frag_lo * 512 + l * 8 + e] = __builtin_bit_cast(unsigned short, lo);
  };
  // stage 1: A operand of v_mfma_f32_16x16x32_f16: lane l holds row l & 15, k = 8 (l >> 4) + e.  The B operand carries the HIGH halves of the
  // samples n1 = k in k < 16 and their LOW halves in k >= 16 (n1 = k - 16), so fragment 0 = [A1h | A1h] and fragment 1 = [A1l | 0]:
  // two MFMAs = A1h xh + A1h xl + A1l xh
  for (int n2 = 0; n2 < NN2; ++n2)
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 8; ++e) {
        const int rho = l & 15, k = 8 * (l >> 4) + e, n1 = k & 15;
        const int n = 25 * n1 + n2;
        const RowOf r = row_of(rho >> 2, rho & 3);
        const double ang = 2.0 * M_PI * (double)(((long)r.k1 * n) % NFFT) / (double)NFFT;
        const double w = A1SCALE * (double)window[n];
        double v;
        if (r.kind == 0) v = w * std::cos(ang);
        else if (r.kind == 1) v = -w * std::sin(ang);
        else if (r.kind == 2) v = w;
        else v = w * std::cos(ang - M_PI * (double)n2 / 25.0);
        const _Float16 hi = (_Float16)(float)v;
        const _Float16 lo = (_Float16)(float)(v - (double)(float)hi);
        const size_t f0 = (size_t)T_STREAM + S_A1 + 2 * n2;
        tab[f0 * 512 + l * 8 + e] = __builtin_bit_cast(unsigned short, hi);
        tab[(f0 + 1) * 512 + l * 8 + e] = k < 16 ? __builtin_bit_cast(unsigned short, lo) : (unsigned short)0;
      }
  // stage 2: fragment (u, s): rows 16 u + (l & 15): r = row & 3 = 2 q + out part, output idx = 2 (row >> 2 & 3) + q + 8 u;
  // k = 8 (l >> 4) + e: m' = e >> 1, in part = e & 1, n2 = 4 (4 s + m') + (l >> 4)
  for (int special = 0; special < 2; ++special)
    for (int u = 0; u < 4; ++u)
      for (int s = 0; s < 2; ++s)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 8; ++e) {
            const int rho = l & 15, Go = rho >> 2, r = rho & 3;
            const int idx = 2 * Go + (r >> 1) + 8 * u, oim = r & 1;
            const int n2 = 4 * (4 * s + (e >> 1)) + (l >> 4), yim = e & 1;
            double v = 0.0;
            if (n2 < NN2) {
              if (!special) {
                if (idx < NN2) {
                  const double th = 2.0 * M_PI * (double)((n2 * idx) % NN2) / (double)NN2;
                  v = oim == 0 ? (yim == 0 ? std::cos(th) : std::sin(th)) : (yim == 0 ? -std::sin(th) : std::cos(th));
                }
              } else if (idx <= 12) {                 // k1 = 0: X[16 idx] from the real Y[0, n2] (in part 0)
                const double th = 2.0 * M_PI * (double)((n2 * idx) % NN2) / (double)NN2;
                if (yim == 0) v = oim == 0 ? std::cos(th) : -std::sin(th);
              } else if (idx <= 25) {                 // k1 = 8: X[8 + 16 k2], k2 = idx - 13, from rho[n2] (in part 1): e^{-i pi n2 (2 k2 + 1) / 25}
                const double ph = M_PI * (double)((n2 * (2 * (idx - 13) + 1)) % 50) / 25.0;
                if (yim == 1) v = oim == 0 ? std::cos(ph) : -std::sin(ph);
              }
            }
            // resident matrix: fragment pair (u, s) at index (u * 2 + s) * 2; the real pair's, streamed: in step order st = 4 s + u
            const size_t base = special ? (size_t)T_STREAM + S_A2S + (size_t)(4 * s + u) * 2 : (size_t)T_A2 + (size_t)(u * 2 + s) * 2;
            put_f16(base, base + 1, l, e, A2SCALE * v);
          }
  // mel weights: problem pr = 2 i + (components 2-3 ? 1 : 0) of pair i; lane l holds row (mel) 16 t + (l & 15), k = 8 (l >> 4) + e:
  // u = e >> 1, q = e & 1, output idx = 2 (l >> 4) + q + 8 u of the problem
  for (int pr = 0; pr < NPROB; ++pr)
    for (int t = 0; t < MEL_TILES; ++t)
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 8; ++e) {
          const int m = 16 * t + (l & 15), idx = 2 * (l >> 4) + (e & 1) + 8 * (e >> 1);
          int bin = -1;
          if (pr < 7) {
            const int k1 = pr + 1;                    // pairs: (1, 2), (3, 4), (5, 6), (7, reals)
            if (idx <= 12) bin = k1 + 16 * idx;
            else if (idx <= 24) bin = 16 - k1 + 16 * (24 - idx);
          } else {
            if (idx <= 12) bin = 16 * idx;
            else if (idx <= 25) bin = 8 + 16 * (idx - 13);
          }
          float w = 0.f;
          if (bin >= 0 && bin < NFREQ && m < n_mels) w = (float)((double)mel_fb[(size_t)bin * n_mels + m] * PSCALE);
          const unsigned short w1 = bf16_bits(w);
          const size_t f0 = (size_t)T_STREAM + S_MEL + (size_t)(pr * MEL_TILES + t) * 2;   // consumption order: problem-major
          tab[f0 * 512 + l * 8 + e] = w1;
          tab[(f0 + 1) * 512 + l * 8 + e] = bf16_bits(w - bf16_value(w1));
        }
  plan->utt16_tables_dev = nullptr;
  hipError_t e = hipMalloc(&plan->utt16_tables_dev, (size_t)T_FRAGS * 1024);
  if (e == hipSuccess) e = hipMemcpy(plan->utt16_tables_dev, tab.data(), (size_t)T_FRAGS * 1024, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    sd_fbank_utt16_destroy_tables(plan);
    return sd_set_error(SD_ERR_HIP, "sd_fbank_plan_create: device table upload failed: %s", hipGetErrorString(e));
  }
  return SD_OK;
}

void sd_fbank_utt16_destroy_tables(sd_fbank_plan* plan) {
  if (plan->utt16_tables_dev) (void)hipFree(plan->utt16_tables_dev);
  plan->utt16_tables_dev = nullptr;
}

bool sd_fbank_utt16_supported(const sd_fbank_plan* plan, int n) {
  static const bool on = [] { const char* e = sd_experiment_env("SD_FBANK_UTT"); return !(e && atoi(e) == 0); }();   // A/B: SD_FBANK_UTT=0 sends every length to the folded kernel
  if (!on || !plan->utt16_tables_dev) return false;
  const int T = 1 + n / HOP;
  return T <= U16_MAX_T && u16_lds_bytes(n, T) <= (size_t)LDS_LIMIT;
}

int sd_fbank_utt16_launch(const sd_fbank_plan* plan, const float* wav_dev, long long n_total, const long long* starts_dev, int B, int n,
                          int mean_norm, float* out_dev, int ld_out, hipStream_t stream) {
  const int T = 1 + n / HOP;
  U16Args a;
  a.wav = wav_dev; a.B = B; a.n = n; a.T = T;
  a.starts = starts_dev; a.n_total = n_total;
  a.tables = static_cast<const char*>(plan->utt16_tables_dev);
  a.n_mels = plan->n_mels; a.pad_mode = plan->pad_mode; a.log_mode = plan->log_mode; a.log_eps = plan->log_eps;
  a.top_db = plan->top_db;
  a.use_floor = plan->log_mode == SD_LOG_DB_TOPDB && plan->top_db >= 0.f;
  a.mean_norm = mean_norm;
  a.out = out_dev; a.ld_out = ld_out;
  const int per_row = plan->n_mels % 4 == 0 ? plan->n_mels / 4 : plan->n_mels;
  a.inv_mels = (unsigned)((((unsigned long long)1 << 32) + per_row - 1) / per_row);
  const size_t lds = u16_lds_bytes(n, T);
  SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(fbank_utt16_kernel), LDS_LIMIT));
  {
    SdProfScope prof(SD_PROF_FBANK, stream, (double)B * ((double)n * 4.0 + (double)T * plan->n_mels * 4.0));
    hipLaunchKernelGGL(fbank_utt16_kernel, dim3((unsigned)B), dim3(U16_THREADS), lds, stream, a);
  }
  SD_CHECK_LAUNCH("fbank_utt16_kernel");
  return SD_OK;
}
