// Fused log-mel filterbank for 25 ms / 10 ms framing at 16 kHz (n_fft = 400, hop = 160).
//
// Replaces torchaudio MelSpectrogram + log + per-utterance mean removal
// [REF speech_encode.py:17-36] and the speechbrain Fbank + sentence mean-norm
// front end inside EncoderClassifier.encode_batch [REF speech_encode.py:77]
// (SURVEY.md Appendix A.1 / A.2).  One launch goes waveform -> log-mel; a second,
// tiny launch applies the utterance-level top_db floor and mean removal.
//
// Algorithm.  The window is symmetric (w[k] == w[400-k]), so the 400-point real
// DFT folds to K = 201:
//     Re X[n] = sum_{k=0..200} Cb[k][n] * (x[k] + x[400-k]),   Cb = w[k] cos(2 pi k n / 400)
//     Im X[n] = sum_{k=1..199} Sb[k][n] * (x[k] - x[400-k]),   Sb = w[k] sin(2 pi k n / 400)
// (x[400] := 0 for k = 0, Cb[200] halved), i.e. two GEMMs  [bins x K] * [K x frames] with a host-computed basis.
// A wave owns 32 consecutive frames and keeps their sample span in LDS (each sample is fetched from HBM once per
// tile, 4.5 % overlap); tiles are flat over (utterance, frame) so no lane idles on T = 201.
//
// The products run on the F16 matrix cores at f32 accuracy.  The exact-f32 MFMA runs at 1/16 of the f16 rate, and
// round 1's kernel (v_mfma_f32_32x32x2_f32) was bound by it: ideal MFMA time 1.4 ms of its 3.0 ms per 5000 segments.
// Here every f32 operand is split into two halves, v = hi + lo with hi = f16(v) and lo = f16(v - hi), and a product
// is  hi.hi + hi.lo + lo.hi  on v_mfma_f32_32x32x16_f16 with f32 accumulation: the dropped lo.lo term and the
// representation error are 2^-22 relative per product, i.e. f32-level, for three MFMAs that do sixteen times the
// work of an f32 one (the scheme of the split-precision affinity, sd_pool.hip).  The samples are scaled by 2^10
// (folded sums <= 2048 < f16 max) so that the LOW halves stay clear of the f16 subnormals down to signals of 1e-7
// full scale (what matters for a quiet segment is the error relative to ITS level: unscaled, a -100 dBFS segment
// would keep only ~12 bits); the power spectrum takes 2^-20 back, exactly.
//   * DFT: 6 MFMAs per (32-bin tile, 16 k): 546 of 32 cycles per wave tile instead of 1428 of 64.
//   * mel: |X|^2 goes from the DFT accumulators straight into the next MFMA as its B operand (the accumulator rows are
//     the next product's k: no lane movement), split into two bf16 halves (f32 exponent range), against mel weights
//     split the same way: 3 products on v_mfma_f32_32x32x16_bf16, 2^-16 relative (the log needs 2e-4).
//   * loop order: k outer, bin tiles inner, in two passes (bin tiles 0-3, then 4-6) so that a pass's accumulators
//     (128 registers) stay resident and the folded, split signal fragments are built once per pass and k step, not per
//     bin tile; the basis fragments of a (pass, k step) are one contiguous 16 / 12 KB block that the four waves share
//     through a double-buffered LDS stage filled by LDS-DMA.
//   * LDS sample image: sample i of a span sits at word i + 4 (i / 160): frame starts (stride 164 words) and every
//     aligned group of 4 samples stay 16-byte aligned, and the 16 frames of a ds_read_b128 lane group fall on 16
//     distinct bank slots (41 j mod 16 = 9 j).
// Measured (MI355X, rocprofv3, 10 000 segments of 2 s per launch): 2.32 ms main kernel + 0.45 ms finalize (round 1: 6.0 + 0.42),
// 830 GB/s of the 192 320 algorithmic bytes per segment for the main kernel alone, 695 GB/s with the finalize pass.  Since round 4
// utterances of up to 32 100 samples (201 frames) take the one-launch kernel of sd_fbank_utt16.hip (1.6 ms for the same work); this kernel serves
// the longer ones.  Two things found on the way (both hipcc codegen, both worth ~1 ms per 5000 segments):
// a multiply placed next to the staging loads made every load wait on its own (79 `s_waitcnt vmcnt(0)`), and predicated
// loads become branches with a wait each; the staging loop therefore loads unconditionally from a clamped index and
// selects the zero padding afterwards.
#include "sd_fbank_internal.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int NFFT = 400;
constexpr int HOP = 160;
constexpr int NFREQ = 201;
constexpr int NBT = 7;       // 32-bin tiles (224 >= 201)
constexpr int FT = 32;       // frames per wave tile
constexpr int WAVES = 4;
constexpr int MELP = 81;     // mel accumulator row stride
constexpr int MAX_MELS = 80;


__device__ __forceinline__ int f32_key(float v) {
  const int b = __float_as_int(v);
  return b >= 0 ? b : b ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float key_f32(int k) {
  return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF);
}

#ifndef SD_FB_SCALE
#define SD_FB_SCALE 1024.f
#endif
#define SD_FB_POISON_KEY 0x7FFFFFFF                  // key of an utterance that holds a NaN sample (key_f32 of it is a NaN)
#define SD_FB_XMAX 16.0f                            // 2 * 16 * 1024 = 32768 < 65504 (f16 max)
constexpr int V2_XS_W = 5952;                       // per-wave sample image, floats (two spans + skew + alignment slack)
constexpr int V2_STAGE_BYTES = 16 * 1024;           // basis fragments of one (pass, k step): 4 tiles x 4 kinds x 1 KB
constexpr int V2_KSTEPS = 13;                       // 208 / 16
#ifndef SD_FB_GROUPS
#define SD_FB_GROUPS 2           // wave groups per workgroup: the 7 bin tiles of a 32-frame tile are split 4 + 3 over TWO waves on the same SIMD
#endif                           // that share the tile's sample image.  3: 3 + 2 + 2 over three waves (measured: 1.369 vs 1.358 ms, no gain over
                                 // two); 1: round 2's four waves, two passes in sequence (1.61 ms)
constexpr int V2_GROUPS = SD_FB_GROUPS;
static_assert(V2_GROUPS >= 1 && V2_GROUPS <= 3, "");
// bin tiles [G_Q0, G_Q0 + G_NT) of group g (SD_FB_GROUPS = 1 keeps the 4 + 3 split as two sequential passes)
__host__ __device__ constexpr int fb_nt(int g) { return V2_GROUPS == 3 ? (g == 0 ? 3 : 2) : (g == 0 ? 4 : 3); }
__host__ __device__ constexpr int fb_q0(int g) { return V2_GROUPS == 3 ? (g == 0 ? 0 : (g == 1 ? 3 : 5)) : (g == 0 ? 0 : 4); }
constexpr int V2_NPASS = V2_GROUPS == 1 ? 2 : V2_GROUPS;                     // basis blocks are packed per pass
__host__ __device__ constexpr size_t fb_basis_off(int g) {                    // bytes in front of pass g's blocks
  size_t o = 0;
  for (int i = 0; i < g; ++i) o += (size_t)13 * fb_nt(i) * 4 * 1024;
  return o;
}
__host__ __device__ constexpr int fb_ring_off(int g) {                        // bytes in front of group g's two-stage ring in LDS
  int o = 0;
  for (int i = 0; i < g; ++i) o += 2 * fb_nt(i) * 4 * 1024;
  return o;
}
constexpr int V2_RING_BYTES = V2_GROUPS == 1 ? 2 * V2_STAGE_BYTES : fb_ring_off(V2_GROUPS);
constexpr int V2_LDS_BYTES = WAVES * V2_XS_W * 4 + V2_RING_BYTES;
static_assert(V2_LDS_BYTES <= 160 * 1024, "sample images + basis stages must fit the CU's LDS");
static_assert(WAVES * FT * MELP * 4 <= WAVES * V2_XS_W * 4, "the log-mel staging reuses the sample image");
constexpr size_t V2_BASIS_BYTES = (size_t)V2_KSTEPS * NBT * 4 * 1024;        // 364 KB
constexpr size_t V2_MELW_BYTES = (size_t)NBT * 2 * 3 * 2 * 1024;             // 84 KB

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));

struct Fbank2Args {
  const float* wav; int B; int n; int T;
  const long long* starts;  // windowed entry: row b = wav[starts[b] .. starts[b] + n), zeros outside [0, n_total); null: row b = wav + b n
  long long n_total;        // samples behind `wav`
  const _Float16* basis;   // [pass][k step][tile in pass][Chi | Clo | Shi | Slo][64 lanes][8]
  const __bf16* melw;      // [bin tile][k half][mel tile][W1 | W2][64 lanes][8]
  int n_mels; int pad_mode; int log_mode; float log_eps;
  float* out; int ld_out;
  int* maxbuf;
  int flat;
  unsigned inv_mels;        // ceil(2^32 / n_mels)
};

#define FB2_GLDS16(gptr, lptr)                                                             \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),  \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// one pass over NTILE bin tiles starting at tile Q0: DFT into acc, then their power spectra into the mel accumulators
template <int NTILE, int Q0, int PASS, bool ZERO_MEL>
__device__ __forceinline__ void fbank2_pass(const Fbank2Args& p, const float* xs, int base, char* stage, int lane, int wid,
                                            f32x16 (&mel)[3]) {
  const int j = lane & 31, h = lane >> 5;
  (void)j;
  f32x16 re[NTILE], im[NTILE];
#pragma unroll
  for (int q = 0; q < NTILE; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) { re[q][r] = 0.f; im[q][r] = 0.f; }
  constexpr int STAGE = NTILE * 4 * 1024;                                     // bytes of one (pass, k step) block
  const char* const gsrc = reinterpret_cast<const char*>(p.basis) + fb_basis_off(PASS);
  auto dma = [&](int s, int buf) {                                            // wave w moves pieces w, w + 4, ...
#pragma unroll
    for (int pc = 0; pc < NTILE; ++pc) {
      const int piece = pc * WAVES + wid;
      FB2_GLDS16(gsrc + (size_t)s * STAGE + piece * 1024 + lane * 16, stage + buf * STAGE + piece * 1024);
    }
  };
  dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < V2_KSTEPS; ++s) {
    if (s + 1 < V2_KSTEPS) dma(s + 1, (s + 1) & 1);
    // ---- folded signal of this lane: k = 16 s + 8 h + e, e < 8:  a = x[k] + x[400 - k],  d = x[k] - x[400 - k]
    const int g = 2 * s + h;                                                  // group of 8 k
    const float* fw = xs + base + 8 * g + 4 * (g / 20);                       // x[8 g ..]: 8 g and 8 g + 7 share a block of 160
    const f32x4 f0 = *reinterpret_cast<const f32x4*>(fw), f1 = *reinterpret_cast<const f32x4*>(fw + 4);
    // x[400 - 8 g - e]: aligned blocks starting at sample 392 - 8 g, 396 - 8 g, 400 - 8 g
    auto blk = [&](int i0) { return *reinterpret_cast<const f32x4*>(xs + base + i0 + 4 * (i0 / HOP)); };
    const int i0 = 392 - 8 * g;
    const f32x4 b0 = blk(i0), b1 = blk(i0 + 4), b2 = blk(i0 + 8);
    float fwd[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
    float bwd[8] = {b2[0], b1[3], b1[2], b1[1], b1[0], b0[3], b0[2], b0[1]};
    if (g == 0) bwd[0] = 0.f;                                                 // x[400] := 0
    h8v ahi, alo, dhi, dlo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // scaled HERE, not when the samples are staged: a multiply next to the staging loads makes hipcc wait for every
      // load separately (79 `s_waitcnt vmcnt(0)`, 1.0 ms more per 5000 segments)
      const float a = SD_FB_SCALE * (fwd[e] + bwd[e]), d = SD_FB_SCALE * (fwd[e] - bwd[e]);
      ahi[e] = (_Float16)a; alo[e] = (_Float16)(a - (float)ahi[e]);
      dhi[e] = (_Float16)d; dlo[e] = (_Float16)(d - (float)dhi[e]);
    }
    const char* st = stage + (s & 1) * STAGE + lane * 16;
#pragma unroll
    for (int q = 0; q < NTILE; ++q) {
      const h8v chi = *reinterpret_cast<const h8v*>(st + (q * 4 + 0) * 1024);
      const h8v clo = *reinterpret_cast<const h8v*>(st + (q * 4 + 1) * 1024);
      const h8v shi = *reinterpret_cast<const h8v*>(st + (q * 4 + 2) * 1024);
      const h8v slo = *reinterpret_cast<const h8v*>(st + (q * 4 + 3) * 1024);
      re[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(chi, ahi, re[q], 0, 0, 0);
      im[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(shi, dhi, im[q], 0, 0, 0);
      re[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(chi, alo, re[q], 0, 0, 0);
      im[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(shi, dlo, im[q], 0, 0, 0);
      re[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(clo, ahi, re[q], 0, 0, 0);
      im[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(slo, dhi, im[q], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // next stage landed
    __syncthreads();                                                          // ... and everybody is done with this one
  }
  // ---- |X|^2 -> mel: accumulator registers 8 s2 .. 8 s2 + 7 are the B fragment of k half s2 (rows = bins = k).
  // The mel weights come straight from global memory (84 KB, L2-resident); the six fragments of step it + 1 are requested in front of
  // step it's MFMAs (round 3 stamps: with the loads issued where they were used, pass 0's mel stage cost 20.9 k cycles, a dependent
  // L2 round trip per fragment, against 3.6 k for pass 1, whose lines the first pass had pulled in)
  if (ZERO_MEL) {                                                             // (a wave that runs ONE pass: the mel accumulators need not live through the k loop)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mel[t][r] = 0.f;
  }
  bf8v wcur[6], wnxt[6];
  auto wload = [&](bf8v (&w)[6], int it) {
    const __bf16* wq = p.melw + ((size_t)(Q0 + it / 2) * 2 * 3 * 2 * 64 + lane) * 8;
    const int s2 = it & 1;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      w[2 * t] = *reinterpret_cast<const bf8v*>(wq + ((size_t)(s2 * 3 + t) * 2 + 0) * 64 * 8);
      w[2 * t + 1] = *reinterpret_cast<const bf8v*>(wq + ((size_t)(s2 * 3 + t) * 2 + 1) * 64 * 8);
    }
  };
  wload(wcur, 0);
#pragma unroll
  for (int it = 0; it < 2 * NTILE; ++it) {
    const int q = it / 2, s2 = it & 1;
    if (it + 1 < 2 * NTILE) wload(wnxt, it + 1);
    bf8v p1, p2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float pw = (re[q][8 * s2 + e] * re[q][8 * s2 + e] + im[q][8 * s2 + e] * im[q][8 * s2 + e]) * (1.0f / (SD_FB_SCALE * SD_FB_SCALE));
      p1[e] = (__bf16)pw;
      p2[e] = (__bf16)(pw - (float)p1[e]);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      mel[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wcur[2 * t], p1, mel[t], 0, 0, 0);
      mel[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wcur[2 * t], p2, mel[t], 0, 0, 0);
      mel[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wcur[2 * t + 1], p1, mel[t], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) wcur[i] = wnxt[i];
  }
}

__global__ __launch_bounds__(256 * V2_GROUPS, 1) void fbank_logmel_kernel(const Fbank2Args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = (tid >> 6) & 3;                     // which of the workgroup's four 32-frame tiles
  const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);   // which share of the tile's 7 bin tiles (waves w, w + 4, w + 8 sit on the same SIMD)
  float* xs = smem + wid * V2_XS_W;
  char* stage = reinterpret_cast<char*>(smem + WAVES * V2_XS_W) + (V2_GROUPS == 1 ? 0 : (grp == 0 ? fb_ring_off(0) : grp == 1 ? fb_ring_off(1) : fb_ring_off(2)));
  float* macc = smem + wid * FT * MELP;               // after the DFT: [frame][mel] staging in the (dead) sample image

  // ---- which frames does this wave own (all wave-uniform)
  const long BT = (long)p.B * p.T;
  const long tile = (long)blockIdx.x * WAVES + wid;
  int bA, tA, nfA, nfB;
  long rowA;
  if (p.flat) {
    const long g0 = tile * FT;
    const long remaining = BT - g0;
    bA = (int)(g0 / p.T);
    tA = (int)(g0 - (long)bA * p.T);
    int a = p.T - tA; if (a > FT) a = FT;
    if (remaining <= 0) a = 0; else if (a > remaining) a = (int)remaining;
    nfA = a;
    long bmax = remaining - a; if (bmax < 0) bmax = 0;
    nfB = FT - a; if (nfB > bmax) nfB = (int)bmax;
    if (remaining <= 0) nfB = 0;
    rowA = g0;
  } else {
    const int tps = (p.T + FT - 1) / FT;
    bA = (int)(tile / tps);
    tA = (int)(tile - (long)bA * tps) * FT;
    nfA = p.T - tA; if (nfA > FT) nfA = FT;
    if (bA >= p.B) nfA = 0;
    nfB = 0;
    rowA = (long)bA * p.T + tA;
  }
  const int nvalid = nfA + nfB;

  // ---- stage the sample spans: sample `rel` of a span at word off + rel + 4 (rel / 160)
  const int lenA = nfA > 0 ? (nfA - 1) * HOP + NFFT : 0;
  // second span: continue the frame pattern (frame nfA at 164 nfA mod 64 words) past the end of the first span
  int offB = lenA + 4 * (lenA / HOP) + 16;
  offB = ((offB + 63) & ~63) + ((164 * nfA) & 63);
  {
    // 32 independent loads in flight per lane (a span is ~84 per lane): the wave is alone on its SIMD and the samples come
    // from HBM, so every batch costs a full memory round trip
    constexpr int SB = 32;
    auto stage_span = [&](int b, int s0, int len, int off) {
      const long long start = p.starts ? p.starts[b] : (long long)b * p.n;      // first sample of row b in `wav`
      bool bad = false;                                                         // a NaN sample: the clamp below would launder it into -16
      for (int rel0 = lane + grp * 64 * SB; rel0 < len; rel0 += V2_GROUPS * 64 * SB) {     // (the pair stages alternate batches)
        float v[SB];
        bool okv[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {                // unconditional loads from a clamped index (a predicated load becomes a
          const int rel = rel0 + 64 * u;              // branch and a wait of its own), zero padding selected afterwards
          int sidx = s0 + rel;
          bool ok = true;
          if (p.pad_mode == SD_PAD_REFLECT) {
            sidx = sidx < 0 ? -sidx : sidx;
            sidx = sidx >= p.n ? 2 * (p.n - 1) - sidx : sidx;
          } else {
            ok = sidx >= 0 && sidx < p.n;
          }
          sidx = sidx < 0 ? 0 : (sidx >= p.n ? p.n - 1 : sidx);
          long long gi = start + sidx;                // a window may hang over either end of the signal: zeros there
          ok = ok && gi >= 0 && gi < p.n_total;
          gi = gi < 0 ? 0 : (gi >= p.n_total ? p.n_total - 1 : gi);
          okv[u] = ok;
          v[u] = p.wav[gi];
        }
        // zero padding, and saturation at +-SD_FB_XMAX: the folded sums x[k] +- x[400-k], scaled by 2^10, must stay inside
        // the f16 range (|x| <= 16 is exact; beyond it the sample is clipped instead of turning the segment into NaNs)
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          bad |= okv[u] && v[u] != v[u] && rel0 + 64 * u < len;
          v[u] = okv[u] ? __builtin_amdgcn_fmed3f(v[u], -SD_FB_XMAX, SD_FB_XMAX) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          const int rel = rel0 + 64 * u;
          if (rel < len) xs[off + rel + 4 * (rel / HOP)] = v[u];
        }
      }
      // the reference's arithmetic would carry the NaN into the utterance's floor and mean, i.e. into every value of its features:
      // the utterance is marked (the largest key) and the finalize pass writes NaN rows for it
      if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicMax(p.maxbuf + b, SD_FB_POISON_KEY);
    };
    if (nfA > 0) stage_span(bA, tA * HOP - NFFT / 2, lenA, 0);
    if (nfB > 0) stage_span(bA + 1, -NFFT / 2, (nfB - 1) * HOP + NFFT, offB);
  }
  const int j = lane & 31, h = lane >> 5;
  int base;                                           // word of this lane's frame start
  if (j < nfA) base = 164 * j;
  else if (j < nvalid) base = offB + 164 * (j - nfA);
  else base = 0;                                      // idle lane: reads something finite-or-not, never stored
  // (every word a STORED frame reads lies inside its own 400 staged samples, except x[400], which is forced to 0)
  __syncthreads();

  f32x16 mel[3];
#if SD_FB_GROUPS > 1
  // every share of the bin tiles at once: the waves of a tile sit on the same SIMD, so one's fragment building, LDS reads and barrier
  // waits run under the others' MFMAs (stamps of the sequential form: a k step took 2100-2900 cycles for 576-768 of matrix work).
  // Same number of barriers in every branch (1 + 13 each).
  if (grp == 0) fbank2_pass<fb_nt(0), fb_q0(0), 0, true>(p, xs, base, stage, lane, wid, mel);
  else if (grp == 1) fbank2_pass<fb_nt(1), fb_q0(1), 1, true>(p, xs, base, stage, lane, wid, mel);
#if SD_FB_GROUPS > 2
  else fbank2_pass<fb_nt(2), fb_q0(2), 2, true>(p, xs, base, stage, lane, wid, mel);
#endif
  __syncthreads();                                    // every wave is done with its sample image
  // the partial mel sums of the tile's waves meet in LDS [frame][mel], in group order (deterministic)
#pragma unroll 1
  for (int g = 0; g < V2_GROUPS; ++g) {
    if (grp == g) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < p.n_mels) macc[j * MELP + m] = g == 0 ? mel[t][r] : macc[j * MELP + m] + mel[t][r];
        }
    }
    __syncthreads();
  }
#else
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) mel[t][r] = 0.f;
  fbank2_pass<4, 0, 0, false>(p, xs, base, stage, lane, wid, mel);
  fbank2_pass<3, 4, 1, false>(p, xs, base, stage, lane, wid, mel);
  __syncthreads();                                    // every wave is done with its sample image

  // mel tile rows -> LDS [frame][mel] for the log / max / coalesced-store passes below
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m < p.n_mels) macc[j * MELP + m] = mel[t][r];
    }
  __syncthreads();
#endif
  {
    // a frame's mels are shared by the lane pair (j, h) and, in the paired build, by the two waves of the tile
    const int mh = (p.n_mels + 2 * V2_GROUPS - 1) / (2 * V2_GROUPS);
    const int m_lo = (grp * 2 + h) * mh < p.n_mels ? (grp * 2 + h) * mh : p.n_mels;
    const int m_hi = (m_lo + mh < p.n_mels) ? m_lo + mh : p.n_mels;
    float vmax = -INFINITY;
    float* mrow = macc + j * MELP;
    // v_log_f32 (log2, 1 ulp; arguments >= eps > 0 are normal numbers) times ln 2 or 10 log10(2): 2e-6 absolute on a log2 of
    // up to +-33, against a tolerance of 2e-4 (ln) / 1e-3 (dB); the library logf / log10f are ~30 instructions each
    // (in-kernel stamps, round 3: log + LDS pass 13.8 k -> 9.7 k cycles per wave tile, store loop 7.0 k -> 4.5 k of ~110 k)
    const bool ln = p.log_mode == SD_LOG_LN_EPS;
    const float lscale = ln ? 0.6931471805599453f : 3.0102999566398120f;
    for (int m = m_lo; m < m_hi; ++m) {
      const float v = mrow[m];
      const float lv = lscale * __builtin_amdgcn_logf(ln ? v + p.log_eps : fmaxf(v, p.log_eps));
      mrow[m] = lv;
      vmax = fmaxf(vmax, lv);
    }
    const float mA = sd_wave_max(j < nfA ? vmax : -INFINITY);
    const float mB = sd_wave_max((j >= nfA && j < nvalid) ? vmax : -INFINITY);
    if (lane == 0) {
      if (nfA > 0) atomicMax(p.maxbuf + bA, f32_key(mA));
      if (nfB > 0) atomicMax(p.maxbuf + bA + 1, f32_key(mB));
    }
  }
  __syncthreads();
  {
    const int total = nvalid * p.n_mels;
    float* const orow = p.out + (size_t)rowA * p.ld_out;
    for (int e = lane + 64 * grp; e < total; e += 64 * V2_GROUPS) {
      const int jj = (int)__umulhi((unsigned)e, p.inv_mels);                  // e / n_mels (exact for e < 2^16: inv_mels = ceil(2^32 / n_mels))
      const int m = e - jj * p.n_mels;
      orow[(size_t)jj * p.ld_out + m] = macc[jj * MELP + m];
    }
  }
}

__global__ void fill_i32_kernel(int* p, int n, int v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// top_db floor (relative to the utterance max) and per-bin mean removal over T.
// One workgroup per utterance; thread (r, c) walks rows r, r+R, ... of column c.
__global__ void fbank_finalize_kernel(float* out, int ld_out, int T, int n_mels, const int* maxbuf,
                                      int use_floor, float top_db, int mean_norm, int R) {
  extern __shared__ float part[];  // [R][n_mels]
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int c = tid % n_mels;
  const int r = tid / n_mels;
  const bool active = r < R;
  float* base = out + (size_t)b * T * ld_out;
  const int key = maxbuf[b];
  if (key == SD_FB_POISON_KEY) {                      // a NaN sample somewhere in the utterance: NaN features, as the reference's floor / mean would give
    if (active)
      for (int t = r; t < T; t += R) base[(size_t)t * ld_out + c] = __int_as_float(0x7FC00000);
    return;
  }
  if (!use_floor && !mean_norm) return;               // (raw features: the pass only exists to mark NaN utterances)
  const float thr = use_floor ? key_f32(key) - top_db : -INFINITY;
  float s = 0.f;
  if (active && mean_norm)
    for (int t = r; t < T; t += R) s += fmaxf(base[(size_t)t * ld_out + c], thr);
  if (active) part[r * n_mels + c] = s;
  __syncthreads();
  float mean = 0.f;
  if (mean_norm) {
    for (int k = 0; k < R; ++k) mean += part[k * n_mels + c];
    mean /= (float)T;
  }
  if (active)
    for (int t = r; t < T; t += R) {
      float* q = base + (size_t)t * ld_out + c;
      *q = fmaxf(*q, thr) - mean;
    }
}

}  // namespace

namespace {
// round-to-nearest-even f32 -> bf16 bits (finite inputs)
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
}  // namespace

extern "C" sd_fbank_plan* sd_fbank_plan_create(const float* window, int n_fft, int hop,
                                               const float* mel_fb, int n_mels,
                                               int pad_mode, int log_mode, float log_eps, float top_db) {
  auto fail = [](int code, const char* msg) -> sd_fbank_plan* { sd_set_error(code, "%s", msg); return nullptr; };
  if (!window || !mel_fb) return fail(SD_ERR_ARG, "sd_fbank_plan_create: null window/mel_fb");
  if (pad_mode != SD_PAD_ZERO && pad_mode != SD_PAD_REFLECT) return fail(SD_ERR_ARG, "sd_fbank_plan_create: bad pad_mode");
  if (log_mode != SD_LOG_LN_EPS && log_mode != SD_LOG_DB_TOPDB) return fail(SD_ERR_ARG, "sd_fbank_plan_create: bad log_mode");
  if (n_fft != NFFT || hop != HOP || n_mels > MAX_MELS) {
    // any other framing (fbank_batch at sr != 16 kHz, [REF speech_encode.py:14-24]): the DFT and the mel product as implicit GEMMs on
    // the exact-f32 conv operator (sd_fbank_generic.hip); any window, no symmetry needed
    if (!sd_fbank_generic_geometry_ok(n_fft, hop, n_mels))
      return fail(SD_ERR_UNSUPPORTED, "sd_fbank_plan_create: need 8 <= n_fft <= 8192, 1 <= hop <= n_fft, 1 <= n_mels <= 256");
    sd_fbank_plan* plan = new sd_fbank_plan{};
    plan->n_fft = n_fft; plan->hop = hop; plan->n_mels = n_mels; plan->pad_mode = pad_mode; plan->log_mode = log_mode;
    plan->log_eps = log_eps; plan->top_db = top_db; plan->generic = true;
    if (sd_fbank_generic_create_tables(plan, window, mel_fb) != SD_OK) { delete plan; return nullptr; }
    return plan;
  }
  if (n_mels < 1) return fail(SD_ERR_UNSUPPORTED, "sd_fbank_plan_create: n_mels must be positive");
  for (int k = 1; k < NFFT / 2; ++k)
    if (window[k] != window[NFFT - k])
      return fail(SD_ERR_UNSUPPORTED, "sd_fbank_plan_create: window must be symmetric (w[k] == w[n_fft-k])");

  // ---- tables of the split-f16 kernel.  Basis: value = w[k] cos / sin(2 pi k bin / 400) in float64, hi = f16(v),
  // lo = f16(v - hi); fragment order of v_mfma_f32_32x32x16_f16's A operand: lane l holds row (bin) l & 31,
  // k = 16 s + 8 (l >> 5) + e.  One (pass, k step) block is contiguous: [tile in pass][kind][lane][e].
  std::vector<_Float16> b16(V2_BASIS_BYTES / 2);
  {
    size_t o = 0;
    for (int pass = 0; pass < V2_NPASS; ++pass) {
      const int q0 = fb_q0(pass), nt = fb_nt(pass);
      for (int st = 0; st < V2_KSTEPS; ++st)
        for (int qi = 0; qi < nt; ++qi)
          for (int kind = 0; kind < 4; ++kind)
            for (int l = 0; l < 64; ++l)
              for (int e = 0; e < 8; ++e, ++o) {
                const int bin = (q0 + qi) * 32 + (l & 31), k = 16 * st + 8 * (l >> 5) + e;
                double v = 0.0;
                if (bin < NFREQ && k <= NFFT / 2) {
                  const int ph = (int)(((long)k * bin) % NFFT);
                  const double ang = 2.0 * M_PI * (double)ph / (double)NFFT;
                  v = (double)window[k] * (kind < 2 ? std::cos(ang) : std::sin(ang));
                  if (k == NFFT / 2) v = kind < 2 ? 0.5 * v : 0.0;
                  if (k == 0 && kind >= 2) v = 0.0;
                }
                const _Float16 hi = (_Float16)(float)v;
                b16[o] = (kind & 1) ? (_Float16)(float)(v - (double)(float)hi) : hi;
              }
    }
  }
  // Mel weights as the A operand of v_mfma_f32_32x32x16_bf16 against the DFT accumulators: lane l holds row (mel)
  // 32 t + (l & 31), element e <-> accumulator register 8 s2 + e of lane half l >> 5 = bin 32 q + 16 s2 + 8 (e >> 2) + 4 (l >> 5) + (e & 3)
  std::vector<unsigned short> m16(V2_MELW_BYTES / 2);
  {
    size_t o = 0;
    for (int q = 0; q < NBT; ++q)
      for (int s2 = 0; s2 < 2; ++s2)
        for (int t = 0; t < 3; ++t)
          for (int part = 0; part < 2; ++part)
            for (int l = 0; l < 64; ++l)
              for (int e = 0; e < 8; ++e, ++o) {
                const int m = 32 * t + (l & 31), bin = 32 * q + 16 * s2 + 8 * (e >> 2) + 4 * (l >> 5) + (e & 3);
                const float w = (m < n_mels && bin < NFREQ) ? mel_fb[(size_t)bin * n_mels + m] : 0.f;
                const unsigned short w1 = bf16_bits(w);
                m16[o] = part == 0 ? w1 : bf16_bits(w - bf16_value(w1));
              }
  }
  sd_fbank_plan* plan = new sd_fbank_plan{};
  plan->n_fft = n_fft; plan->hop = hop; plan->n_mels = n_mels; plan->pad_mode = pad_mode; plan->log_mode = log_mode;
  plan->log_eps = log_eps; plan->top_db = top_db;
  hipError_t e1 = hipMalloc(&plan->basis16_dev, V2_BASIS_BYTES);
  hipError_t e2 = e1 == hipSuccess ? hipMalloc(&plan->melw16_dev, V2_MELW_BYTES) : e1;
  if (e1 == hipSuccess && e2 == hipSuccess) {
    e1 = hipMemcpy(plan->basis16_dev, b16.data(), V2_BASIS_BYTES, hipMemcpyHostToDevice);
    e2 = hipMemcpy(plan->melw16_dev, m16.data(), V2_MELW_BYTES, hipMemcpyHostToDevice);
  }
  if (e1 != hipSuccess || e2 != hipSuccess) {
    sd_set_error(SD_ERR_HIP, "sd_fbank_plan_create: device table upload failed: %s",
                 hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    if (plan->basis16_dev) (void)hipFree(plan->basis16_dev);
    if (plan->melw16_dev) (void)hipFree(plan->melw16_dev);
    delete plan;
    return nullptr;
  }
  if (sd_fbank_utt16_create_tables(plan, window, mel_fb) != SD_OK) {      // tables of the one-launch kernel (sd_fbank_utt16.hip)
    (void)hipFree(plan->basis16_dev);
    (void)hipFree(plan->melw16_dev);
    delete plan;
    return nullptr;
  }
  return plan;
}

extern "C" void sd_fbank_plan_destroy(sd_fbank_plan* plan) {
  if (!plan) return;
  if (plan->generic) { sd_fbank_generic_destroy_tables(plan); delete plan; return; }
  (void)hipFree(plan->basis16_dev);
  (void)hipFree(plan->melw16_dev);
  sd_fbank_utt16_destroy_tables(plan);
  delete plan;
}

extern "C" int sd_fbank_num_frames(const sd_fbank_plan* plan, int n) {
  if (!plan || n < 0) return sd_set_error(SD_ERR_ARG, "sd_fbank_num_frames: bad arguments");
  if (plan->generic) return sd_fbank_generic_num_frames(plan, n);
  return 1 + n / plan->hop;
}

extern "C" size_t sd_fbank_workspace_bytes(const sd_fbank_plan* plan, int B, int n) {
  if (plan && plan->generic) return sd_fbank_generic_workspace_bytes(plan, B, n);
  return ((size_t)(B > 0 ? B : 0) * sizeof(int) + 255) & ~(size_t)255;
}

static int fbank_launch(const sd_fbank_plan* plan, const float* wav_dev, long long n_total, const long long* starts_dev, int B, int n,
                        int mean_norm, float* out_dev, int ld_out, void* ws_dev, size_t ws_bytes, sd_stream_t stream_);

extern "C" int sd_fbank_f32(const sd_fbank_plan* plan, const float* wav_dev, int B, int n,
                            int mean_norm, float* out_dev, int ld_out,
                            void* ws_dev, size_t ws_bytes, sd_stream_t stream_) {
  return fbank_launch(plan, wav_dev, (long long)B * n, nullptr, B, n, mean_norm, out_dev, ld_out, ws_dev, ws_bytes, stream_);
}

extern "C" int sd_fbank_windows_f32(const sd_fbank_plan* plan, const float* wav_dev, long long n_total, const long long* starts_dev,
                                    int B, int n, int mean_norm, float* out_dev, int ld_out,
                                    void* ws_dev, size_t ws_bytes, sd_stream_t stream_) {
  SD_CHECK_ARG(n_total >= 1, "sd_fbank_windows_f32: n_total=%lld", n_total);
  SD_CHECK_ARG(B == 0 || starts_dev != nullptr, "sd_fbank_windows_f32: null starts");
  return fbank_launch(plan, wav_dev, n_total, starts_dev, B, n, mean_norm, out_dev, ld_out, ws_dev, ws_bytes, stream_);
}

static int fbank_launch(const sd_fbank_plan* plan, const float* wav_dev, long long n_total, const long long* starts_dev, int B, int n,
                        int mean_norm, float* out_dev, int ld_out, void* ws_dev, size_t ws_bytes, sd_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SD_CHECK_ARG(plan != nullptr, "sd_fbank_f32: null plan");
  SD_CHECK_ARG(B >= 0 && n >= 0, "sd_fbank_f32: B=%d n=%d", B, n);
  if (B == 0) return SD_OK;
  SD_CHECK_ARG(wav_dev && out_dev, "sd_fbank_f32: null wav/out");
  SD_CHECK_ARG(ld_out >= plan->n_mels, "sd_fbank_f32: ld_out=%d < n_mels=%d", ld_out, plan->n_mels);
  // torch.stft(center=True, pad_mode="reflect") rejects n <= n_fft/2; zero padding needs one sample
  if (plan->pad_mode == SD_PAD_REFLECT)
    SD_CHECK_ARG(n > plan->n_fft / 2, "sd_fbank_f32: reflect padding of %d needs more than %d samples (got %d)", plan->n_fft / 2, plan->n_fft / 2, n);
  else
    SD_CHECK_ARG(n >= 1, "sd_fbank_f32: empty waveform");
  if (plan->generic) return sd_fbank_generic_launch(plan, wav_dev, n_total, starts_dev, B, n, mean_norm, out_dev, ld_out, ws_dev, ws_bytes, stream);
  SD_CHECK_ARG(ws_dev != nullptr && ws_bytes >= sd_fbank_workspace_bytes(plan, B, n),
               "sd_fbank_f32: workspace too small (%zu < %zu)", ws_bytes, sd_fbank_workspace_bytes(plan, B, n));
  // utterances whose padded signal fits the CU's LDS: ONE launch, one workgroup per utterance (sd_fbank_utt16.hip)
  if (n_total >= 4 && sd_fbank_utt16_supported(plan, n))    // (16-byte loads: four samples behind `wav`)
    return sd_fbank_utt16_launch(plan, wav_dev, n_total, starts_dev, B, n, mean_norm, out_dev, ld_out, stream);
  const int T = 1 + n / HOP;
  Fbank2Args a;
  a.wav = wav_dev; a.B = B; a.n = n; a.T = T;
  a.starts = starts_dev; a.n_total = n_total;
  a.basis = static_cast<const _Float16*>(plan->basis16_dev); a.melw = static_cast<const __bf16*>(plan->melw16_dev);
  a.n_mels = plan->n_mels; a.pad_mode = plan->pad_mode; a.log_mode = plan->log_mode; a.log_eps = plan->log_eps;
  a.out = out_dev; a.ld_out = ld_out;
  a.maxbuf = static_cast<int*>(ws_dev);
  a.flat = T >= FT;
  a.inv_mels = (unsigned)((((unsigned long long)1 << 32) + plan->n_mels - 1) / plan->n_mels);
  const long tiles = a.flat ? ((long)B * T + FT - 1) / FT : (long)B * ((T + FT - 1) / FT);
  const long blocks = (tiles + WAVES - 1) / WAVES;
  SD_CHECK_ARG(blocks < (1L << 31), "sd_fbank_f32: grid too large");
  // reset the per-utterance max keys with a kernel, not hipMemsetAsync: a byte-pattern memset node
  // replayed from a captured hipGraph did not reproduce the eager result (configs[3] test)
  hipLaunchKernelGGL(fill_i32_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, a.maxbuf, B, (int)0x80808080);
  SD_CHECK_LAUNCH("fill_i32_kernel");
  SD_CHECK_HIP(sd_func_max_lds(reinterpret_cast<const void*>(fbank_logmel_kernel), V2_LDS_BYTES));
  {
    // algorithmic bytes: waveform read once + log-mel written once (192 320 B per 2 s segment)
    SdProfScope prof(SD_PROF_FBANK, stream, (double)B * ((double)n * 4.0 + (double)T * plan->n_mels * 4.0));
    hipLaunchKernelGGL(fbank_logmel_kernel, dim3((unsigned)blocks), dim3(256 * V2_GROUPS), V2_LDS_BYTES, stream, a);
  }
  SD_CHECK_LAUNCH("fbank_logmel_kernel");
  const int use_floor = plan->log_mode == SD_LOG_DB_TOPDB && plan->top_db >= 0.f;
  {
    int R = 320 / plan->n_mels; if (R < 1) R = 1; if (R > T) R = T;
    int threads = ((R * plan->n_mels + 63) / 64) * 64;
    hipLaunchKernelGGL(fbank_finalize_kernel, dim3((unsigned)B), dim3(threads), (size_t)R * plan->n_mels * sizeof(float),
                       stream, out_dev, ld_out, T, plan->n_mels, a.maxbuf, use_floor, plan->top_db, mean_norm, R);
    SD_CHECK_LAUNCH("fbank_finalize_kernel");
  }
  return SD_OK;
}
