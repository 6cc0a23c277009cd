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
// (x[400] := 0 for k = 0, Cb[200] halved), i.e. two GEMMs  [bins x K] * [K x frames]
// that run on the f32 matrix cores (v_mfma_f32_32x32x2_f32, exact f32 fma chain).
// The basis is a host-computed table; a wave owns 32 consecutive frames, keeps their
// sample span in LDS (each sample is fetched from HBM once per tile, 4.5 % overlap),
// forms the folded operands on the fly, and accumulates |X|^2 into the mel bins kept
// in LDS.  Tiles are flat over (utterance, frame) so no lane idles on T = 201.
//
// LDS: sample spans are skewed by one word per 160 samples so that the 32 frames of
// a tile (stride 160 words = 0 mod 32 banks) hit 32 distinct banks.
#include "sd_common.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int NFFT = 400;
constexpr int HOP = 160;
constexpr int NFREQ = 201;
constexpr int KP = 204;      // folded K, padded
constexpr int KC = 34;       // K per staged basis chunk
constexpr int NCHUNK = KP / KC;  // 6
constexpr int NBT = 7;       // 32-bin tiles (224 >= 201)
constexpr int FT = 32;       // frames per wave tile
constexpr int WAVES = 4;
constexpr int XS_W = 5760;   // per-wave sample LDS, floats
constexpr int MELP = 81;     // mel accumulator row stride
constexpr int MAX_MELS = 80;
constexpr int CHUNK_FLOATS = KC * 64;  // [KC][cos 32 | sin 32]
constexpr int LDS_FLOATS = WAVES * XS_W + WAVES * FT * MELP + 2 * CHUNK_FLOATS;

struct MelEntry { int i0, i1; float w0, w1; };

struct FbankArgs {
  const float* wav; int B; int n; int T;
  const float* basis;      // [NBT][KP][64]
  const MelEntry* mel_tab; // [NBT*32]
  int n_mels; int pad_mode; int log_mode; float log_eps;
  float* out; int ld_out;
  int* maxbuf;             // [B] ordered-int keys of the utterance max
  int flat;                // tiles flat over B*T (T >= 32) or one tile row per utterance
};

__device__ __forceinline__ int f32_key(float v) {
  const int b = __float_as_int(v);
  return b >= 0 ? b : b ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float key_f32(int k) {
  return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF);
}

__global__ __launch_bounds__(256, 1) void fbank_logmel_kernel(const FbankArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float* xs = smem + wid * XS_W;
  float* macc = smem + WAVES * XS_W + wid * FT * MELP;
  float* bs = smem + WAVES * XS_W + WAVES * FT * MELP;

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

  // ---- stage sample spans (skewed) into this wave's LDS
  const int lenA = nfA > 0 ? (nfA - 1) * HOP + NFFT : 0;
  int offB = lenA + lenA / HOP + 1;
  offB = ((offB + 31) & ~31) + (nfA & 31);
  {
    const float* src = p.wav + (size_t)bA * p.n;
    const int s0 = tA * HOP - NFFT / 2;
    for (int rel = lane; rel < lenA; rel += 64) {
      int s = s0 + rel;
      float v;
      if (p.pad_mode == SD_PAD_REFLECT) {
        s = s < 0 ? -s : s;
        s = s >= p.n ? 2 * (p.n - 1) - s : s;
        v = src[s];
      } else {
        v = (s >= 0 && s < p.n) ? src[s] : 0.f;
      }
      xs[rel + rel / HOP] = v;
    }
    if (nfB > 0) {
      const float* srcB = p.wav + (size_t)(bA + 1) * p.n;
      const int lenB = (nfB - 1) * HOP + NFFT;
      for (int rel = lane; rel < lenB; rel += 64) {
        int s = rel - NFFT / 2;
        float v;
        if (p.pad_mode == SD_PAD_REFLECT) {
          s = s < 0 ? -s : s;
          s = s >= p.n ? 2 * (p.n - 1) - s : s;
          v = srcB[s];
        } else {
          v = (s >= 0 && s < p.n) ? srcB[s] : 0.f;
        }
        xs[offB + rel + rel / HOP] = v;
      }
    }
  }
  for (int i = lane; i < FT * MELP; i += 64) macc[i] = 0.f;

  const int j = lane & 31;   // frame within tile (MFMA column)
  const int h = lane >> 5;   // k slot / row half
  int base;                  // skewed LDS word of this lane's frame start
  if (j < nfA) base = 161 * j;
  else if (j < nvalid) base = offB + 161 * (j - nfA);
  else base = 0;             // idle lane: reads something harmless, never stored

  // ---- basis chunk pipeline (all 4 waves share the staged chunk)
  f32x4 pre[3];
  auto bload = [&](int ch) {
    const f32x4* g = reinterpret_cast<const f32x4*>(p.basis + (size_t)ch * CHUNK_FLOATS);
    pre[0] = g[tid];
    pre[1] = g[tid + 256];
    if (tid < CHUNK_FLOATS / 4 - 512) pre[2] = g[tid + 512];
  };
  auto bstore = [&](int buf) {
    f32x4* d = reinterpret_cast<f32x4*>(bs + buf * CHUNK_FLOATS);
    d[tid] = pre[0];
    d[tid + 256] = pre[1];
    if (tid < CHUNK_FLOATS / 4 - 512) d[tid + 512] = pre[2];
  };

  bload(0);
  bstore(0);
  __syncthreads();
  int cur = 0;

  for (int q = 0; q < NBT; ++q) {
    f32x16 accRe, accIm;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accRe[r] = 0.f; accIm[r] = 0.f; }

    for (int c = 0; c < NCHUNK; ++c) {
      const int ch = q * NCHUNK + c;
      const bool more = ch + 1 < NBT * NCHUNK;
      if (more) bload(ch + 1);
      const float* bcur = bs + cur * CHUNK_FLOATS + h * 64 + j;
      const int kbase = c * KC + h;
#pragma unroll
      for (int kk = 0; kk < KC / 2; ++kk) {
        const int k = kbase + 2 * kk;
        const int kb = NFFT - k;
        const int ka_s = k + (k >= HOP) + (k >= 2 * HOP);
        const int kb_s = kb + (kb >= HOP) + (kb >= 2 * HOP);
        const float xa = xs[base + ka_s];
        float xb = xs[base + kb_s];
        xb = (k == 0) ? 0.f : xb;
        const float cb = bcur[kk * 128];
        const float sb = bcur[kk * 128 + 32];
        accRe = __builtin_amdgcn_mfma_f32_32x32x2f32(cb, xa + xb, accRe, 0, 0, 0);
        accIm = __builtin_amdgcn_mfma_f32_32x32x2f32(sb, xa - xb, accIm, 0, 0, 0);
      }
      if (more) bstore(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }

    // ---- |X|^2 -> mel bins. Lane holds bins 32q + (r&3) + 8(r>>2) + 4h of frame j.
    // The two lane halves own different bins of the same frame and may hit the same
    // filter, so they take turns (fixed order, deterministic).
    float* mrow = macc + j * MELP;
#pragma unroll
    for (int turn = 0; turn < 2; ++turn) {
      if (h == turn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int bin = 32 * q + (r & 3) + 8 * (r >> 2) + 4 * h;
          const MelEntry e = p.mel_tab[bin];
          const float pw = accRe[r] * accRe[r] + accIm[r] * accIm[r];
          if (e.i0 >= 0) mrow[e.i0] += e.w0 * pw;
          if (e.i1 >= 0) mrow[e.i1] += e.w1 * pw;
        }
      }
      __builtin_amdgcn_wave_barrier();  // keep the two turns' LDS updates in program order
    }
  }
  __syncthreads();

  // ---- log + utterance max; lane (j, h) takes half of the mel bins of frame j
  {
    const int mh = (p.n_mels + 1) / 2;
    const int m_lo = h * mh;
    const int m_hi = (m_lo + mh < p.n_mels) ? m_lo + mh : p.n_mels;
    float vmax = -INFINITY;
    float* mrow = macc + j * MELP;
    for (int m = m_lo; m < m_hi; ++m) {
      const float v = mrow[m];
      const float lv = (p.log_mode == SD_LOG_LN_EPS) ? logf(v + p.log_eps)
                                                      : 10.0f * log10f(fmaxf(v, p.log_eps));
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
  // ---- coalesced store of the tile's [nvalid][n_mels] block
  {
    const int total = nvalid * p.n_mels;
    for (int e = lane; e < total; e += 64) {
      const int jj = e / p.n_mels;
      const int m = e - jj * p.n_mels;
      p.out[(size_t)(rowA + jj) * p.ld_out + m] = macc[jj * MELP + m];
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
  const float thr = use_floor ? key_f32(maxbuf[b]) - top_db : -INFINITY;
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

struct sd_fbank_plan {
  int n_fft, hop, n_mels, pad_mode, log_mode;
  float log_eps, top_db;
  float* basis_dev;
  MelEntry* mel_dev;
};

extern "C" sd_fbank_plan* sd_fbank_plan_create(const float* window, int n_fft, int hop,
                                               const float* mel_fb, int n_mels,
                                               int pad_mode, int log_mode, float log_eps, float top_db) {
  auto fail = [](int code, const char* msg) -> sd_fbank_plan* { sd_set_error(code, "%s", msg); return nullptr; };
  if (!window || !mel_fb) return fail(SD_ERR_ARG, "sd_fbank_plan_create: null window/mel_fb");
  if (n_fft != NFFT || hop != HOP)
    return fail(SD_ERR_UNSUPPORTED, "sd_fbank_plan_create: only n_fft=400, hop=160 (25 ms / 10 ms at 16 kHz) is implemented");
  if (n_mels < 1 || n_mels > MAX_MELS) return fail(SD_ERR_UNSUPPORTED, "sd_fbank_plan_create: n_mels must be in [1, 80]");
  if (pad_mode != SD_PAD_ZERO && pad_mode != SD_PAD_REFLECT) return fail(SD_ERR_ARG, "sd_fbank_plan_create: bad pad_mode");
  if (log_mode != SD_LOG_LN_EPS && log_mode != SD_LOG_DB_TOPDB) return fail(SD_ERR_ARG, "sd_fbank_plan_create: bad log_mode");
  for (int k = 1; k < NFFT / 2; ++k)
    if (window[k] != window[NFFT - k])
      return fail(SD_ERR_UNSUPPORTED, "sd_fbank_plan_create: window must be symmetric (w[k] == w[n_fft-k])");

  std::vector<float> basis((size_t)NBT * KP * 64, 0.f);
  for (int q = 0; q < NBT; ++q)
    for (int k = 0; k <= NFFT / 2; ++k)
      for (int i = 0; i < 32; ++i) {
        const int bin = q * 32 + i;
        if (bin >= NFREQ) continue;
        // reduce k*bin mod 400 in integers so the angle stays small and exact
        const int ph = (int)(((long)k * bin) % NFFT);
        const double ang = 2.0 * M_PI * (double)ph / (double)NFFT;
        double cw = (double)window[k] * std::cos(ang);
        double sw = (double)window[k] * std::sin(ang);
        if (k == NFFT / 2) { cw *= 0.5; sw = 0.0; }
        if (k == 0) sw = 0.0;
        float* row = basis.data() + ((size_t)q * KP + k) * 64;
        row[i] = (float)cw;
        row[32 + i] = (float)sw;
      }
  std::vector<MelEntry> tab(NBT * 32);
  for (int bin = 0; bin < NBT * 32; ++bin) {
    MelEntry e{-1, -1, 0.f, 0.f};
    if (bin < NFREQ) {
      int cnt = 0;
      for (int m = 0; m < n_mels; ++m) {
        const float w = mel_fb[(size_t)bin * n_mels + m];
        if (w == 0.f) continue;
        if (cnt == 0) { e.i0 = m; e.w0 = w; }
        else if (cnt == 1) { e.i1 = m; e.w1 = w; }
        else return fail(SD_ERR_UNSUPPORTED, "sd_fbank_plan_create: more than 2 non-zero mel filters on one frequency bin");
        ++cnt;
      }
    }
    tab[bin] = e;
  }
  sd_fbank_plan* plan = new sd_fbank_plan{n_fft, hop, n_mels, pad_mode, log_mode, log_eps, top_db, nullptr, nullptr};
  hipError_t e1 = hipMalloc(&plan->basis_dev, basis.size() * sizeof(float));
  hipError_t e2 = e1 == hipSuccess ? hipMalloc(&plan->mel_dev, tab.size() * sizeof(MelEntry)) : e1;
  if (e1 == hipSuccess && e2 == hipSuccess) {
    e1 = hipMemcpy(plan->basis_dev, basis.data(), basis.size() * sizeof(float), hipMemcpyHostToDevice);
    e2 = hipMemcpy(plan->mel_dev, tab.data(), tab.size() * sizeof(MelEntry), hipMemcpyHostToDevice);
  }
  if (e1 != hipSuccess || e2 != hipSuccess) {
    sd_set_error(SD_ERR_HIP, "sd_fbank_plan_create: device table upload failed: %s",
                 hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    if (plan->basis_dev) (void)hipFree(plan->basis_dev);
    if (plan->mel_dev) (void)hipFree(plan->mel_dev);
    delete plan;
    return nullptr;
  }
  return plan;
}

extern "C" void sd_fbank_plan_destroy(sd_fbank_plan* plan) {
  if (!plan) return;
  (void)hipFree(plan->basis_dev);
  (void)hipFree(plan->mel_dev);
  delete plan;
}

extern "C" int sd_fbank_num_frames(const sd_fbank_plan* plan, int n) {
  if (!plan || n < 0) return sd_set_error(SD_ERR_ARG, "sd_fbank_num_frames: bad arguments");
  return 1 + n / plan->hop;
}

extern "C" size_t sd_fbank_workspace_bytes(const sd_fbank_plan*, int B, int) {
  return ((size_t)(B > 0 ? B : 0) * sizeof(int) + 255) & ~(size_t)255;
}

extern "C" int sd_fbank_f32(const sd_fbank_plan* plan, const float* wav_dev, int B, int n,
                            int mean_norm, float* out_dev, int ld_out,
                            void* ws_dev, size_t ws_bytes, sd_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SD_CHECK_ARG(plan != nullptr, "sd_fbank_f32: null plan");
  SD_CHECK_ARG(B >= 0 && n >= 0, "sd_fbank_f32: B=%d n=%d", B, n);
  if (B == 0) return SD_OK;
  SD_CHECK_ARG(wav_dev && out_dev, "sd_fbank_f32: null wav/out");
  SD_CHECK_ARG(ld_out >= plan->n_mels, "sd_fbank_f32: ld_out=%d < n_mels=%d", ld_out, plan->n_mels);
  // torch.stft(center=True, pad_mode="reflect") rejects n <= n_fft/2; zero padding needs one sample
  if (plan->pad_mode == SD_PAD_REFLECT)
    SD_CHECK_ARG(n > NFFT / 2, "sd_fbank_f32: reflect padding of %d needs more than %d samples (got %d)", NFFT / 2, NFFT / 2, n);
  else
    SD_CHECK_ARG(n >= 1, "sd_fbank_f32: empty waveform");
  SD_CHECK_ARG(ws_dev != nullptr && ws_bytes >= sd_fbank_workspace_bytes(plan, B, n),
               "sd_fbank_f32: workspace too small (%zu < %zu)", ws_bytes, sd_fbank_workspace_bytes(plan, B, n));
  const int T = 1 + n / HOP;
  FbankArgs a;
  a.wav = wav_dev; a.B = B; a.n = n; a.T = T;
  a.basis = plan->basis_dev; a.mel_tab = plan->mel_dev;
  a.n_mels = plan->n_mels; a.pad_mode = plan->pad_mode; a.log_mode = plan->log_mode; a.log_eps = plan->log_eps;
  a.out = out_dev; a.ld_out = ld_out;
  a.maxbuf = static_cast<int*>(ws_dev);
  a.flat = T >= FT;
  const long tiles = a.flat ? ((long)B * T + FT - 1) / FT : (long)B * ((T + FT - 1) / FT);
  const long blocks = (tiles + WAVES - 1) / WAVES;
  SD_CHECK_ARG(blocks < (1L << 31), "sd_fbank_f32: grid too large");
  // reset the per-utterance max keys with a kernel, not hipMemsetAsync: a byte-pattern memset node
  // replayed from a captured hipGraph did not reproduce the eager result (configs[3] test)
  hipLaunchKernelGGL(fill_i32_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, a.maxbuf, B, (int)0x80808080);
  SD_CHECK_LAUNCH("fill_i32_kernel");
  const size_t lds = (size_t)LDS_FLOATS * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    SD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fbank_logmel_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  {
    // algorithmic bytes: waveform read once + log-mel written once (192 320 B per 2 s segment)
    SdProfScope prof(SD_PROF_FBANK, stream, (double)B * ((double)n * 4.0 + (double)T * plan->n_mels * 4.0));
    hipLaunchKernelGGL(fbank_logmel_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, a);
  }
  SD_CHECK_LAUNCH("fbank_logmel_kernel");
  const int use_floor = plan->log_mode == SD_LOG_DB_TOPDB && plan->top_db >= 0.f;
  if (use_floor || mean_norm) {
    int R = 320 / plan->n_mels; if (R < 1) R = 1; if (R > T) R = T;
    int threads = ((R * plan->n_mels + 63) / 64) * 64;
    hipLaunchKernelGGL(fbank_finalize_kernel, dim3((unsigned)B), dim3(threads), (size_t)R * plan->n_mels * sizeof(float),
                       stream, out_dev, ld_out, T, plan->n_mels, a.maxbuf, use_floor, plan->top_db, mean_norm, R);
    SD_CHECK_LAUNCH("fbank_finalize_kernel");
  }
  return SD_OK;
}
