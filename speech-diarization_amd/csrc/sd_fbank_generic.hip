// Log-mel filterbank for ANY framing (n_fft, hop): `fbank_batch(wavs, sr)` with sr != 16000 [REF speech_encode.py:14-24]
// (win_length = n_fft = int(sr * 0.025), hop = int(sr * 0.010): 200 / 80 at 8 kHz, 551 / 220 at 22.05 kHz, 1200 / 480 at 48 kHz).
//
// The 16 kHz framing (400 / 160) has kernels of its own (sd_fbank_utt16.hip, sd_fbank.hip: factored / folded DFT on split f16).  Every
// other framing takes this path; speed is secondary here (the reference's pipeline resamples to 16 kHz before it embeds,
// [REF anti_stick_diarize.py:29-41]), accuracy is not:
//
//   1. fbg_pad_kernel       the padded signal of every utterance (center = True: n_fft / 2 samples of reflect or zero padding at both ends).
//   2. fbg_dft_f64_kernel   the windowed real DFT of every frame as a tiled [frames x n_fft] . [n_fft x (re | im)] product in FLOAT64 (v_fma_f64,
//                           64 x 64 tiles, 4 x 4 outputs per thread, operands through LDS), frames gathered from the padded signal in place;
//                           a thread owns re and im of a bin, so |X|^2 is formed in float64 and stored once as f32.
//                           (A first version ran this product on the exact-f32 conv operator: 2.2e-4 (ln) from the float64 oracle in bins
//                           50 dB below the frame's level -- the f32 accumulator of a 400-term sum whose partial sums reach ~1 while the
//                           result is ~1e-2 -- against 3e-5 for torch's f32 FFT.  In float64 the DFT adds nothing to the error budget.)
//   3. sd_conv1d_cl_f32     the mel product [n_freq] -> [n_mels] as a pointwise conv on the f32 matrix cores (a sum of non-negative terms:
//                           no cancellation; dense, any filter shapes).
//   4. fbg_finalize_kernel  log law, utterance maximum / top_db floor, mean over T, the write of out[b][t][m]; one workgroup per utterance.
//
// Utterances are processed in chunks so that the workspace stays below ~256 MB whatever B is.  A NaN sample propagates the way the
// reference's arithmetic does: its frames' bins are NaN, so is the utterance maximum (all features NaN under the top_db floor) and the
// mean over T (all NaN with mean removal).
#include <cmath>
#include <cstring>
#include <vector>

#include "sd_fbank_internal.h"

namespace {

constexpr int FBG_MAX_NFFT = 8192;
constexpr size_t FBG_CHUNK_BYTES = (size_t)256 << 20;

struct PadArgs {
  const float* wav; const long long* starts; long long n_total;
  float* xpad; int B0, Bc, n, Lp, pad, pad_mode;
};

// xpad[b][i] = padded signal at s = i - pad, i < Lp = n + 2 pad
__global__ void fbg_pad_kernel(PadArgs p) {
  const long long total = (long long)p.Bc * p.Lp;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(e / p.Lp);
    long long s = e - (long long)b * p.Lp - p.pad;
    if (p.pad_mode == SD_PAD_REFLECT) {
      s = s < 0 ? -s : s;
      s = s >= p.n ? 2LL * (p.n - 1) - s : s;
    }
    float v = 0.f;
    if (s >= 0 && s < p.n) {
      const long long start = p.starts ? p.starts[p.B0 + b] : (long long)(p.B0 + b) * p.n;
      const long long gi = start + s;
      if (gi >= 0 && gi < p.n_total) v = p.wav[gi];              // a window may hang over either end of the signal: zeros there
    }
    p.xpad[e] = v;
  }
}

struct DftArgs {
  const float* xpad; int Lp; int hop; int T; long long M;         // frame m = (b, t): n_fft samples at xpad[b * Lp + t * hop]
  const double* tw; int n_fft; int ncol;                          // tw [n_fft][ncol]: column 2 k = w[p] cos(2 pi k p / n_fft), 2 k + 1 = -w[p] sin
  float* pw; int nfp;                                             // pw [M][nfp] = re^2 + im^2 (bins past n_freq: zero)
};

constexpr int DT = 64, DK = 16;

__global__ __launch_bounds__(256) void fbg_dft_f64_kernel(DftArgs p) {
  __shared__ double As[DK][DT + 1];
  __shared__ double Bs[DK][DT];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const long long m0 = (long long)blockIdx.x * DT;
  const int c0 = blockIdx.y * DT;
  // the frame this thread stages (one of the tile's 64) and its 4 K positions; the table element it stages
  const int sf = tid & 63, sk = tid >> 6;
  const long long sm = m0 + sf;
  const float* srow = nullptr;
  if (sm < p.M) { const long long b = sm / p.T; srow = p.xpad + b * p.Lp + (sm - b * p.T) * (long long)p.hop; }
  double acc[4][4] = {};
  for (int k0 = 0; k0 < p.n_fft; k0 += DK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kk = sk * 4 + i, pos = k0 + kk;
      As[kk][sf] = (srow && pos < p.n_fft) ? (double)srow[pos] : 0.0;
      Bs[kk][sf] = pos < p.n_fft ? p.tw[(size_t)pos * p.ncol + c0 + sf] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < DK; ++kk) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][4 * ty + i]; b[i] = Bs[kk][4 * tx + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
  // columns 4 tx .. 4 tx + 3 of the tile = (re, im) of bins (c0 + 4 tx) / 2 and + 1
  const int bin = (c0 >> 1) + 2 * tx;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long m = m0 + 4 * ty + i;
    if (m >= p.M) continue;
    float* q = p.pw + m * p.nfp + bin;
    q[0] = (float)(acc[i][0] * acc[i][0] + acc[i][1] * acc[i][1]);
    q[1] = (float)(acc[i][2] * acc[i][2] + acc[i][3] * acc[i][3]);
  }
}

struct FinArgs {
  const float* mel; int ldm; int R; int row0;       // mel rows of utterance b: (b * R + row0 + t), t < T
  float* out; int ld_out; int B0; int T; int n_mels;
  int log_mode; float log_eps; float top_db; int use_floor; int mean_norm;
};

constexpr int FIN_THREADS = 256;

__global__ __launch_bounds__(FIN_THREADS) void fbg_finalize_kernel(FinArgs p) {
  __shared__ float red[FIN_THREADS];
  __shared__ float colsum[FIN_THREADS];
  __shared__ float mean_s[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* src = p.mel + ((size_t)b * p.R + p.row0) * p.ldm;
  float* dst = p.out + (size_t)(p.B0 + b) * p.T * p.ld_out;
  const int total = p.T * p.n_mels;
  const bool ln = p.log_mode == SD_LOG_LN_EPS;
  const float qnan = __int_as_float(0x7FC00000);
  // pass 1: the log law into the output rows, the utterance maximum, "a NaN was seen"
  float vmax = -INFINITY;
  bool bad = false;
  for (int e = tid; e < total; e += FIN_THREADS) {
    const int t = e / p.n_mels, m = e - t * p.n_mels;
    const float v = src[(size_t)t * p.ldm + m];
    const float lv = ln ? logf(v + p.log_eps) : 10.f * log10f(v != v ? v : fmaxf(v, p.log_eps));
    dst[(size_t)t * p.ld_out + m] = lv;
    vmax = fmaxf(vmax, lv);
    bad |= lv != lv;
  }
  red[tid] = bad ? qnan : vmax;
  __syncthreads();
  float thr = -INFINITY;
  if (p.use_floor) {
    float mx = -INFINITY;
    bool anybad = false;
    for (int i = 0; i < FIN_THREADS; ++i) { const float r = red[i]; anybad |= r != r; mx = fmaxf(mx, r); }
    thr = anybad ? qnan : mx - p.top_db;                      // torch: amax over the utterance carries a NaN into every floored value
  }
  if (!p.use_floor && !p.mean_norm) return;
  __syncthreads();
  // pass 2: per-bin mean over T of the floored values: RG row groups x n_mels columns, combined in a fixed order
  const int RG = p.n_mels <= FIN_THREADS ? FIN_THREADS / p.n_mels : 1;
  const int rg = tid / p.n_mels, col = tid - rg * p.n_mels;
  auto floored = [&](float x) -> float { return thr != thr ? thr : (x < thr ? thr : x); };      // keeps a NaN x
  if (p.mean_norm) {
    float s = 0.f;
    if (rg < RG)
      for (int t = rg; t < p.T; t += RG) s += floored(dst[(size_t)t * p.ld_out + col]);
    colsum[tid] = s;
    __syncthreads();
    if (tid < p.n_mels) {
      float sum = 0.f;
      for (int k = 0; k < RG; ++k) sum += colsum[k * p.n_mels + tid];
      mean_s[tid] = sum / (float)p.T;
    }
    __syncthreads();
  }
  for (int e = tid; e < total; e += FIN_THREADS) {
    const int t = e / p.n_mels, m = e - t * p.n_mels;
    float* q = dst + (size_t)t * p.ld_out + m;
    *q = floored(*q) - (p.mean_norm ? mean_s[m] : 0.f);
  }
}

int round_up(int v, int m) { return (v + m - 1) / m * m; }

size_t pad_len(const sd_fbank_plan* plan, int n) { return ((size_t)n + 2 * (size_t)(plan->n_fft / 2) + 3) & ~(size_t)3; }

size_t per_utterance_bytes(const sd_fbank_plan* plan, int n, int T) {
  return (pad_len(plan, n) + (size_t)T * ((size_t)plan->g_nfp + (size_t)plan->g_nmp)) * sizeof(float);
}

int chunk_utterances(const sd_fbank_plan* plan, int B, int n, int T) {
  const size_t per = per_utterance_bytes(plan, n, T);
  size_t bc = FBG_CHUNK_BYTES / (per ? per : 1);
  if (bc < 1) bc = 1;
  return (size_t)B < bc ? B : (int)bc;
}

}  // namespace

bool sd_fbank_generic_geometry_ok(int n_fft, int hop, int n_mels) {
  return n_fft >= 8 && n_fft <= FBG_MAX_NFFT && hop >= 1 && hop <= n_fft && n_mels >= 1 && n_mels <= 256;
}

// frames of torch.stft(center = True): 1 + (n + 2 (n_fft / 2) - n_fft) / hop  (= 1 + n / hop for an even n_fft)
int sd_fbank_generic_num_frames(const sd_fbank_plan* plan, int n) {
  const long long padded = (long long)n + 2 * (plan->n_fft / 2);
  return padded < plan->n_fft ? 0 : (int)(1 + (padded - plan->n_fft) / plan->hop);
}

int sd_fbank_generic_create_tables(sd_fbank_plan* plan, const float* window, const float* mel_fb) {
  const int n_fft = plan->n_fft, n_mels = plan->n_mels;
  const int n_freq = n_fft / 2 + 1;
  plan->g_nfreq = n_freq;
  plan->g_nfp = round_up(n_freq, 32);                          // 32 bins = one 64-column tile of the DFT kernel; the mel product's K step
  plan->g_nmp = round_up(n_mels, 4);
  const int nfp = plan->g_nfp, ncol = 2 * nfp;
  // twiddles in float64, [n_fft][2 nfp]: (re, im) columns of a bin side by side
  std::vector<double> tw((size_t)n_fft * ncol, 0.0);
  for (int pos = 0; pos < n_fft; ++pos)
    for (int k = 0; k < n_freq; ++k) {
      const long long ph = ((long long)k * pos) % n_fft;
      const double ang = 2.0 * M_PI * (double)ph / (double)n_fft;
      tw[(size_t)pos * ncol + 2 * k] = (double)window[pos] * std::cos(ang);
      tw[(size_t)pos * ncol + 2 * k + 1] = -(double)window[pos] * std::sin(ang);
    }
  // mel weights [cout = n_mels][1][cin_pad = nfp]
  std::vector<float> wm((size_t)n_mels * nfp, 0.f);
  for (int m = 0; m < n_mels; ++m)
    for (int f = 0; f < n_freq; ++f) wm[(size_t)m * nfp + f] = mel_fb[(size_t)f * n_mels + m];
  hipError_t e = hipMalloc(&plan->g_wdft_dev, tw.size() * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&plan->g_wmel_dev, wm.size() * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(plan->g_wdft_dev, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(plan->g_wmel_dev, wm.data(), wm.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    sd_fbank_generic_destroy_tables(plan);
    return sd_set_error(SD_ERR_HIP, "sd_fbank_plan_create: device table upload failed: %s", hipGetErrorString(e));
  }
  return SD_OK;
}

void sd_fbank_generic_destroy_tables(sd_fbank_plan* plan) {
  if (plan->g_wdft_dev) (void)hipFree(plan->g_wdft_dev);
  if (plan->g_wmel_dev) (void)hipFree(plan->g_wmel_dev);
  plan->g_wdft_dev = plan->g_wmel_dev = nullptr;
}

size_t sd_fbank_generic_workspace_bytes(const sd_fbank_plan* plan, int B, int n) {
  if (B <= 0 || n < 0) return 256;
  const int T = sd_fbank_generic_num_frames(plan, n);
  return ((size_t)chunk_utterances(plan, B, n, T) * per_utterance_bytes(plan, n, T) + 1023) & ~(size_t)255;
}

int sd_fbank_generic_launch(const sd_fbank_plan* plan, const float* wav_dev, long long n_total, const long long* starts_dev, int B, int n,
                            int mean_norm, float* out_dev, int ld_out, void* ws_dev, size_t ws_bytes, hipStream_t stream) {
  const int T = sd_fbank_generic_num_frames(plan, n);
  SD_CHECK_ARG(T >= 1, "sd_fbank_f32: %d samples give no frame at n_fft=%d", n, plan->n_fft);
  const int nfp = plan->g_nfp, nmp = plan->g_nmp;
  SD_CHECK_ARG((long long)T * B < (1LL << 31), "sd_fbank_f32: too many frames");
  SD_CHECK_ARG(sd_aligned16(ws_dev) && ws_bytes >= sd_fbank_generic_workspace_bytes(plan, B, n), "sd_fbank_f32: workspace too small or misaligned");
  const int Bc = chunk_utterances(plan, B, n, T);
  const int Lp = (int)pad_len(plan, n);
  float* xpad = static_cast<float*>(ws_dev);
  float* pw = xpad + (size_t)Bc * Lp;
  float* mel = pw + (size_t)Bc * T * nfp;
  const int use_floor = plan->log_mode == SD_LOG_DB_TOPDB && plan->top_db >= 0.f;
  for (int b0 = 0; b0 < B; b0 += Bc) {
    const int nb = B - b0 < Bc ? B - b0 : Bc;
    const long long M = (long long)nb * T;
    PadArgs pa{wav_dev, starts_dev, n_total, xpad, b0, nb, n, Lp, plan->n_fft / 2, plan->pad_mode};
    const long long tot = (long long)nb * Lp;
    const unsigned grid = (unsigned)((tot + 255) / 256 < 65536 ? (tot + 255) / 256 : 65536);
    hipLaunchKernelGGL(fbg_pad_kernel, dim3(grid), dim3(256), 0, stream, pa);
    SD_CHECK_LAUNCH("fbg_pad_kernel");
    DftArgs da{xpad, Lp, plan->hop, T, M, static_cast<const double*>(plan->g_wdft_dev), plan->n_fft, 2 * nfp, pw, nfp};
    hipLaunchKernelGGL(fbg_dft_f64_kernel, dim3((unsigned)((M + DT - 1) / DT), (unsigned)(2 * nfp / DT)), dim3(256), 0, stream, da);
    SD_CHECK_LAUNCH("fbg_dft_f64_kernel");
    sd_conv_args a;
    std::memset(&a, 0, sizeof(a));
    a.x = pw; a.lda = nfp; a.w = plan->g_wmel_dev; a.w_dtype = SD_DT_F32; a.y = mel; a.ldo = nmp;
    a.M = (int)M; a.T = 1; a.cin = nfp; a.cin_pad = nfp; a.cout = plan->n_mels; a.taps = 1; a.dil = 1;
    a.act = SD_ACT_NONE; a.act2 = SD_ACT_NONE;
    if (int e = sd_conv1d_cl_f32(&a, stream)) return e;
    FinArgs fa{mel, nmp, T, 0, out_dev, ld_out, b0, T, plan->n_mels, plan->log_mode, plan->log_eps, plan->top_db, use_floor, mean_norm};
    hipLaunchKernelGGL(fbg_finalize_kernel, dim3((unsigned)nb), dim3(FIN_THREADS), 0, stream, fa);
    SD_CHECK_LAUNCH("fbg_finalize_kernel");
  }
  return SD_OK;
}
