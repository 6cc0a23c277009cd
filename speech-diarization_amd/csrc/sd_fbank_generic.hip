// Log-mel filterbank for ANY framing (n_fft, hop): `fbank_batch(wavs, sr)` with sr != 16000 [REF speech_encode.py:14-24]
// (win_length = n_fft = int(sr * 0.025), hop = int(sr * 0.010): 200 / 80 at 8 kHz, 551 / 220 at 22.05 kHz, 1200 / 480 at 48 kHz).
//
// The 16 kHz framing (400 / 160) has kernels of its own (sd_fbank_utt16.hip, sd_fbank.hip: factored / folded DFT on split f16).  Every
// other framing takes this path, which is built from the library's exact-f32 operators instead of a kernel per (n_fft, hop):
//
//   1. fbg_rows_kernel      the padded signal (center = True: n_fft / 2 samples of reflect or zero padding at both ends) laid out as rows of
//                           `hop` samples, channel-last [B * R][hop_pad]: frame t is the n_fft samples that start at row t, i.e. rows
//                           t .. t + taps - 1 with taps = ceil(n_fft / hop) (made odd).
//   2. sd_conv1d_cl_f32     the windowed real DFT of every frame as ONE implicit GEMM over those rows on the f32 matrix cores: a `taps`-tap
//                           convolution whose weights are window[p] * cos / sin(2 pi k p / n_fft) at p = tap * hop + column (zero past
//                           n_fft), cout = [re bins | im bins].  Output row t + taps / 2 is frame t; the reflect rows of the operator's
//                           "same" padding only reach rows outside [taps / 2, taps / 2 + T) and are never read.
//   3. fbg_power_kernel     re^2 + im^2.
//   4. sd_conv1d_cl_f32     the mel product [n_freq] -> [n_mels] as a pointwise conv (dense: any filter shapes).
//   5. fbg_finalize_kernel  log law, utterance maximum / top_db floor, mean over T, the write of out[b][t][m]; one workgroup per utterance.
//
// f32 products with f32 accumulation throughout (v_mfma_f32_32x32x2_f32 / 16x16x4): the arithmetic of torch's own f32 STFT.  Speed is
// secondary here (the reference's pipeline resamples to 16 kHz before it embeds, [REF anti_stick_diarize.py:29-41]): 1.2 x the DFT's
// flops for the zero-padded taps, four passes over a [frames][n_fft]-sized intermediate.  Utterances are processed in chunks so that the
// workspace stays below ~256 MB whatever B is.  A NaN sample propagates the way the reference's arithmetic does: its frames' bins are
// NaN, so is the utterance maximum (all features NaN under the top_db floor) and the mean over T (all NaN with mean removal).
#include <cmath>
#include <cstring>
#include <vector>

#include "sd_fbank_internal.h"

namespace {

constexpr int FBG_MAX_NFFT = 8192;
constexpr size_t FBG_CHUNK_BYTES = (size_t)256 << 20;

struct RowsArgs {
  const float* wav; const long long* starts; long long n_total;
  float* xp; int B0, Bc, n, R, hop, hop_pad, pad, pad_mode;
};

// xp[(b, r)][c] = padded signal at s = r * hop + c - pad (c < hop), 0 for c >= hop and wherever s lies outside the padded extent
__global__ void fbg_rows_kernel(RowsArgs p) {
  const long long total = (long long)p.Bc * p.R * p.hop_pad;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % p.hop_pad);
    const long long row = e / p.hop_pad;
    const int r = (int)(row % p.R), b = (int)(row / p.R);
    float v = 0.f;
    if (c < p.hop) {
      long long s = (long long)r * p.hop + c - p.pad;
      bool ok = s >= -(long long)p.pad && s < (long long)p.n + p.pad;
      if (p.pad_mode == SD_PAD_REFLECT) {
        s = s < 0 ? -s : s;
        s = s >= p.n ? 2LL * (p.n - 1) - s : s;
        ok = ok && s >= 0 && s < p.n;
      } else {
        ok = ok && s >= 0 && s < p.n;
      }
      if (ok) {
        const long long start = p.starts ? p.starts[p.B0 + b] : (long long)(p.B0 + b) * p.n;
        const long long gi = start + s;
        if (gi >= 0 && gi < p.n_total) v = p.wav[gi];            // a window may hang over either end of the signal: zeros there
      }
    }
    p.xp[e] = v;
  }
}

// pw[m][f] = y[m][f]^2 + y[m][nfp + f]^2, f < nfp (columns past n_freq are zero in y, hence in pw)
__global__ void fbg_power_kernel(const float* y, float* pw, long long rows, int nfp) {
  const int q = nfp >> 2;
  const long long total = rows * q;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long m = e / q;
    const int f = (int)(e - m * q) * 4;
    const f32x4 re = *reinterpret_cast<const f32x4*>(y + m * 2 * nfp + f);
    const f32x4 im = *reinterpret_cast<const f32x4*>(y + m * 2 * nfp + nfp + f);
    *reinterpret_cast<f32x4*>(pw + m * nfp + f) = re * re + im * im;
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

size_t per_utterance_bytes(const sd_fbank_plan* plan, int R) {
  return (size_t)R * ((size_t)plan->g_hop_pad + 3 * (size_t)plan->g_nfp + (size_t)plan->g_nmp) * sizeof(float);
}

int chunk_utterances(const sd_fbank_plan* plan, int B, int R) {
  const size_t per = per_utterance_bytes(plan, R);
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
  const int n_fft = plan->n_fft, hop = plan->hop, n_mels = plan->n_mels;
  int taps = (n_fft + hop - 1) / hop;
  if (!(taps & 1)) ++taps;                                     // the operator centres an odd number of taps
  const int n_freq = n_fft / 2 + 1;
  plan->g_taps = taps;
  plan->g_hop_pad = round_up(hop, 32);
  plan->g_nfreq = n_freq;
  plan->g_nfp = round_up(n_freq, 64);
  plan->g_nmp = round_up(n_mels, 4);
  const int hp = plan->g_hop_pad, nfp = plan->g_nfp;
  // DFT weights, packed as the conv operator wants them: [cout = 2 nfp][taps][cin_pad = hop_pad]
  std::vector<float> wd((size_t)2 * nfp * taps * hp, 0.f);
  for (int k = 0; k < n_freq; ++k)
    for (int j = 0; j < taps; ++j)
      for (int c = 0; c < hop; ++c) {
        const int pos = j * hop + c;
        if (pos >= n_fft) continue;
        const long long ph = ((long long)k * pos) % n_fft;
        const double ang = 2.0 * M_PI * (double)ph / (double)n_fft;
        const double w = (double)window[pos];
        wd[((size_t)k * taps + j) * hp + c] = (float)(w * std::cos(ang));
        wd[((size_t)(nfp + k) * taps + j) * hp + c] = (float)(-w * std::sin(ang));
      }
  // mel weights [cout = n_mels][1][cin_pad = nfp]
  std::vector<float> wm((size_t)n_mels * nfp, 0.f);
  for (int m = 0; m < n_mels; ++m)
    for (int f = 0; f < n_freq; ++f) wm[(size_t)m * nfp + f] = mel_fb[(size_t)f * n_mels + m];
  hipError_t e = hipMalloc(&plan->g_wdft_dev, wd.size() * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&plan->g_wmel_dev, wm.size() * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(plan->g_wdft_dev, wd.data(), wd.size() * sizeof(float), hipMemcpyHostToDevice);
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
  const int R = T + plan->g_taps - 1;
  return ((size_t)chunk_utterances(plan, B, R) * per_utterance_bytes(plan, R) + 1023) & ~(size_t)255;
}

int sd_fbank_generic_launch(const sd_fbank_plan* plan, const float* wav_dev, long long n_total, const long long* starts_dev, int B, int n,
                            int mean_norm, float* out_dev, int ld_out, void* ws_dev, size_t ws_bytes, hipStream_t stream) {
  const int T = sd_fbank_generic_num_frames(plan, n);
  SD_CHECK_ARG(T >= 1, "sd_fbank_f32: %d samples give no frame at n_fft=%d", n, plan->n_fft);
  const int taps = plan->g_taps, hp = plan->g_hop_pad, nfp = plan->g_nfp, nmp = plan->g_nmp;
  const int R = T + taps - 1;
  SD_CHECK_ARG((long long)R * B < (1LL << 31), "sd_fbank_f32: too many frame rows");
  SD_CHECK_ARG(sd_aligned16(ws_dev) && ws_bytes >= sd_fbank_generic_workspace_bytes(plan, B, n), "sd_fbank_f32: workspace too small or misaligned");
  const int Bc = chunk_utterances(plan, B, R);
  float* xp = static_cast<float*>(ws_dev);
  float* y = xp + (size_t)Bc * R * hp;
  float* pw = y + (size_t)Bc * R * 2 * nfp;
  float* mel = pw + (size_t)Bc * R * nfp;
  const int use_floor = plan->log_mode == SD_LOG_DB_TOPDB && plan->top_db >= 0.f;
  for (int b0 = 0; b0 < B; b0 += Bc) {
    const int nb = B - b0 < Bc ? B - b0 : Bc;
    const long long rows = (long long)nb * R;
    RowsArgs ra{wav_dev, starts_dev, n_total, xp, b0, nb, n, R, plan->hop, hp, plan->n_fft / 2, plan->pad_mode};
    const long long tot = rows * hp;
    unsigned grid = (unsigned)((tot + 255) / 256 < 65536 ? (tot + 255) / 256 : 65536);
    hipLaunchKernelGGL(fbg_rows_kernel, dim3(grid), dim3(256), 0, stream, ra);
    SD_CHECK_LAUNCH("fbg_rows_kernel");
    sd_conv_args a;
    std::memset(&a, 0, sizeof(a));
    a.x = xp; a.lda = hp; a.w = plan->g_wdft_dev; a.w_dtype = SD_DT_F32; a.y = y; a.ldo = 2 * nfp;
    a.M = (int)rows; a.T = R; a.cin = hp; a.cin_pad = hp; a.cout = 2 * nfp; a.taps = taps; a.dil = 1;
    a.act = SD_ACT_NONE; a.act2 = SD_ACT_NONE;
    if (int e = sd_conv1d_cl_f32(&a, stream)) return e;
    const long long pt = rows * (nfp >> 2);
    grid = (unsigned)((pt + 255) / 256 < 65536 ? (pt + 255) / 256 : 65536);
    hipLaunchKernelGGL(fbg_power_kernel, dim3(grid), dim3(256), 0, stream, y, pw, rows, nfp);
    SD_CHECK_LAUNCH("fbg_power_kernel");
    std::memset(&a, 0, sizeof(a));
    a.x = pw; a.lda = nfp; a.w = plan->g_wmel_dev; a.w_dtype = SD_DT_F32; a.y = mel; a.ldo = nmp;
    a.M = (int)rows; a.T = 1; a.cin = nfp; a.cin_pad = nfp; a.cout = plan->n_mels; a.taps = 1; a.dil = 1;
    a.act = SD_ACT_NONE; a.act2 = SD_ACT_NONE;
    if (int e = sd_conv1d_cl_f32(&a, stream)) return e;
    FinArgs fa{mel, nmp, R, taps / 2, out_dev, ld_out, b0, T, plan->n_mels, plan->log_mode, plan->log_eps, plan->top_db, use_floor, mean_norm};
    hipLaunchKernelGGL(fbg_finalize_kernel, dim3((unsigned)nb), dim3(FIN_THREADS), 0, stream, fa);
    SD_CHECK_LAUNCH("fbg_finalize_kernel");
  }
  return SD_OK;
}
