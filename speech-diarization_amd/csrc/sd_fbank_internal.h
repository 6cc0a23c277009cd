// Shared between sd_fbank.hip (plan, tables of the folded-DFT kernel, dispatch) and sd_fbank_utt16.hip (the one-launch,
// one-workgroup-per-utterance kernel with the factored DFT).
#pragma once
#include "sd_common.h"

struct sd_fbank_plan {
  int n_fft, hop, n_mels, pad_mode, log_mode;
  float log_eps, top_db;
  void* basis16_dev;       // folded-DFT kernel: f16 [pass][k step][tile][Chi | Clo | Shi | Slo][64][8]
  void* melw16_dev;        // folded-DFT kernel: bf16 [bin tile][k half][mel tile][W1 | W2][64][8]
  void* utt16_tables_dev;  // factored one-launch kernel (sd_fbank_utt16.hip): all of its fragment tables, 1 KB each
  // any other framing (sd_fbank_generic.hip): a float64 DFT kernel + the mel product on the exact-f32 conv operator
  bool generic;
  int g_nfreq, g_nfp, g_nmp;
  void* g_wdft_dev;        // f64 [n_fft][2 nfp]: window[p] cos / -sin(2 pi k p / n_fft), (re, im) of a bin side by side
  void* g_wmel_dev;        // f32 [n_mels][1][nfp]
};

bool sd_fbank_generic_geometry_ok(int n_fft, int hop, int n_mels);
int sd_fbank_generic_create_tables(sd_fbank_plan* plan, const float* window, const float* mel_fb);
void sd_fbank_generic_destroy_tables(sd_fbank_plan* plan);
int sd_fbank_generic_num_frames(const sd_fbank_plan* plan, int n);
size_t sd_fbank_generic_workspace_bytes(const sd_fbank_plan* plan, int B, int n);
int sd_fbank_generic_launch(const sd_fbank_plan* plan, const float* wav_dev, long long n_total, const long long* starts_dev, int B, int n,
                            int mean_norm, float* out_dev, int ld_out, void* ws_dev, size_t ws_bytes, hipStream_t stream);

// device tables of the factored kernel from the window (n_fft values) and the mel filterbank [n_fft / 2 + 1][n_mels]; utterances of up to
// 32 100 samples = 201 frames (the padded signal must fit the CU's LDS beside the table ring)
int sd_fbank_utt16_create_tables(sd_fbank_plan* plan, const float* window, const float* mel_fb);
void sd_fbank_utt16_destroy_tables(sd_fbank_plan* plan);
bool sd_fbank_utt16_supported(const sd_fbank_plan* plan, int n);
int sd_fbank_utt16_launch(const sd_fbank_plan* plan, const float* wav_dev, long long n_total, const long long* starts_dev, int B, int n,
                          int mean_norm, float* out_dev, int ld_out, hipStream_t stream);
