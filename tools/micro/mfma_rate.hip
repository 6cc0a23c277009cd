// Microbenchmark: what does the matrix pipe deliver with NO memory operation at all?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/micro/mfma_rate.hip && /tmp/mfma_rate
// Each wave runs `iters` rounds of independent accumulator chains (4 of 32x32x16, 8 of 16x16x32) on register operands;
// 256-thread workgroups (one wave per SIMD), 1 / 2 / 4 of them resident per CU (limited by a dynamic LDS allocation),
// 8 rounds of workgroups per CU; operands zero or random.  Reports cycles per MFMA seen by a wave (s_memtime), the shader
// clock (s_memtime / s_memrealtime at 100 MHz) and chip TFLOP/s from HIP events.  The 2.5 PFLOP/s dense f16 figure
// is 1024 flop per clock and SIMD at 2.4 GHz: one 32x32x16 MFMA per 32 cycles, one 16x16x32 per 16; the 157.3 TFLOP/s f32
// figure is 64 flop per clock and SIMD: one 32x32x2 per 64 cycles, one 16x16x4 per 32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>   // 0: v_mfma_f32_32x32x16_f16 (4 chains), 1: v_mfma_f32_16x16x32_f16 (8 chains), 2: 32x32x16 with 8 chains and 4 + 2 operand fragments (the 4 x 2 tile pattern of a conv wave)
__global__ __launch_bounds__(256) void spin(const h8* __restrict__ ops, int iters, float* sink, unsigned long long* stamps) {
  const int tid = threadIdx.x;
  const h8 a = ops[tid & 63], b = ops[64 + (tid & 63)];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float out = 0.f;
  if (SHAPE == 0) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out += acc[i][0] + acc[i][15];
  } else if (SHAPE == 3 || SHAPE == 4) {   // exact f32: v_mfma_f32_32x32x2_f32 (4 chains) / v_mfma_f32_16x16x4_f32 (8 chains)
    const float fa = (float)a[0], fb = (float)b[1];
    if (SHAPE == 3) {
      f32x16 acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) out += acc[i][0] + acc[i][15];
    } else {
      f32x4 acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) out += acc[i][0] + acc[i][3];
    }
  } else if (SHAPE == 2) {
    f32x16 acc[4][2];
    h8 fa[4], fb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = ops[(tid + i) & 63];
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = ops[64 + ((tid + j) & 63)];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) out += acc[i][j][0] + acc[i][j][15];
  } else {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // inline asm: with the builtin, hipcc (ROCm 7.2) rotates the eight accumulators through ~50 v_accvgpr moves per
    // round of 8 MFMAs, and round 2's "45 cycles per MFMA" was that copy traffic, not the instruction
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) out += acc[i][0] + acc[i][3];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((tid & 63) == 0) {
    stamps[(blockIdx.x * 4 + (tid >> 6)) * 2] = t1 - t0;
    stamps[(blockIdx.x * 4 + (tid >> 6)) * 2 + 1] = r1 - r0;
  }
  if (out == 123.456f) sink[0] = out;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int max_blocks = cus * 4 * 8;
  h8* ops; float* sink; unsigned long long* stamps;
  CHECK(hipMalloc(&ops, 128 * sizeof(h8)));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipMalloc(&stamps, (size_t)max_blocks * 4 * 2 * sizeof(unsigned long long)));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  std::vector<_Float16> host(128 * 8);
  for (int random = 0; random < 2; ++random) {
    for (auto& v : host) v = (_Float16)(random ? (float)(rand() % 2001 - 1000) / 1000.f : 0.f);
    CHECK(hipMemcpy(ops, host.data(), host.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    for (int shape = 0; shape < 5; ++shape) {
      for (int wps : {1, 2, 4}) {
        if (shape == 2 && wps == 4) continue;              // 140 registers: two waves per SIMD
        const int chains = shape == 0 || shape == 3 ? 4 : 8;
        const int iters = shape == 1 ? 40000 : shape == 0 ? 20000 : shape == 2 ? 10000 : shape == 3 ? 10000 : 10000;
        const size_t lds = (size_t)(160 / wps) * 1024;           // wps workgroups fit a CU
        const int blocks = cus * wps * 8;
        const double flop_per_mfma = shape == 3 ? 32.0 * 32 * 2 * 2 : shape == 4 ? 16.0 * 16 * 4 * 2 : shape != 1 ? 32.0 * 32 * 16 * 2 : 16.0 * 16 * 32 * 2;
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) {       // the second launch is the measurement
          CHECK(hipEventRecord(e0, 0));
          if (shape == 0) hipLaunchKernelGGL(spin<0>, dim3(blocks), dim3(256), lds, 0, ops, iters, sink, stamps);
          else if (shape == 1) hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(256), lds, 0, ops, iters, sink, stamps);
          else if (shape == 2) hipLaunchKernelGGL(spin<2>, dim3(blocks), dim3(256), lds, 0, ops, iters, sink, stamps);
          else if (shape == 3) hipLaunchKernelGGL(spin<3>, dim3(blocks), dim3(256), lds, 0, ops, iters, sink, stamps);
          else hipLaunchKernelGGL(spin<4>, dim3(blocks), dim3(256), lds, 0, ops, iters, sink, stamps);
          CHECK(hipEventRecord(e1, 0));
          CHECK(hipEventSynchronize(e1));
          CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        std::vector<unsigned long long> st((size_t)blocks * 4 * 2);
        CHECK(hipMemcpy(st.data(), stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double cyc = 0, real = 0;
        for (size_t i = 0; i < (size_t)blocks * 4; ++i) { cyc += (double)st[2 * i]; real += (double)st[2 * i + 1]; }
        const double n = (double)blocks * 4;
        const double per_mfma_wave = cyc / n / ((double)iters * chains);
        const double ghz = cyc / real * 0.1;
        const double tf = flop_per_mfma * iters * chains * n / (ms * 1e-3) / 1e12;
        printf("%s operands, %s, %d wave(s) per SIMD: %.1f cycles per MFMA seen by a wave = %.1f per SIMD, shader clock %.2f GHz, %.0f TFLOP/s\n",
               random ? "random" : "zero", shape == 0 ? "32x32x16 f16" : shape == 1 ? "16x16x32 f16" : shape == 2 ? "32x32x16 f16 (4 x 2 tiles)" : shape == 3 ? "32x32x2 f32" : "16x16x4 f32", wps, per_mfma_wave, per_mfma_wave / wps, ghz, tf);
      }
    }
  }
  return 0;
}
