// Microbenchmark: how many bytes per clock per CU can waves pull through the vector-memory path?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_rate tools/micro/dma_rate.hip && /tmp/dma_rate
// Variants: LDS-DMA (global_load_lds_dwordx4) vs register loads (global_load_dwordx4); source resident in L2
// (each workgroup re-reads its own 64 KB window) vs streaming from HBM.  8 waves per CU, as the 256x256 conv kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool DMA>
__global__ __launch_bounds__(512) void pull(const char* __restrict__ src, size_t window, size_t stride_per_block, int iters,
                                             float* sink, unsigned long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, wid = tid >> 6;
  const char* base = src + (size_t)blockIdx.x * stride_per_block;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  size_t off = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {                       // 4 pieces of 1 KB per wave per iteration = 32 KB per workgroup
      const char* g = base + off + (size_t)(j * 8 + wid) * 1024 + (tid & 63) * 16;
      if (DMA) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(lds + ((it & 3) * 32 + j * 8 + wid) * 1024), 16, 0, 0);
      } else {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g);
        acc += v;
      }
    }
    off += 32768;
    if (off + 32768 > window) off = 0;
    if (DMA && (it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0] + lds[tid];
}

int main() {
  int cus = 256;
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const size_t total = (size_t)4 << 30;
  char* src; float* sink; unsigned long long* cyc;
  CHECK(hipMalloc(&src, total)); CHECK(hipMemset(src, 1, total));
  CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&cyc, cus * sizeof(unsigned long long)));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pull<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  const int iters = 4096;                                 // 128 MB per workgroup
  struct Case { const char* name; bool dma; size_t window, stride; };
  const Case cases[] = {
      {"LDS-DMA, 64 KB window per workgroup (L2 hits)", true, 65536, 65536},
      {"LDS-DMA, one 2 MB window shared by all workgroups (L2 hits)", true, (size_t)2 << 20, 0},
      {"LDS-DMA, streaming 16 MB per workgroup (HBM)", true, (size_t)16 << 20, (size_t)16 << 20},
      {"register loads, 64 KB window per workgroup (L2 hits)", false, 65536, 65536},
      {"register loads, streaming 16 MB per workgroup (HBM)", false, (size_t)16 << 20, (size_t)16 << 20},
  };
  for (const Case& c : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      CHECK(hipEventRecord(e0));
      if (c.dma) hipLaunchKernelGGL(pull<true>, dim3(cus), dim3(512), 131072, 0, src, c.window, c.stride, iters, sink, cyc);
      else hipLaunchKernelGGL(pull<false>, dim3(cus), dim3(512), 0, 0, src, c.window, c.stride, iters, sink, cyc);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> h(cus);
      CHECK(hipMemcpy(h.data(), cyc, cus * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      double avg = 0; for (auto v : h) avg += (double)v; avg /= cus;
      const double bytes = (double)iters * 32768.0;
      if (rep == 1) printf("%-62s %6.1f B/clk/CU  (%.2f TB/s chip-wide, %.2f ms)\n", c.name, bytes / avg, bytes * cus / (ms * 1e-3) / 1e12, ms);
    }
  }
  return 0;
}
