// Sustained rate of v_mfma_f32_32x32x16_bf16 streams on gfx950: one wave per SIMD on every CU, NACC independent
// accumulators round-robin, operands all-zero or random bit patterns.  Prints ns and shader cycles (s_memtime) per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256, 1) k(const unsigned* in, float* out, unsigned long long* cyc, int iters) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    unsigned w[4];
    for (int q = 0; q < 4; ++q) w[q] = in[(threadIdx.x * 8 + i * 4 + q) & 4095];
    a[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(w));
    for (int q = 0; q < 4; ++q) w[q] = in[(threadIdx.x * 8 + i * 4 + q + 2048) & 4095];
    b[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(w));
  }
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 48 / NACC; ++u)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(n + u) & 3], b[(n * 3 + u) & 3], acc[n], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(const char* name, const unsigned* din, float* dout, unsigned long long* dcyc, int grid) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, din, dout, dcyc, 200);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, din, dout, dcyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> c(grid);
  hipMemcpy(c.data(), dcyc, grid * 8, hipMemcpyDeviceToHost);
  double mc = 0; for (auto v : c) mc += (double)v; mc /= grid;
  const double nm = (double)iters * 48;
  printf("{\"case\": \"%s\", \"nacc\": %d, \"ms\": %.3f, \"ns_per_mfma\": %.2f, \"memtime_ticks_per_mfma\": %.2f, \"tflops\": %.0f}\n", name, NACC, ms,
         ms * 1e6 / nm, mc / nm, grid * 4.0 * nm * 32768.0 / (ms * 1e-3) / 1e12);
}
int main() {
  std::vector<unsigned> h(4096);
  unsigned* din; float* dout; unsigned long long* dcyc;
  hipMalloc(&din, 4096 * 4); hipMalloc(&dout, 1024 * 256 * 4); hipMalloc(&dcyc, 1024 * 8);
  for (int pass = 0; pass < 3; ++pass) {
    const char* nm = pass == 0 ? "zeros" : pass == 1 ? "random_bits" : "normal_bf16_values";
    for (int i = 0; i < 4096; ++i) {
      if (pass == 0) h[i] = 0;
      else if (pass == 1) h[i] = (unsigned)rand() * 2654435761u;
      else { unsigned short lo = 0x3f00 + (rand() & 0xff) | ((rand() & 1) << 15), hi = 0x3f00 + (rand() & 0xff) | ((rand() & 1) << 15); h[i] = lo | ((unsigned)hi << 16); }
    }
    hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<4>(nm, din, dout, dcyc, 256);
    run<8>(nm, din, dout, dcyc, 256);
    run<16>(nm, din, dout, dcyc, 256);
    run<4>(nm, din, dout, dcyc, 64);     // a quarter of the chip: power headroom
  }
  return 0;
}
