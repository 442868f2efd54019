// Micro-benchmark: does v_mfma_f32_32x32x2_f32 share execution resources with ordinary f32 VALU work on gfx950?
// 8 waves per workgroup (2 per SIMD), 1 workgroup per CU (LDS-limited like the correlation kernel).
//   mode 0: every wave runs a dependent MFMA chain
//   mode 1: waves 0..3 MFMA chain, waves 4..7 idle (exit)                  -> one MFMA wave per SIMD
//   mode 2: waves 0..3 MFMA chain, waves 4..7 run an f32 VALU fma loop     -> MFMA + VALU wave per SIMD
//   mode 3: waves 0..3 MFMA chain, waves 4..7 run an LDS read loop         -> MFMA + LDS wave per SIMD
//   mode 4: waves 4..7 only the VALU loop (for its standalone time)
// Reports ms and the MFMA TFLOP/s of the MFMA waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(768, 3) k(float* out, int iters, int valu_iters) {
  extern __shared__ float lds[];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  // modes 5/6/7: waves 0..7 run the MFMA chain (2 per SIMD = full matrix rate), waves 8..11 run VALU / LDS / nothing
  if (MODE <= 4 && w >= 8) return;
  const bool mf = (MODE == 0) || (MODE >= 5 ? w < 8 : (w < 4));   // waves map to SIMD w % 4
  if (mf && MODE != 4) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = 1.0f + l, b = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int t = 0; t < 32; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (acc[0] == 12345.f) out[threadIdx.x] = acc[3];
  } else if (MODE == 2 || MODE == 4 || MODE == 5) {
    float x0 = l, x1 = l + 1, x2 = l + 2, x3 = l + 3;
    for (int i = 0; i < valu_iters; ++i) {
#pragma unroll
      for (int t = 0; t < 32; ++t) {
        x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
      }
    }
    if (x0 + x1 + x2 + x3 == 12345.f) out[threadIdx.x] = x0;
  } else if (MODE == 3 || MODE == 6) {
    float s = 0.f;
    for (int i = 0; i < valu_iters; ++i) {
#pragma unroll
      for (int t = 0; t < 32; ++t) s += lds[(l + t * 64 + i) & 8191];
    }
    if (s == 12345.f) out[threadIdx.x] = s;
  }
}

template <int NCH>
__global__ void __launch_bounds__(256, 1) kchain(float* out, int iters) {   // 4 waves per workgroup = 1 per SIMD
  const int l = threadIdx.x & 63;
  f32x16 acc[NCH];
  for (int c = 0; c < NCH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = 1.0f + l, b = 0.5f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 32 / NCH; ++t)
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < NCH; ++c) s += acc[c][0];
  if (s == 12345.f) out[threadIdx.x] = s;
}

template <int NCH>
float run_chain(float* d, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kchain<NCH>, dim3(256), dim3(256), 0, 0, d, 16);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(kchain<NCH>, dim3(256), dim3(256), 0, 0, d, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}

template <int MODE>
float run(float* d, int iters, int vi) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(768), 160 * 1024, 0, d, 16, 16);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(768), 160 * 1024, 0, d, iters, vi);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  float* d; hipMalloc(&d, 1 << 20);
  const int iters = 20000;   // x32 MFMAs per MFMA wave
  const double fl_per_wave = (double)iters * 32 * 4096;
  float t0 = run<0>(d, iters, 0), t1 = run<1>(d, iters, 0);
  // VALU loop sized to last about as long as the MFMA chain: 32*4 fma per iter, 2 cycles each -> 256 cyc/iter vs 2048 cyc/iter MFMA
  float t4 = run<4>(d, iters, iters * 2);
  float t2 = run<2>(d, iters, iters * 2), t3 = run<3>(d, iters, iters * 2);
  {
    float c1 = run_chain<1>(d, iters), c2 = run_chain<2>(d, iters), c4 = run_chain<4>(d, iters);
    const double fl = 256.0 * 4 * iters * 32 * 4096;
    printf("{\"one_wave_per_simd_1chain_tflops\": %.1f, \"2chains_tflops\": %.1f, \"4chains_tflops\": %.1f}\n",
           fl / (c1 * 1e-3) / 1e12, fl / (c2 * 1e-3) / 1e12, fl / (c4 * 1e-3) / 1e12);
  }
  float t7 = run<7>(d, iters, 0);
  float t5a = run<5>(d, iters, iters * 2), t5b = run<5>(d, iters, iters * 4), t6a = run<6>(d, iters, iters / 2), t6b = run<6>(d, iters, iters);
  printf("{\"mode7_8mfma_4idle_ms\": %.3f, \"mode5_8mfma_4valu_quarter_ms\": %.3f, \"mode5_8mfma_4valu_half_ms\": %.3f, "
         "\"mode6_8mfma_4lds_a_ms\": %.3f, \"mode6_8mfma_4lds_b_ms\": %.3f}\n", t7, t5a, t5b, t6a, t6b);
  printf("{\"mode0_all_mfma_ms\": %.3f, \"mode0_tflops\": %.1f, \"mode1_one_mfma_wave_per_simd_ms\": %.3f, \"mode1_tflops\": %.1f, "
         "\"mode4_valu_only_ms\": %.3f, \"mode2_mfma_plus_valu_wave_ms\": %.3f, \"mode3_mfma_plus_lds_wave_ms\": %.3f}\n",
         t0, 256 * 8 * fl_per_wave / (t0 * 1e-3) / 1e12, t1, 256 * 4 * fl_per_wave / (t1 * 1e-3) / 1e12, t4, t2, t3);
  return 0;
}
