// Micro-benchmark: are matrix time and memory time ADDITIVE on gfx950 because of issue interference -- or because the chip is
// power-limited?  (the convolution family takes 1.00 x (matrix floor + HBM floor).)  mfma_vmem_share.hip shows
// that with constant MFMA operands a wave's MFMAs and its own / its neighbour's vector-memory instructions overlap almost
// perfectly (max, not sum).  Here the same streams with operands whose bits CHANGE from one MFMA to the next (four random
// register sets in rotation -- what real activations do to the multiplier array) against constant operands:
//   MFMA stream alone, memory stream alone (full-line streaming loads / stores of 1 KiB per wave instruction), both together;
//   wall time from events.  (s_memtime of one wave was tried as a clock read-out: its tick rate is not the shader clock here.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// what: bit 0 MFMAs, bit 1 loads, bit 2 stores;  rnd: random operand sets;  nmem: memory instructions per wave and 32 MFMAs
__global__ void __launch_bounds__(512, 2) k(int what, int rnd, int iters, int nmem, unsigned* __restrict__ buf, unsigned bytes_per_wave,
                                            float* __restrict__ out, unsigned long long* __restrict__ cycles) {
  extern __shared__ float lds[];
  const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long t0 = __builtin_readcyclecounter();
  char* base = reinterpret_cast<char*>(buf) + ((size_t)blockIdx.x * 8 + w) * bytes_per_wave;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes_per_wave, 0x00020000);
  unsigned off = (unsigned)l * 16u;
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  f16x8 a[4], b[4];
  for (int s = 0; s < 4; ++s)
    for (int e = 0; e < 8; ++e) {
      const unsigned ha = hash32((unsigned)(threadIdx.x * 64 + s * 8 + e) * 2654435761u + 17u), hb = hash32(ha + 0x9e3779b9u);
      // |values| < 2 with random mantissa bits; the constant variant uses one set for every MFMA
      a[s][e] = rnd ? (_Float16)((float)(int)(ha & 0xffff) * (1.0f / 32768.0f) - 1.0f) : (_Float16)(0.001f * (l + e));
      b[s][e] = rnd ? (_Float16)((float)(int)(hb & 0xffff) * (1.0f / 32768.0f) - 1.0f) : (_Float16)(0.5f + 0.01f * e);
    }
  u32x4 data = {(unsigned)l, hash32(l), 2u, 3u}, sink = {0u, 0u, 0u, 0u};
  const unsigned wrap = bytes_per_wave - 1024u;
  for (int i = 0; i < iters; ++i) {
    if (what & 1) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t & 3], b[(t + 1) & 3], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(t + 2) & 3], b[(t + 3) & 3], acc1, 0, 0, 0);
      }
    }
    if (what & 6) {
      for (int m = 0; m < nmem; ++m) {
        if (what & 2) sink ^= __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(data, rs, off, 0, 0);
        off += 1024u;
        if (off >= wrap) off = (unsigned)l * 16u;
      }
    }
  }
  if (acc0[0] + acc1[1] == 12345.f || sink[0] == 0x12345u) out[threadIdx.x] = acc0[3] + (float)sink[1];
  if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = __builtin_readcyclecounter() - t0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  const unsigned per_wave = 4u << 20;   // 4 MiB per wave, 8 GiB in all: streamed once per ~4 000 instructions, never cached
  unsigned* buf; float* out; unsigned long long* cyc;
  hipMalloc(&buf, (size_t)per_wave * 8 * 256); hipMalloc(&out, 1 << 16); hipMalloc(&cyc, 64);
  hipMemset(buf, 0, (size_t)per_wave * 8 * 256);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  struct R { float ms; double ghz; };
  auto run = [&](int what, int rnd, int nmem) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 160 * 1024, 0, what, rnd, 64, nmem, buf, per_wave, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 160 * 1024, 0, what, rnd, iters, nmem, buf, per_wave, out, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    return R{ms, (double)c / (ms * 1e6)};
  };
  const double flops = 256.0 * 8 * iters * 32 * 32768.0;
  for (int rnd : {0, 1})
    for (int nmem : {1, 2, 4}) {
      const R m = run(1, rnd, nmem), ld = run(2, rnd, nmem), st = run(4, rnd, nmem), ml = run(3, rnd, nmem), ms_ = run(5, rnd, nmem);
      const double gb = 256.0 * 8 * iters * nmem * 1024 / 1e9;
      printf("{\"random_operands\": %d, \"vmem_per_wave_per_32_mfma\": %d, \"GB\": %.1f, "
             "\"mfma_ms\": %.3f, \"mfma_tflops\": %.0f, "
             "\"loads_ms\": %.3f, \"loads_TBs\": %.2f, \"stores_ms\": %.3f, \"stores_TBs\": %.2f, "
             "\"mfma_and_loads_ms\": %.3f, \"vs_max\": %.2f, \"vs_sum\": %.2f, "
             "\"mfma_and_stores_ms\": %.3f, \"st_vs_max\": %.2f, \"st_vs_sum\": %.2f, \"store_stream_adds_fraction_of_its_own_time\": %.2f}\n",
             rnd, nmem, gb, m.ms, flops / (m.ms * 1e-3) / 1e12, ld.ms, gb / ld.ms, st.ms, gb / st.ms, ml.ms,
             ml.ms / (m.ms > ld.ms ? m.ms : ld.ms), ml.ms / (m.ms + ld.ms), ms_.ms, ms_.ms / (m.ms > st.ms ? m.ms : st.ms),
             ms_.ms / (m.ms + st.ms), (ms_.ms - (m.ms > st.ms ? m.ms : st.ms)) / (m.ms > st.ms ? st.ms : m.ms));
    }
  return 0;
}
