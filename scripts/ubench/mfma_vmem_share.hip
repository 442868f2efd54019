// Micro-benchmark: what does a vector-memory instruction cost the MATRIX work of its SIMD on gfx950 -- and is the cost paid
// on the issuing wave's SIMD only?  (DESIGN.md 6.2 found the split kernel's VMEM and MFMA times ADD; section 11.1 proposes
// loader waves.  This decides where loader waves would have to live.)
// One workgroup of 8 waves per CU (160 KiB of LDS requested), roles by the SIMD each wave actually runs on (HW_ID.SIMD_ID):
//   mode 0  every wave runs MFMAs (two independent v_mfma_f32_32x32x16_f16 chains)              -> 2 MFMA waves per SIMD
//   mode 1  as 0, and every wave also issues one 16-byte-per-lane buffer STORE per `period` MFMAs  (what the conv kernel does)
//   mode 2  as 1 with LOADS (buffer_load_dwordx4 into VGPRs, consumed at the end)
//   mode 3  the first wave of every SIMD runs MFMAs, the second one only the stores of mode 1 (same total number)
//   mode 4  as 3 with loads
//   mode 5  SIMDs 0..2: both waves MFMAs; SIMD 3: both waves issue ALL the stores                -> memory work on its own SIMD
//   mode 6  as 5 with loads
//   mode 7  SIMDs 0..2: both waves MFMAs; SIMD 3 idle                                            -> reference for 5 / 6
//   mode 8  the first wave of every SIMD runs MFMAs, the second idles                           -> reference for 3 / 4
//   mode 9 / 10  no MFMAs at all: the second waves issue the stores / loads of mode 3 / 4         -> the memory stream alone
// `nmem` = VMEM instructions per wave and 32 MFMAs (the conv kernel: 19 per 108, i.e. ~6 per 32); two footprints per workgroup:
// 64 KiB (stays in L2) and 32 MiB (HBM) -- issue cost against bandwidth.
// Reported: kernel ms, MFMA instructions per SIMD-cycle equivalent (TFLOP/s of the MFMA waves), stores or loads issued.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512, 2) k(int mode, int iters, int nmem, unsigned* __restrict__ buf, unsigned buf_bytes_per_wg,
                                            float* __restrict__ out, int* __restrict__ simd_seen) {
  extern __shared__ float lds[];
  const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
  const int simd = (hwid >> 4) & 3;
  if (l == 0) atomicAdd(&simd_seen[simd * 8 + w], 1);
  // second wave of its SIMD?  (waves w and w + 4 share a SIMD when the dispatcher places them round-robin; checked on the host)
  const bool second = w >= 4;
  bool do_mfma, do_mem;
  int mem_per_period = nmem;
  switch (mode) {
    case 0: do_mfma = true; do_mem = false; break;
    case 1: case 2: do_mfma = true; do_mem = true; break;
    case 3: case 4: do_mfma = !second; do_mem = second; mem_per_period = 2 * nmem; break;     // the second wave issues both waves' share
    case 5: case 6: do_mfma = simd != 3; do_mem = simd == 3; mem_per_period = 4 * nmem; break;   // SIMD 3 issues all four SIMDs' share
    case 7: do_mfma = simd != 3; do_mem = false; break;
    case 9: case 10: do_mfma = false; do_mem = second; mem_per_period = 2 * nmem; break;
    default: do_mfma = !second; do_mem = false; break;
  }
  const bool loads = mode == 2 || mode == 4 || mode == 6 || mode == 10;
  char* base = reinterpret_cast<char*>(buf) + (size_t)blockIdx.x * buf_bytes_per_wg;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)buf_bytes_per_wg, 0x00020000);
  // the conv kernel's pattern: 16 bytes per lane at a 256-byte pixel pitch (32-byte runs of two lanes)
  unsigned off = (unsigned)((l & 31) * 256 + (l >> 5) * 16 + w * 8192);
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (l + e)); b[e] = (_Float16)(0.5f + 0.01f * e); }
  u32x4 data = {(unsigned)l, 1u, 2u, 3u}, sink = {0u, 0u, 0u, 0u};
  const unsigned step = buf_bytes_per_wg > (1u << 20) ? 65536u : 0u, wrap = buf_bytes_per_wg - 65536u;
  for (int i = 0; i < iters; ++i) {
    if (do_mfma) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
      }
    }
    if (do_mem) {
      for (int m = 0; m < mem_per_period; ++m) {
        if (loads) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
          sink ^= v;
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(data, rs, off, 0, 0);
        }
        off += step;
        if (off >= wrap) off -= wrap;
      }
    }
  }
  if (acc0[0] + acc1[1] == 12345.f || sink[0] == 0x12345u) out[threadIdx.x] = acc0[3] + (float)sink[1];
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;    // x32 MFMAs per MFMA wave
  const unsigned big = 32u << 20;                       // 32 MiB per workgroup: 8 GiB for 256 workgroups (misses every cache)
  unsigned* buf; float* out; int* seen;
  hipMalloc(&buf, (size_t)big * 256); hipMalloc(&out, 1 << 16); hipMalloc(&seen, 4 * 8 * sizeof(int));
  hipMemset(buf, 0, (size_t)big * 256);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  auto run = [&](int mode, int nmem, unsigned per_wg) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 160 * 1024, 0, mode, 64, nmem, buf, per_wg, out, seen);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 160 * 1024, 0, mode, iters, nmem, buf, per_wg, out, seen);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
  };
  hipMemset(seen, 0, 4 * 8 * sizeof(int));
  run(0, 1, big);
  int h[32]; hipMemcpy(h, seen, sizeof(h), hipMemcpyDeviceToHost);
  printf("{\"wave_to_simd_counts\": [");
  for (int s = 0; s < 4; ++s) { printf("["); for (int w = 0; w < 8; ++w) printf("%d%s", h[s * 8 + w], w < 7 ? "," : ""); printf("]%s", s < 3 ? "," : ""); }
  printf("]}\n");
  const double mf8 = 256.0 * 8 * iters * 32;   // MFMA instructions per launch with all eight waves on MFMAs
  for (unsigned per_wg : {65536u + 65536u, big})
    for (int nmem : {1, 4}) {
      float t0 = run(0, nmem, per_wg), t1 = run(1, nmem, per_wg), t2 = run(2, nmem, per_wg), t8 = run(8, nmem, per_wg),
            t3 = run(3, nmem, per_wg), t4 = run(4, nmem, per_wg), t9 = run(9, nmem, per_wg), t10 = run(10, nmem, per_wg),
            t7 = run(7, nmem, per_wg), t5 = run(5, nmem, per_wg), t6 = run(6, nmem, per_wg);
      const double per_wave = (double)iters * nmem;   // VMEM instructions per wave (modes 1 / 2)
      printf("{\"footprint_per_wg_KiB\": %u, \"vmem_per_wave_per_32_mfma\": %d, \"GB_moved\": %.2f, "
             "\"all8_mfma_ms\": %.3f, \"all8_mfma_tflops\": %.0f, \"all8_plus_own_stores_ms\": %.3f, \"all8_plus_own_loads_ms\": %.3f, "
             "\"first4_mfma_ms\": %.3f, \"first4_mfma_second4_stores_ms\": %.3f, \"first4_mfma_second4_loads_ms\": %.3f, "
             "\"second4_stores_alone_ms\": %.3f, \"second4_loads_alone_ms\": %.3f, "
             "\"simd012_mfma_ms\": %.3f, \"simd012_mfma_simd3_stores_ms\": %.3f, \"simd012_mfma_simd3_loads_ms\": %.3f, "
             "\"ns_per_store_same_wave\": %.1f, \"ns_per_load_same_wave\": %.1f, \"ns_per_store_other_wave_same_simd\": %.1f, "
             "\"ns_per_load_other_wave_same_simd\": %.1f, \"ns_per_store_on_simd3\": %.1f, \"ns_per_load_on_simd3\": %.1f}\n",
             per_wg >> 10, nmem, 256.0 * 8 * per_wave * 1024 / 1e9, t0, mf8 * 32768.0 / (t0 * 1e-3) / 1e12, t1, t2, t8, t3, t4, t9, t10, t7, t5,
             t6, (t1 - t0) * 1e6 / per_wave, (t2 - t0) * 1e6 / per_wave, (t3 - t8) * 1e6 / (2 * per_wave), (t4 - t8) * 1e6 / (2 * per_wave),
             (t5 - t7) * 1e6 / (8 * per_wave), (t6 - t7) * 1e6 / (8 * per_wave));
    }
  return 0;
}
