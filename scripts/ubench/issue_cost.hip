// Micro-benchmark: what does ONE extra instruction of a given kind cost a saturated fp32 MFMA stream on gfx950?
// Workgroup = 8 waves (2 per SIMD), 1 workgroup per CU (160 KiB dynamic LDS, like the correlation kernel).  Every wave
// runs a dependent v_mfma_f32_32x32x2_f32 chain; per 32 MFMAs it also issues NX instructions of kind KIND, spread one
// after every (32/NX)-th MFMA.  cost = (T(NX) - T(0)) * clock / (instructions per wave * 2 waves per SIMD)  [cycles of
// SIMD time per instruction].
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { K_NONE, K_VADD, K_VPKADD, K_VCNDMASK, K_VCMP, K_VADD_DPP, K_PERMLANE, K_DSREAD, K_DSREAD2ST64, K_DSREAD128,
            K_DSWRITE, K_DSWRITE2, K_SALU, K_VADDU32, K_GLDS, K_NKIND };
static const char* kind_name[] = {"none", "v_add_f32", "v_pk_add_f32", "v_cndmask_b32", "v_cmp_gt_f32", "v_add_f32_dpp",
                                  "v_permlane32_swap", "ds_read_b32", "ds_read2st64_b32", "ds_read_b128", "ds_write_b32",
                                  "ds_write2_b32", "s_add_u32", "v_add_u32", "global_load_lds_dword"};

template <int KIND>
__device__ __forceinline__ void extra(float& x, f32x2& p, unsigned lds_lane, const float* g, float* lds_wave, float4& q) {
  if constexpr (KIND == K_VADD) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x));
  if constexpr (KIND == K_VPKADD) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p));
  if constexpr (KIND == K_VCNDMASK) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(x) : : );
  if constexpr (KIND == K_VCMP) asm volatile("v_cmp_gt_f32 vcc, %0, %0" : : "v"(x) : "vcc");
  if constexpr (KIND == K_VADD_DPP) asm volatile("v_add_f32_dpp %0, %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x));
  if constexpr (KIND == K_PERMLANE) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(p.x));
  if constexpr (KIND == K_DSREAD) asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(lds_lane) : "memory");
  if constexpr (KIND == K_DSREAD2ST64) asm volatile("ds_read2st64_b32 %0, %1 offset1:1" : "=v"(p) : "v"(lds_lane) : "memory");
  if constexpr (KIND == K_DSREAD128) asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"((lds_lane * 4u) & 0xffffu) : "memory");
  if constexpr (KIND == K_DSWRITE) asm volatile("ds_write_b32 %0, %1" : : "v"(lds_lane), "v"(x) : "memory");
  if constexpr (KIND == K_DSWRITE2) asm volatile("ds_write2_b32 %0, %1, %1 offset1:64" : : "v"(lds_lane), "v"(x) : "memory");
  if constexpr (KIND == K_SALU) asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");
  if constexpr (KIND == K_VADDU32) asm volatile("v_add_u32 %0, %0, %0" : "+v"(x));
  if constexpr (KIND == K_GLDS)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave, 4, 0, 0);
}

template <int KIND, int NX>
__global__ void __launch_bounds__(512, 2) k(float* out, const float* gsrc, int iters) {
  extern __shared__ float lds[];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float a = 1.0f + l, b = 0.5f, x = l;
  f32x2 p = {1.f, 2.f};
  float4 q = {0, 0, 0, 0};
  const unsigned lds_lane = (w * 1024 + l) * 4;  // byte address (dynamic LDS starts at 0)
  float* lds_wave = lds + 16384 + w * 64;
  const float* g = gsrc + (blockIdx.x * 512 + threadIdx.x);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      if constexpr (NX > 0) {
        constexpr int every = 32 / NX;
        if (t % every == every - 1) {
          __builtin_amdgcn_sched_barrier(0);
          extra<KIND>(x, p, lds_lane, g, lds_wave, q);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (KIND == K_DSREAD || KIND == K_DSREAD2ST64 || KIND == K_DSREAD128 || KIND == K_DSWRITE || KIND == K_DSWRITE2)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (KIND == K_GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (acc[0] + x + p.x + p.y + q.x == 12345.f) out[threadIdx.x] = acc[3];
}

template <int KIND, int NX>
float run(float* d, const float* g, int iters) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND, NX>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<KIND, NX>), dim3(256), dim3(512), 160 * 1024, 0, d, g, 16);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<KIND, NX>), dim3(256), dim3(512), 160 * 1024, 0, d, g, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}

template <int KIND>
void report(float* d, const float* g, int iters, float t0, double ghz) {
  const float t8 = run<KIND, 8>(d, g, iters), t16 = run<KIND, 16>(d, g, iters);
  // per SIMD: 2 waves x iters x NX instructions
  const double c8 = (t8 - t0) * 1e-3 * ghz * 1e9 / (2.0 * iters * 8), c16 = (t16 - t0) * 1e-3 * ghz * 1e9 / (2.0 * iters * 16);
  printf("{\"kind\": \"%s\", \"ms_8_per_32mfma\": %.3f, \"ms_16_per_32mfma\": %.3f, \"cycles_per_instr_at8\": %.2f, \"cycles_per_instr_at16\": %.2f}\n",
         kind_name[KIND], t8, t16, c8, c16);
}

int main() {
  float *d, *g;
  hipMalloc(&d, 1 << 20);
  hipMalloc(&g, 256 * 512 * 4);
  hipMemset(g, 0, 256 * 512 * 4);
  const int iters = 8000;
  const float t0 = run<K_NONE, 0>(d, g, iters);
  // clock from the pure MFMA stream: 2 waves x iters x 32 MFMAs x 64 cycles per SIMD
  const double ghz = 2.0 * iters * 32 * 64 / (t0 * 1e-3) / 1e9;
  printf("{\"baseline_ms\": %.3f, \"implied_clock_ghz_if_mfma_bound\": %.3f}\n", t0, ghz);
  report<K_VADD>(d, g, iters, t0, ghz);
  report<K_VPKADD>(d, g, iters, t0, ghz);
  report<K_VCNDMASK>(d, g, iters, t0, ghz);
  report<K_VCMP>(d, g, iters, t0, ghz);
  report<K_VADD_DPP>(d, g, iters, t0, ghz);
  report<K_PERMLANE>(d, g, iters, t0, ghz);
  report<K_VADDU32>(d, g, iters, t0, ghz);
  report<K_SALU>(d, g, iters, t0, ghz);
  report<K_DSREAD>(d, g, iters, t0, ghz);
  report<K_DSREAD2ST64>(d, g, iters, t0, ghz);
  report<K_DSREAD128>(d, g, iters, t0, ghz);
  report<K_DSWRITE>(d, g, iters, t0, ghz);
  report<K_DSWRITE2>(d, g, iters, t0, ghz);
  report<K_GLDS>(d, g, iters, t0, ghz);
  return 0;
}
