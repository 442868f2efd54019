// Micro-benchmark (round 6): can ONE wave hide its own vector-ALU / LDS instructions under its own MFMAs on gfx950?
// One workgroup of 4 waves per CU = one wave per SIMD.  Every wave runs a loop of groups
//     [ NM independent v_mfma_f32_32x32x16_f16 ]  [ K other instructions ]
// in two program orders: "after" (all NM MFMAs, then the K others -- what a sched_barrier-fenced group looks like) and
// "interleaved" (K / NM others behind each MFMA).  If the others hide under the matrix pipe, time stays flat in K until
// K * issue > NM * 32 cycles.  others: 0 = v_fma_f32 (independent chains), 1 = ds_read_b128 + the lgkmcnt wait at the group's end.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_self_overlap mfma_self_overlap.hip ; run: ./mfma_self_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, int ORDER, int KIND, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 1) kern(float* out, int iters) {
  __shared__ f32x4 lds[1024];
  const int l = threadIdx.x & 63;
  lds[threadIdx.x & 1023] = f32x4{1.f, 2.f, 3.f, (float)l};
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (l + e)); b[e] = (_Float16)(0.002f * (l - e)); }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = l + i;
  f32x4 rd[8];
  for (int i = 0; i < 8; ++i) rd[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) f32x4*)lds + l * 16;
  auto other = [&](int i) __attribute__((always_inline)) {
    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i & 7]) : "v"(1.0001f));
    else asm volatile("ds_read_b128 %0, %1" : "=v"(rd[i & 7]) : "v"(la) : "memory");
  };
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ORDER == 0) {
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < K; ++i) other(i);
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < K / 4; ++i) other(m * (K / 4) + i);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (KIND == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  for (int i = 0; i < 8; ++i) s += x[i] + rd[i][0];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int K, int ORDER, int KIND, int WAVES>
static void run(float* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((kern<K, ORDER, KIND, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, 64);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((kern<K, ORDER, KIND, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns_per_group = ms * 1e6 / iters;
  const double tf = 256.0 * WAVES * iters * 4.0 * 32 * 32 * 16 * 2 / (ms * 1e-3) / 1e12;
  printf("{\"waves_per_simd\": %d, \"other\": \"%s\", \"order\": \"%s\", \"K_others_per_4_mfma\": %d, \"ns_per_group\": %.1f, \"mfma_TFLOPs\": %.0f}\n", WAVES / 4,
         KIND == 0 ? "v_fma_f32" : "ds_read_b128", ORDER == 0 ? "after" : "interleaved", K, ns_per_group, tf);
}

int main() {
  float* out;
  hipMalloc(&out, 1 << 20);
  const int iters = 200000;
#define SWEEP(ORDER, KIND, WAVES)                                                                                           \
  run<0, ORDER, KIND, WAVES>(out, iters); run<8, ORDER, KIND, WAVES>(out, iters); run<16, ORDER, KIND, WAVES>(out, iters);   \
  run<24, ORDER, KIND, WAVES>(out, iters); run<32, ORDER, KIND, WAVES>(out, iters); run<48, ORDER, KIND, WAVES>(out, iters);
  SWEEP(0, 0, 4) SWEEP(1, 0, 4) SWEEP(0, 0, 8) SWEEP(1, 0, 8)
  run<0, 0, 1, 4>(out, iters); run<8, 0, 1, 4>(out, iters); run<16, 0, 1, 4>(out, iters);
  run<8, 1, 1, 4>(out, iters); run<16, 1, 1, 4>(out, iters);
  run<8, 0, 1, 8>(out, iters); run<16, 0, 1, 8>(out, iters);
  return 0;
}
