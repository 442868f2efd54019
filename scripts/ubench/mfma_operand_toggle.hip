// Micro-benchmark: how much of the f16 MFMA's data-dependent power (mfma_power_share.hip: 1.86 PF with constant operands, 1.53
// with operands that change every instruction) comes from the A side, the B side, and from how often they change?  Eight waves per
// CU, two accumulator chains per wave, random f16 register sets; per pattern the sustained TFLOP/s.
//   pattern 0  A and B constant
//   pattern 1  A and B change every MFMA (4 sets each, rotating)
//   pattern 2  A changes every MFMA, B constant           pattern 3  B changes every MFMA, A constant
//   pattern 4  A changes every 2nd MFMA, B every MFMA      pattern 5  A every 4th, B every MFMA
//   pattern 6  A and B change every 2nd MFMA (pairs of identical instructions on the two chains)
//   pattern 7  as 1 with both operands all zero  (lower bound of the data term)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int PAT>
__global__ void __launch_bounds__(512, 2) k(int iters, float* __restrict__ out) {
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  f16x8 a[4], b[4];
  for (int s = 0; s < 4; ++s)
    for (int e = 0; e < 8; ++e) {
      const unsigned ha = hash32((unsigned)(threadIdx.x * 64 + s * 8 + e) * 2654435761u + 17u), hb = hash32(ha + 0x9e3779b9u);
      a[s][e] = PAT == 7 ? (_Float16)0.0f : (_Float16)((float)(int)(ha & 0xffff) * (1.0f / 32768.0f) - 1.0f);
      b[s][e] = PAT == 7 ? (_Float16)0.0f : (_Float16)((float)(int)(hb & 0xffff) * (1.0f / 32768.0f) - 1.0f);
    }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 32; ++t) {   // MFMA number t of the iteration goes to chain t & 1
      const int ia = PAT == 0 || PAT == 3 ? 0 : PAT == 4 || PAT == 6 ? (t >> 1) & 3 : PAT == 5 ? (t >> 2) & 3 : t & 3;
      const int ib = PAT == 0 || PAT == 2 ? 0 : PAT == 6 ? ((t >> 1) + 1) & 3 : (t + 1) & 3;
      if (t & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ia], b[ib], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ia], b[ib], acc0, 0, 0, 0);
    }
  }
  if (acc0[0] + acc1[1] == 12345.f) out[threadIdx.x] = acc0[3];
}

template <int PAT>
double run(int iters, float* out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(512), 0, 0, 64, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(512), 0, 0, iters, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return 256.0 * 8 * iters * 32 * 32768.0 / (ms * 1e-3) / 1e12;
}

int main() {
  float* out; hipMalloc(&out, 1 << 16);
  const int iters = 4000;
  for (int rep = 0; rep < 5; ++rep)
    printf("{\"tflops\": {\"const\": %.0f, \"A_and_B_every_mfma\": %.0f, \"A_every_B_const\": %.0f, \"B_every_A_const\": %.0f, "
           "\"A_every_2nd_B_every\": %.0f, \"A_every_4th_B_every\": %.0f, \"A_and_B_every_2nd\": %.0f, \"all_zero\": %.0f}}\n",
           run<0>(iters, out), run<1>(iters, out), run<2>(iters, out), run<3>(iters, out), run<4>(iters, out), run<5>(iters, out),
           run<6>(iters, out), run<7>(iters, out));
  return 0;
}
