// Layout probe for v_mfma_f32_32x32x16_bf16 on gfx950: checks the assumed operand mapping
//   A: lane l holds A[i = l % 32][k = 8 * (l / 32) + e], e = 0..7      B: lane l holds B[k = 8 * (l / 32) + e][j = l % 32]
//   C: lane l, register r holds C[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31]
// with an asymmetric integer-valued A and B (exact in bf16).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* C) {   // A[32][16], B[16][32], C[32][32]
  const int l = threadIdx.x;
  bf16x8 va, vb;
  for (int e = 0; e < 8; ++e) {
    va[e] = (__bf16)A[(l % 32) * 16 + 8 * (l / 32) + e];
    vb[e] = (__bf16)B[(8 * (l / 32) + e) * 32 + (l % 32)];
  }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
int main() {
  std::vector<float> A(32 * 16), B(16 * 32), C(32 * 32), R(32 * 32, 0.f);
  for (int i = 0; i < 32; ++i) for (int kk = 0; kk < 16; ++kk) A[i * 16 + kk] = (float)((i * 3 + kk * 5) % 7 - 3);
  for (int kk = 0; kk < 16; ++kk) for (int j = 0; j < 32; ++j) B[kk * 32 + j] = (float)((kk * 11 + j * 2) % 9 - 4);
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int kk = 0; kk < 16; ++kk) R[i * 32 + j] += A[i * 16 + kk] * B[kk * 32 + j];
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int e = 0; e < 32 * 32; ++e) bad += C[e] != R[e];
  printf("{\"mfma_f32_32x32x16_bf16_layout_mismatches\": %d}\n", bad);
  return bad != 0;
}
