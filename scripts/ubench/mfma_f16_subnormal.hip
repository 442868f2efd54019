// Micro-benchmark (round 6): does v_mfma_f32_32x32x16_f16 honour f16 SUBNORMAL inputs on gfx950, or flush them to zero?
// A[row][k] = 2^-20 (f16 bits 0x0010: subnormal) at k = 0 for every row, B[k][col] = 1024 at k = 0: every output should be 2^-10 if
// subnormals are multiplied exactly, 0 if flushed.  Also: a product of two subnormals (2^-20 * 2^-20 = 2^-40, representable in fp32).
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_f16_subnormal mfma_f16_subnormal.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__global__ void k(float* out, unsigned short abits, unsigned short bbits) {
  const int l = threadIdx.x;
  u16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
  if (l < 32) { a[0] = abits; b[0] = bbits; }   // k = 0 lives in element 0 of the lanes 0..31 (k half 0)
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  out[l] = acc[0];
}

int main() {
  float* d; float h[64];
  (void)hipMalloc(&d, 256);
  struct { unsigned short a, b; const char* what; double want; } cases[] = {
    {0x0010, 0x6400, "2^-20 (subnormal) x 1024", 9.765625e-4}, {0x0001, 0x7800, "2^-24 (smallest subnormal) x 32768", 0.001953125},
    {0x0010, 0x0010, "2^-20 x 2^-20 (both subnormal)", 9.094947017729282e-13}, {0x03ff, 0x3c00, "largest subnormal x 1", 6.097555160522461e-05},
    {0x0400, 0x3c00, "2^-14 (smallest normal) x 1", 6.103515625e-05}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c.a, c.b);
    (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("{\"case\": \"%s\", \"got\": %.10e, \"exact\": %.10e, \"honoured\": %s}\n", c.what, (double)h[0], c.want, (double)h[0] == c.want ? "true" : "false");
  }
  return 0;
}
