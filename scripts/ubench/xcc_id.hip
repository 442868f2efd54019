// Micro-benchmark (round 6): which XCD does workgroup b of a 1-D grid run on?  Reads HW_REG_XCC_ID (gfx940+) and compares it with the
// b % 8 round-robin the XCD-aware tile orders assume (c2m_common.h xcd_remap).  build: hipcc --offload-arch=gfx950 -O2 -o xcc_id xcc_id.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  const int x = __builtin_amdgcn_s_getreg(20 | (3 << 11));   // hwreg(HW_REG_XCC_ID, 0, 4)
  if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
  for (int n : {512, 2560, 8192, 100}) {
    int* d; (void)hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 64 * 1024, 0, d);
    int* h = new int[n];
    (void)hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    int agree = 0, hist[16] = {0};
    for (int b = 0; b < n; ++b) { agree += (h[b] & 15) == (b % 8); hist[h[b] & 15]++; }
    printf("{\"grid\": %d, \"xcc_id_equals_block_mod_8\": %d, \"workgroups_per_xcc\": [%d,%d,%d,%d,%d,%d,%d,%d], \"first16\": [", n, agree, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
    for (int b = 0; b < 16 && b < n; ++b) printf("%d%s", h[b], b == 15 ? "" : ",");
    printf("]}\n");
  }
  return 0;
}
