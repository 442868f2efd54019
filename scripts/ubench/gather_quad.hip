// Micro-benchmark (round 6): is a 16-byte-per-lane GATHER cheaper for the CU's texture path when the four lanes of a quad fetch the four
// pieces of ONE 64-byte segment (16 segments per instruction) than when every lane fetches from its own segment (64 segments per
// instruction, the DCNv2 forward's pattern: DESIGN.md 5.2 -- 43 TA cycles per gather instruction, TA 0.95 busy)?  Same bytes, same
// instruction count, throughput-bound (8 independent loads in flight per wave, 8 waves per CU), segments at pseudo-random positions in
// a window of `win` KiB per workgroup (L2-resident; larger than the CU's 32 KiB L1 unless win <= 16).
//   mode 0  own segment per lane: instruction k of a group of 4 reads piece k of the lane's segment        (64 segments / instruction)
//   mode 1  quad-shared: lane 4q+i reads piece i of segment (k, q)                                         (16 segments / instruction)
//   mode 3  as 1, and every group of four instructions is followed by what bringing the pieces back to a one-segment-per-lane layout costs
//           through LDS: 4 ds_write_b128 (piece (segment, i) -> [segment][i], 144-byte segment stride), a wait, 4 ds_read_b128 of the lane's
//           own segment -- the transposition a DCNv2 forward with quad-shared gathers would need in front of its blend
//   mode 2  as 0 with 32-byte segments split over lanes l and l+32 (the real kernel's two half-waves): instruction k reads piece k & 1 of
//           the segment of (l & 31, l >> 5) ... i.e. own 32-byte run per lane, two instructions per run
// build: hipcc --offload-arch=gfx950 -O3 -o gather_quad gather_quad.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned rnd(unsigned x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

template <int MODE>
__global__ void __launch_bounds__(512) k(int n, const unsigned* __restrict__ buf, unsigned win_bytes, unsigned long long* __restrict__ clk, unsigned* __restrict__ out) {
  const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __shared__ __attribute__((aligned(16))) unsigned char tr[8][64 * 144 + 64];   // per wave: 64 segments x (64 + 80 pad) bytes
  const char* base = reinterpret_cast<const char*>(buf) + (size_t)blockIdx.x * win_bytes;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)win_bytes, 0x00020000);
  const unsigned nseg = win_bytes / 64;
  unsigned s = 0x9e3779b9u * (unsigned)(blockIdx.x * 512 + threadIdx.x + 1);
  u32x4 sink = {0u, 0u, 0u, 0u};
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
    u32x4 v[8];
#pragma unroll
    for (int g = 0; g < 2; ++g) {          // two groups of four instructions = what one (tap, group) step of the DCNv2 forward issues
      unsigned off[4];
      if (MODE == 0) {
        s = rnd(s);
        const unsigned seg = s % nseg;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) off[kk] = seg * 64 + kk * 16;
      } else if (MODE == 1) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          s = rnd(s);
          const unsigned sq = __shfl(s, l & ~3, 64);     // the quad's common segment
          off[kk] = (sq % nseg) * 64 + (l & 3) * 16;
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          s = rnd(s);
          const unsigned seg32 = s % (2 * nseg);
          off[2 * kk] = seg32 * 32; off[2 * kk + 1] = seg32 * 32 + 16;
        }
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) v[4 * g + kk] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[kk], 0, 0);
      if (MODE == 3) {
        // instruction kk, lane 4q+i holds piece i of segment 16kk+q -> LDS [16kk+q][i]; then lane l reads the four pieces of segment l
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) *reinterpret_cast<u32x4*>(&tr[w][(16 * kk + (l >> 2)) * 144 + (l & 3) * 16]) = v[4 * g + kk];
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) v[4 * g + kk] = *reinterpret_cast<const u32x4*>(&tr[w][l * 144 + kk * 16]);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) sink ^= v[kk];
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (l == 0 && w == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
  if (sink[0] == 0x12345u) out[threadIdx.x] = sink[1];
}

template <int MODE>
static void run(int n, unsigned* buf, unsigned long long* clk, unsigned* out, int nwg, unsigned win_kib) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE>), dim3(nwg), dim3(512), 0, 0, n, buf, win_kib * 1024u, clk, out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    if (rep < 2) continue;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long* h = (unsigned long long*)malloc(16 * nwg);
    (void)hipMemcpy(h, clk, 16 * nwg, hipMemcpyDeviceToHost);
    double cyc = 0, ref = 0;
    for (int i = 0; i < nwg; ++i) { cyc += (double)h[2 * i]; ref += (double)h[2 * i + 1]; }
    free(h);
    const double ghz = cyc / (ref * 10.0);
    const double instr_per_cu = (double)n * 8.0 * 8.0 * nwg / 256.0;
    const double ns = ms * 1e6 / instr_per_cu;
    printf("{\"mode\": %d, \"window_KiB_per_CU\": %u, \"ms\": %.3f, \"clock_GHz\": %.2f, \"cycles_per_gather_instr_per_CU\": %.1f, \"bytes_per_clk_per_CU\": %.1f}\n",
           MODE, win_kib, ms, ghz, ns * ghz, 1024.0 / (ns * ghz));
    fflush(stdout);
  }
}

int main() {
  const int nwg = 256;
  unsigned* buf; unsigned long long* clk; unsigned* out;
  (void)hipMalloc(&buf, (size_t)nwg * (1u << 20)); (void)hipMalloc(&clk, 16 * nwg); (void)hipMalloc(&out, 4096);
  (void)hipMemset(buf, 1, (size_t)nwg * (1u << 20));
  for (unsigned win : {16u, 128u, 1024u}) {   // L1-resident, L2-resident (4 MiB per XCD), beyond L2 (32 MiB per XCD: MALL / HBM)
    run<0>(2000, buf, clk, out, nwg, win);
    run<2>(2000, buf, clk, out, nwg, win);
    run<1>(2000, buf, clk, out, nwg, win);
    run<3>(2000, buf, clk, out, nwg, win);
  }
  return 0;
}
