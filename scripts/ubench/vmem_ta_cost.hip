// Micro-benchmark (round 6): what does ONE 16-byte-per-lane vector-memory instruction cost the CU's texture addresser / L1 path on gfx950,
// by kind (store / load), address pattern and where the bytes live -- with nothing else running?  (DESIGN.md 6.2 / 6.9 derived "a channels-last
// store holds the TA for ~143 cycles, ~7 B/clk/CU" from whole kernels; this isolates it.)
// One workgroup of 8 waves per CU, every wave issues `n` instructions of one kind, back to back, waits at the end.
//   pattern 0  channels-last convolution output: lanes l and l+32 write the two 16-byte halves of a 32-byte run, runs at a 256-byte pitch
//   pattern 1  quad runs: lanes 4q..4q+3 cover 64 contiguous bytes, runs at a 256-byte pitch
//   pattern 2  one contiguous KiB per instruction
//   pattern 3  every lane its own 128-byte line (a gather: 64 lines per instruction)
//   footprint  L2: each wave cycles through 64 KiB of its own (stays in the XCD's L2);  HBM: strides through 32 MiB per workgroup
// Reported per (kind, pattern, footprint): ns per instruction per CU (wall time / instructions issued by one CU's share), bytes per clock and CU
// at the clock the kernel measured for itself (s_memtime against s_memrealtime), GB/s over the chip.  Run under
// `rocprofv3 --pmc TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE` for the TA's own view (kernel names carry kind and pattern).
// build: hipcc --offload-arch=gfx950 -O3 -o vmem_ta_cost vmem_ta_cost.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int PAT>   // KIND 0 store, 1 load
__global__ void __launch_bounds__(512) k(int n, unsigned* __restrict__ buf, unsigned bytes_per_wg, unsigned step, unsigned long long* __restrict__ clk, unsigned* __restrict__ out) {
  extern __shared__ float lds[];
  const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* base = reinterpret_cast<char*>(buf) + (size_t)blockIdx.x * bytes_per_wg;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes_per_wg, 0x00020000);
  unsigned lane_off;
  if (PAT == 0) lane_off = (unsigned)((l & 31) * 256 + (l >> 5) * 16);
  else if (PAT == 1) lane_off = (unsigned)((l >> 2) * 256 + (l & 3) * 16);
  else if (PAT == 2) lane_off = (unsigned)(l * 16);
  else lane_off = (unsigned)(l * 128);
  // a wave's own window of 8 KiB inside the workgroup's region (64 KiB per workgroup and position); `step` moves the position
  unsigned off = lane_off + (unsigned)w * 8192u;
  const unsigned wrap = bytes_per_wg - 65536u;
  u32x4 data = {(unsigned)l, 1u, 2u, 3u}, sink = {0u, 0u, 0u, 0u};
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
    if (KIND == 1) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0); sink ^= v; }
    else __builtin_amdgcn_raw_buffer_store_b128(data, rs, off, 0, 0);
    off += step;
    if (off >= wrap + lane_off + (unsigned)w * 8192u) off -= wrap;
  }
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (l == 0 && w == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
  if (sink[0] == 0x12345u) out[threadIdx.x] = sink[1];
}

template <int KIND, int PAT>
static void run(int n, unsigned* buf, unsigned long long* clk, unsigned* out, int nwg, bool hbm) {
  const unsigned bytes_per_wg = hbm ? (32u << 20) : (128u << 10)   /* L2: 64 KiB touched per CU = 2 MiB per XCD, twice the CU's L1 */, step = hbm ? 65536u : 0u;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND, PAT>), dim3(nwg), dim3(512), 70 * 1024, 0, n, buf, bytes_per_wg, step, clk, out);
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
    cyc /= nwg; ref /= nwg;
    const double ghz = cyc / (ref * 10.0);                  // s_memrealtime ticks at 100 MHz
    const double instr_per_cu = (double)n * 8.0 * nwg / 256.0;
    const double ns_per_instr = ms * 1e6 / instr_per_cu;
    printf("{\"kind\": \"%s\", \"pattern\": %d, \"where\": \"%s\", \"ms\": %.3f, \"clock_GHz\": %.2f, \"ns_per_instr_per_CU\": %.1f, \"cycles_per_instr_per_CU\": %.1f, "
           "\"bytes_per_clk_per_CU\": %.1f, \"chip_GBps\": %.0f}\n", KIND ? "load" : "store", PAT, hbm ? "HBM" : "L2", ms, ghz, ns_per_instr, ns_per_instr * ghz,
           1024.0 / (ns_per_instr * ghz), (double)n * 8.0 * nwg * 1024.0 / (ms * 1e6));
    fflush(stdout);
  }
}

int main(int argc, char** argv) {
  const int nwg = 256;   // one workgroup of 8 waves per CU
  const int only = argc > 1 ? atoi(argv[1]) : -1;   // 0: L2 footprint only, 1: HBM only (for the counter passes)
  unsigned* buf; unsigned long long* clk; unsigned* out;
  if (hipMalloc(&buf, (size_t)nwg * (32u << 20)) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }   // 16 GiB
  (void)hipMalloc(&clk, 16 * nwg); (void)hipMalloc(&out, 4096);
  (void)hipMemset(buf, 0, (size_t)nwg * (32u << 20));
  for (int hbm = 0; hbm < 2; ++hbm) {
    if (only >= 0 && hbm != only) continue;
    const int n = hbm ? 448 : 4096;   // HBM: 448 positions x 64 KiB < 32 MiB: every byte touched once
    run<0, 0>(n, buf, clk, out, nwg, hbm); run<0, 1>(n, buf, clk, out, nwg, hbm); run<0, 2>(n, buf, clk, out, nwg, hbm); run<0, 3>(n, buf, clk, out, nwg, hbm);
    run<1, 0>(n, buf, clk, out, nwg, hbm); run<1, 1>(n, buf, clk, out, nwg, hbm); run<1, 2>(n, buf, clk, out, nwg, hbm); run<1, 3>(n, buf, clk, out, nwg, hbm);
  }
  return 0;
}
