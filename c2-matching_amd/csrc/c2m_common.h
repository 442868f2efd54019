// c2m_common.h -- shared helpers for the gfx950 kernels (device-side idioms + host-side launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/c2m_hip.h"

namespace c2m {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;  // CDNA wavefront

// Record of the last failing HIP call on this host thread (read back through c2m_last_hip_error()).
void set_last_error(hipError_t e);

inline int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_error(e);
    return C2M_ERR_LAUNCH;
  }
  return C2M_OK;
}

// Optional hipEvent bracket around the dominant kernel of an API call (c2m_profile_enable / _collect).
struct ProfileScope {
  ProfileScope(int kernel_id, hipStream_t st);
  ~ProfileScope();
  int slot;
  hipStream_t st;
};

// Raise a kernel's dynamic-LDS limit once per (kernel instantiation, device): the attribute is per device, and
// nn.DataParallel drives several GPUs from one process.  `done` is a per-instantiation bitmask of devices already set
// (a race only repeats the idempotent call).
inline int ensure_dynamic_lds(const void* fn, size_t bytes, unsigned long long& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess && dev >= 0 && dev < 64 && ((done >> dev) & 1ull)) return C2M_OK;
  if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    set_last_error(e);
    return C2M_ERR_LAUNCH;
  }
  if (dev >= 0 && dev < 64) done |= 1ull << dev;
  return C2M_OK;
}

inline hipStream_t as_stream(c2m_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Bijective XCD-aware remap (guide T1): the dispatcher places block b on XCD b % 8; give every XCD one contiguous
// chunk of the logical tile space so that blocks sharing operands also share a private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  constexpr int NX = 8;
  const int q = nblocks / NX, r = nblocks % NX;
  const int xcd = bid % NX, slot = bid / NX;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// async global -> LDS copy of one dword per lane: LDS destination = wave-uniform base + lane * 4.
__device__ __forceinline__ void glds_b32(const float* gsrc, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

// same, 16 bytes per lane (gfx950 global_load_lds_dwordx4): LDS destination = wave-uniform base + lane * 16
__device__ __forceinline__ void glds_b128(const float* gsrc, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

}  // namespace c2m
