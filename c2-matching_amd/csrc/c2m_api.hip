// c2m_api.hip -- ABI bookkeeping shared by all kernels: version, status strings, last-HIP-error, device arch.
#include <string.h>

#include <mutex>

#include "c2m_common.h"

namespace c2m {
static thread_local hipError_t g_last_error = hipSuccess;
void set_last_error(hipError_t e) { g_last_error = e; }
}  // namespace c2m

namespace c2m {
namespace {
constexpr int kMaxProf = 16384;   // a full restoration forward launches ~150 timed kernels per step
struct ProfState {
  std::mutex mu;
  bool on = false;
  int n = 0;
  hipEvent_t a[kMaxProf], b[kMaxProf];
  bool made[kMaxProf] = {};
  int id[kMaxProf];
} g_prof;
}  // namespace

ProfileScope::ProfileScope(int kernel_id, hipStream_t s) : slot(-1), st(s) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  if (!g_prof.on || g_prof.n >= kMaxProf) return;
  const int i = g_prof.n;
  if (!g_prof.made[i]) {
    if (hipEventCreate(&g_prof.a[i]) != hipSuccess || hipEventCreate(&g_prof.b[i]) != hipSuccess) return;
    g_prof.made[i] = true;
  }
  if (hipEventRecord(g_prof.a[i], st) != hipSuccess) return;
  g_prof.id[i] = kernel_id;
  slot = i;
  g_prof.n = i + 1;
}

ProfileScope::~ProfileScope() {
  if (slot >= 0) (void)hipEventRecord(g_prof.b[slot], st);
}
}  // namespace c2m

extern "C" int c2m_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(c2m::g_prof.mu);
  c2m::g_prof.on = on != 0;
  return C2M_OK;
}

extern "C" int c2m_profile_collect(float* ms, int* kernel_id, int capacity, int* count) {
  if (!ms || !count || capacity < 0) return C2M_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(c2m::g_prof.mu);
  int k = 0;
  for (int i = 0; i < c2m::g_prof.n && k < capacity; ++i) {
    hipError_t e = hipEventSynchronize(c2m::g_prof.b[i]);
    float t = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, c2m::g_prof.a[i], c2m::g_prof.b[i]);
    if (e != hipSuccess) { c2m::set_last_error(e); c2m::g_prof.n = 0; return C2M_ERR_LAUNCH; }
    ms[k] = t;
    if (kernel_id) kernel_id[k] = c2m::g_prof.id[i];
    ++k;
  }
  c2m::g_prof.n = 0;
  *count = k;
  return C2M_OK;
}

extern "C" int c2m_abi_version(void) { return 3; }

extern "C" const char* c2m_status_string(int status) {
  switch (status) {
    case C2M_OK: return "ok";
    case C2M_ERR_INVALID_ARG: return "invalid argument";
    case C2M_ERR_UNSUPPORTED: return "unsupported configuration";
    case C2M_ERR_WORKSPACE: return "workspace missing or too small";
    case C2M_ERR_LAUNCH: return "HIP launch failed (see c2m_last_hip_error)";
    case C2M_ERR_NO_DEVICE: return "no gfx950 device";
    default: return "unknown status";
  }
}

extern "C" const char* c2m_last_hip_error(void) { return hipGetErrorString(c2m::g_last_error); }

extern "C" int c2m_device_arch(char* buf, int buflen) {
  if (!buf || buflen <= 0) return C2M_ERR_INVALID_ARG;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  hipDeviceProp_t prop;
  if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) {
    c2m::set_last_error(e);
    buf[0] = 0;
    return C2M_ERR_NO_DEVICE;
  }
  strncpy(buf, prop.gcnArchName, (size_t)buflen - 1);
  buf[buflen - 1] = 0;
  return C2M_OK;
}
