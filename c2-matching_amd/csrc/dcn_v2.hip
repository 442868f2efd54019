// dcn_v2.hip -- DCNv2 forward/backward for gfx950 (placeholder entry points; kernels land in the next commit).
#include "c2m_common.h"

extern "C" size_t c2m_dcn_v2_forward_workspace_bytes(int, int, int, int, int, int, int, int) { return 0; }
extern "C" int c2m_dcn_v2_forward_f32(c2m_stream_t, const float*, const float*, const float*, const float*,
                                      const float*, int, int, int, int, int, int, int, int, int, int, int, int, int,
                                      int, float*, void*, size_t) {
  return C2M_ERR_UNSUPPORTED;
}
extern "C" size_t c2m_dcn_v2_backward_workspace_bytes(int, int, int, int, int, int, int, int, int, int, int, int, int,
                                                      int) {
  return 0;
}
extern "C" int c2m_dcn_v2_backward_f32(c2m_stream_t, const float*, const float*, const float*, const float*,
                                       const float*, const float*, int, int, int, int, int, int, int, int, int, int,
                                       int, int, int, int, float*, float*, float*, float*, float*, void*, size_t) {
  return C2M_ERR_UNSUPPORTED;
}
extern "C" int c2m_dcn_fuse_offsets_f32(c2m_stream_t, const float*, const float*, int, int, int, int, int, float*,
                                        float*, double*) {
  return C2M_ERR_UNSUPPORTED;
}
