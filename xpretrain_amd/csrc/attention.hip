#include "common.h"
extern "C" size_t xp_attn_workspace_bytes(int32_t, int64_t, int64_t, int64_t, int64_t, int64_t) { return 0; }
extern "C" int xp_attn_fwd(const void*, int64_t, void*, int64_t, float*, const int64_t*, int32_t, int64_t, int64_t, int64_t,
                           int64_t, int64_t, int64_t, int32_t, void*, size_t, void*) {
  xp_set_error("xp_attn_fwd: not built yet"); return XP_ERR_UNSUPPORTED;
}
extern "C" int xp_attn_bwd(const void*, int64_t, const void*, const void*, int64_t, const float*, const int64_t*, void*, float,
                           int32_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int32_t, void*, size_t, void*) {
  xp_set_error("xp_attn_bwd: not built yet"); return XP_ERR_UNSUPPORTED;
}
