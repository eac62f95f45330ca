// Host-side helpers shared by every translation unit of libxpretrain_hip.so.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void xp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* xp_last_error(void) { return g_err; }

// XPRETRAIN_DEBUG=flag[,flag...]: the one environment variable of the debug / test facilities (read at every query, so a tool can
// flip a flag between two calls).  C side: gemm_no_glds, gemm_slow_epi (128x128 family: register-staged loader / generic epilogue --
// tools/race_repro.py), dw_tile_major (split-K launches on the (tile, z) grid instead of the chunk-major 1-D grid: the bit-identity
// test), attn_bwd_split (the dQ / dK-dV kernel pair instead of the fused attention backward: A/B and the cross-check test),
// attn_bwd_static (fused attention backward without its problem counter).  MEASUREMENT ONLY -- wrong results, timing probes of
// round 6: fc1_no_pre (the layer forward drops fc1's second output), skip_splitk_reduce, no_wgrad_join.
// Python side (xpretrain_amd/_lib.py): sync, op_by_op, no_comm.
bool xp_debug_flag(const char* name) {
  const char* env = getenv("XPRETRAIN_DEBUG");
  if (!env || !*env) return false;
  const size_t n = strlen(name);
  for (const char* p = env; (p = strstr(p, name)) != nullptr; p += n) {
    const bool left = p == env || p[-1] == ',', right = p[n] == '\0' || p[n] == ',';
    if (left && right) return true;
  }
  return false;
}
extern "C" int xp_abi_version(void) { return XP_ABI_VERSION; }
