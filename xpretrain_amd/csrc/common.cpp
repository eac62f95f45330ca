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
extern "C" int xp_abi_version(void) { return XP_ABI_VERSION; }
