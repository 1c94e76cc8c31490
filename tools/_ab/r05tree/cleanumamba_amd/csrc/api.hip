// ABI bookkeeping: version, thread-local error string.
#include <string.h>
#include "common.h"

static thread_local char g_err[256] = "";

extern "C" void cum_set_error(const char *msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

extern "C" const char *cum_last_error(void) { return g_err; }

extern "C" int cum_abi_version(void) { return CUM_ABI_VERSION; }
