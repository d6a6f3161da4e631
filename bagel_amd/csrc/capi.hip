// Error reporting shared by every C-ABI entry point of libbagel_hip.so.
// Convention: entry points return 0 on success, non-zero on failure; the message of the most recent
// failure on the calling thread is available from bagel_last_error().
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_last_error[512] = "";

void bagel_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

extern "C" const char* bagel_last_error(void) { return g_last_error; }

extern "C" int bagel_abi_version(void) { return 1; }
