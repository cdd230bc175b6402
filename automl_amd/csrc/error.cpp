// Thread-local error message of the C ABI (edet_last_error).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/edet_hip.h"

static thread_local char g_err[512] = "";

void edet_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* edet_last_error(void) { return g_err; }
extern "C" int edet_version(void) { return 1; }
