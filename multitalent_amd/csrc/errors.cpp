// Error plumbing of the C ABI: thread-local last-error string (never throws across the boundary).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/mtseg.h"

static thread_local char g_err[512] = "";

void mt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* mt_last_error(void) { return g_err; }
extern "C" int mt_abi_version(void) { return MT_ABI_VERSION; }
