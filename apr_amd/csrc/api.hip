// Error reporting and library-level entry points of libapr_hip.so.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void apr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

APR_API const char* apr_last_error(void) { return g_err; }

APR_API int apr_version(void) { return 100; }

APR_API int apr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
