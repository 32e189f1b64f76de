// C-ABI plumbing: version, thread-local error string.
#include "hugs_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

extern "C" void hugs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
extern "C" const char* hugs_last_error(void) { return g_err; }
extern "C" int hugs_version(void) { return 10001; }  // 1.00.01
extern "C" int hugs_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
