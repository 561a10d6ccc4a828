// core.hip -- error channel and device queries of libctcn.so
#include "common.h"

static thread_local char g_err[512] = "";

void ctcn_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ctcn_version(void) { return 100; }
extern "C" const char *ctcn_last_error(void) { return g_err; }
extern "C" int ctcn_device_cus(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return n;
}
