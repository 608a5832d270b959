// common.cpp -- osh_last_error / osh_version / osh_device_count.
#include "common.h"
#include <cstring>

namespace osh {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }
}  // namespace osh

extern "C" {
const char* osh_last_error(void) { return osh::get_error(); }
const char* osh_version(void) { return "orbslam3_hip 0.1 gfx950"; }
int osh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
}
