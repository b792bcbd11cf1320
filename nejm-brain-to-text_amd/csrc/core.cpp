// core.cpp — error plumbing shared by every entry point of libb2t_hip.so.
#include <stdarg.h>
#include "common.h"

namespace b2t {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
  return 1;
}
}  // namespace b2t

extern "C" int b2t_version(void) { return B2T_VERSION; }
extern "C" const char* b2t_last_error(void) { return b2t::g_err; }
