// Library-level entry points: version and per-thread error string.
#include <stdarg.h>

#include "common.hpp"

namespace immoco {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace immoco

extern "C" int immoco_version(void) { return 100; }  // 0.1.0

extern "C" int immoco_last_error(char* buf, size_t n) {
  if (!buf || n == 0) return IMMOCO_E_INVALID;
  strncpy(buf, immoco::g_err, n - 1);
  buf[n - 1] = '\0';
  return IMMOCO_OK;
}
