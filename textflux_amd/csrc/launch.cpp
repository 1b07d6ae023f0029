// Error plumbing shared by all launchers (see launch.h).
#include "launch.h"

#include <cstdarg>
#include <cstdio>

namespace tfx {

static thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("%s: launch failed: %s", what, hipGetErrorString(e));
  return 0;
}

const char* last_error() { return g_err; }

}  // namespace tfx
