// Error string + version of the C-ABI library.
#include <stdarg.h>

#include "common.h"

namespace mafed {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mafed

extern "C" int mafed_version(void) { return 100; }
extern "C" const char* mafed_last_error_string(void) { return mafed::g_err; }
