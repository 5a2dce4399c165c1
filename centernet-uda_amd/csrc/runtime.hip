// Error reporting and ABI bookkeeping shared by every entry point.
#include <stdarg.h>

#include "common.h"

namespace cnuda {
namespace {
thread_local char g_error[512] = "";
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}
}  // namespace cnuda

extern "C" int cnuda_abi_version(void) { return CNUDA_ABI_VERSION; }
extern "C" const char* cnuda_last_error(void) { return cnuda::g_error; }
