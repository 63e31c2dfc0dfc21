// Thread-local error text for the C ABI (include/gnnome_hip.h: gnnome_last_error).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/gnnome_hip.h"

namespace gnnome {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace gnnome

extern "C" int gnnome_abi_version(void) { return GNNOME_ABI_VERSION; }
extern "C" const char* gnnome_last_error(void) { return gnnome::g_err; }
