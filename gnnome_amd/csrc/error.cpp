// Thread-local error text for the C ABI (include/gnnome_hip.h: gnnome_last_error).
#include <stdarg.h>
#include <stdio.h>

#include <hip/hip_runtime.h>

#include "../../include/gnnome_hip.h"

namespace gnnome {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
constexpr int kTuneSlots = 16;   // common.h: kTuneCount
static int g_tuning[kTuneSlots] = {0};
int tuning(int key) { return (key >= 0 && key < kTuneSlots) ? g_tuning[key] : 0; }

int persistent_grid() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cus = cus > 256 ? 256 : cus;
        cached[dev] = cus >= 8 ? cus / 8 * 8 : 8;
    }
    return cached[dev];
}
}  // namespace gnnome

extern "C" int gnnome_set_tuning(int key, int value) {
    if (key < 0 || key >= gnnome::kTuneSlots) {
        gnnome::set_error("set_tuning: key %d out of range", key);
        return GNNOME_EINVAL;
    }
    gnnome::g_tuning[key] = value;
    return GNNOME_OK;
}

extern "C" int gnnome_abi_version(void) { return GNNOME_ABI_VERSION; }
extern "C" const char* gnnome_last_error(void) { return gnnome::g_err; }
