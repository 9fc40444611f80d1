// dme_capi.hip — error text, ABI version and device queries of libdme_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "dme_common.h"

static thread_local char g_err[512] = "";

int dme_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* dme_last_error(void) { return g_err; }

extern "C" int dme_abi_version(void) { return DME_ABI_VERSION; }

extern "C" int dme_device_cus(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    static thread_local int cached_dev = -1, cached_cus = -1;
    if (dev == cached_dev) return cached_cus;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return -1;
    cached_dev = dev;
    cached_cus = cus;
    return cus;
}
