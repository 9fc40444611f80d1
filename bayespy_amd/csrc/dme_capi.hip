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

static thread_local int g_cu_limit = 0;

extern "C" int dme_set_cu_limit(int cus) {
    if (cus < 0) return dme_set_error(DME_ERR_INVALID_ARG, "CU limit must be >= 0 (0 = the device's)");
    g_cu_limit = cus;
    return DME_OK;
}

extern "C" int dme_device_cus(void) {
    if (g_cu_limit > 0) return g_cu_limit;
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

// A stream whose kernels run only on the CUs whose bits are set in `mask` (nwords 32-bit words; bit i
// = CU i in the runtime's numbering): hipExtStreamCreateWithCUMask.
extern "C" int dme_stream_create_masked(const uint32_t* mask, int nwords, void** stream) {
    DME_REQUIRE(stream != nullptr && mask != nullptr && nwords >= 1, "bad arguments");
    hipStream_t s = nullptr;
    DME_HIP_CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask));
    *stream = (void*)s;
    return DME_OK;
}

extern "C" int dme_stream_destroy(void* stream) {
    DME_HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
    return DME_OK;
}
