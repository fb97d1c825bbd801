// ABI bookkeeping: version, thread-local error text, device query.
#include "common.h"

#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void made_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int made_abi_version(void) { return MADE_ABI_VERSION; }

extern "C" const char* made_last_error(void) { return g_err; }

extern "C" int made_device_info(char* name, int name_len, int* cu_count, int* is_gfx950) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        made_set_error("made_device_info: no HIP device: %s", hipGetErrorString(e));
        return MADE_ERR_HIP;
    }
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) {
        made_set_error("made_device_info: %s", hipGetErrorString(e));
        return MADE_ERR_HIP;
    }
    if (name && name_len > 0) {
        strncpy(name, p.gcnArchName, (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (is_gfx950) *is_gfx950 = strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
    return MADE_OK;
}
