// Error plumbing + trivial entry points of libarvae_hip.so.
#include <stdarg.h>

#include "common.h"

namespace arvae {

thread_local char g_last_error[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ARVAE_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return ARVAE_OK;
}

}  // namespace arvae

extern "C" int arvae_abi_version(void) { return ARVAE_ABI_VERSION; }

extern "C" const char *arvae_last_error_string(void) { return arvae::g_last_error; }

extern "C" int arvae_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
