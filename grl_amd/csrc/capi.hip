// Error plumbing of the C ABI (include/grl_hip.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/grl_hip.h"
#include "common.h"

static thread_local char g_err[512] = "";

int grl_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int grl_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return grl_fail(GRL_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return GRL_OK;
}

extern "C" const char* grl_last_error(void) { return g_err; }
extern "C" int grl_abi_version(void) { return GRL_ABI_VERSION; }
