// abi.hip -- error reporting and ABI version for liblidbox_hip.so
#include "common.h"

static thread_local char g_err[512] = "";

void lidbox_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* lidbox_hip_last_error(void) { return g_err; }
extern "C" int lidbox_hip_abi_version(void) { return LIDBOX_HIP_ABI_VERSION; }
