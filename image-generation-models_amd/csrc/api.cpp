// Error plumbing of the C ABI (include/mi_ddpm.h).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/mi_ddpm.h"

static thread_local char g_err[512] = "";

int mi_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int mi_abi_version(void) { return MI_ABI_VERSION; }
extern "C" const char* mi_last_error(void) { return g_err; }
