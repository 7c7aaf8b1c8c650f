// Error channel of the C ABI (thread-local message, returned by otvm_last_error()).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/otvm_hip.h"

static thread_local char g_err[512] = "";

void otvm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* otvm_last_error(void) { return g_err; }
extern "C" int otvm_abi_version(void) { return OTVM_ABI_VERSION; }
