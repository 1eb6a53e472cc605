// thread-local error string + version for the C ABI
#include <stdarg.h>
#include <stdio.h>
#include "mrfa_hip.h"

static thread_local char g_err[512] = "";

void mrfa_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mrfa_last_error(void) { return g_err; }
extern "C" int mrfa_version(void) { return MRFA_ABI_VERSION; }
