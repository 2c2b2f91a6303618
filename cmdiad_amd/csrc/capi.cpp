// Error channel + version of the C ABI (include/cmdiad_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/cmdiad_hip.h"

static thread_local char g_err[512] = "";

void cmdiad_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* cmdiad_last_error(void) { return g_err; }
extern "C" int cmdiad_abi_version(void) { return 1; }
