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
extern "C" int cmdiad_abi_version(void) { return 6; }

// 1 in the test-only build (make ab, -DCMDIAD_AB_VARIANTS) that also carries the superseded kernel formulations the A/B
// tools and the variant parity tests select through CMDIAD_L2_TILE / CMDIAD_FPS_PK / CMDIAD_KNN_WAVE / CMDIAD_GEMM_WIDE /
// CMDIAD_STAGE1_ONCE; 0 in the production library, which contains one formulation of every kernel.
extern "C" int cmdiad_has_ab_variants(void)
{
#ifdef CMDIAD_AB_VARIANTS
    return 1;
#else
    return 0;
#endif
}
