// Error plumbing + version for the C ABI (include/hidvae.h).
#include <cstdarg>
#include <cstdio>
#include "../../include/hidvae.h"

static thread_local char g_err[512] = "";

int hv_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *hidvae_last_error(void) { return g_err; }
extern "C" const char *hidvae_version(void) { return "hidvae-mi355 0.1 (gfx950)"; }
