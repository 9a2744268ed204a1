// Error plumbing + version for the C ABI (include/hidvae.h).
#include <cstdarg>
#include <cstdio>
#include "../../include/hidvae.h"
#include "rules.h"

static thread_local char g_err[512] = "";

int hv_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *hidvae_last_error(void) { return g_err; }
extern "C" const char *hidvae_version(void) { return "hidvae-mi355 0.2 (gfx950)"; }

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Workspace sizes of every entry point that takes a caller-allocated workspace / scratch buffer (the caller owns all
// device memory; this is where it learns how much).  The formulas are the ones the launch code assumes -- keep them in one place.
extern "C" int hidvae_query_workspace(int op, const int64_t *d, int n, int64_t *bytes) {
    if (bytes == nullptr || d == nullptr) return hv_fail(HIDVAE_EINVAL, "query_workspace: null argument");
    auto need = [&](int k) { return n >= k; };
    int64_t fl = -1;
    switch (op) {
    case HIDVAE_WS_GEMM: {  // M, N, K, split_k
        if (!need(4)) break;
        const int64_t M = d[0], N = d[1], K = d[2], sk = d[3];
        if (sk > 1 && cdiv(M, 32) * cdiv(N, 32) >= 2048) fl = sk * M * N;
        else if (sk == 0 && K >= 4096) fl = 16 * M * N;
        else fl = 0;
        break;
    }
    case HIDVAE_WS_LINEAR_BWD: {  // B, n_out, n_in, has_bias
        if (!need(4)) break;
        fl = d[3] ? cdiv(d[0], 64) * d[1] : 0;
        if (d[0] >= 4096 && 16 * d[1] * d[2] > fl) fl = 16 * d[1] * d[2];
        // shapes of the balanced kernel: its arrival counters lead the buffer, everything else (its partial tiles, or the scratch of
        // the other paths when a call of this shape does not take it -- dX == NULL halves the work) comes after them
        if (hv_lbwd_balanced(d[0], d[1], d[2])) fl = HV_SK_COUNTERS + (fl > HV_SK_WS_FLOATS - HV_SK_COUNTERS ? fl : HV_SK_WS_FLOATS - HV_SK_COUNTERS);
        break;
    }
    case HIDVAE_WS_LINEAR_BWD_ZEROED:  // B, n_out, n_in, has_bias: leading bytes that must be zero on entry (and are zero on return)
        if (need(3)) fl = hv_lbwd_balanced(d[0], d[1], d[2]) ? HV_SK_COUNTERS : 0;
        break;
    case HIDVAE_WS_COLSUM: if (need(2)) fl = cdiv(d[0], 64) * d[1]; break;                             // M, N
    case HIDVAE_WS_CODEBOOK_GRAD: if (need(3)) fl = d[0] > 2048 ? d[1] * d[2] * cdiv(d[0], 2048) * 32 : 0; break;  // B, L, K  (the slabbed form belongs to the 32-wide kernels)
    case HIDVAE_WS_LAYERNORM_PARAM_GRAD: if (need(2)) fl = 2 * cdiv(d[0], 128) * d[1]; break;          // M, N
    case HIDVAE_WS_LAYERNORM_BWD_ALL: if (need(2)) fl = 2 * cdiv(d[0], 4) * d[1]; break;               // M, N
    case HIDVAE_WS_BATCHNORM_FWD: if (need(2)) fl = 3 * cdiv(d[0], 64) * d[1]; break;                  // M, N
    case HIDVAE_WS_BATCHNORM_BWD: if (need(2)) fl = 2 * cdiv(d[0], 32) * d[1]; break;                  // M, N
    case HIDVAE_WS_ID_CENSUS: if (need(1)) { *bytes = (4 * d[0] + 3) * 8; return HIDVAE_OK; } break;   // B   (int64 slots, zero-filled once)
    case HIDVAE_WS_RQ_FORWARD: if (need(3)) fl = d[0] >= 65536 ? d[0] + 4 : 0; break;                       // B, L, K  (counter + item list)
    case HIDVAE_WS_KMEANS: if (need(2)) fl = d[1]; break;                                               // N, K
    case HIDVAE_WS_TAG_LOSS: if (need(2)) fl = 2 * d[0] + d[0] * d[1]; break;                           // B, C  (row_loss, row_hit, zbuf)
    default: return hv_fail(HIDVAE_EINVAL, "query_workspace: unknown op %d", op);
    }
    if (fl < 0) return hv_fail(HIDVAE_EINVAL, "query_workspace: op %d needs more dimensions than %d", op, n);
    *bytes = fl * 4;
    return HIDVAE_OK;
}
