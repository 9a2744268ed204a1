// GUMBEL_SOFTMAX training branch of one quantisation level (reference modules/quantize.py:125-130 with
// distributions/gumbel.py:8-18): p = softmax((-dist + G)/T), emb = p @ codebook, G = -log(-log(U + 1e-20) + 1e-20).
// The two matrix products (x cb^T and p cb) and their gradients run on the MFMA GEMM kernels; these are the row kernels
// around them.  Not on any shipped config's path (both gin files use ROTATION_TRICK), so they are written for clarity:
// one wave per row, K <= a few thousand, any embedding width D <= 64 (quantize.py:108-130 is width-independent); `cosine`: the
// QuantizeDistance.COSINE ranking -- S arrives as x^ c^T of the NORMALISED operands and dist = -S (quantize.py:115-119).
#include <math.h>
#include "common.h"

namespace {

// S [B,K] = x cb^T on entry -> P = softmax((-(|x|^2 + |c|^2 - 2 S) + G)/T) on exit; ids = first argmin of the distance
__global__ __launch_bounds__(256) void gumbel_rows_fwd_kernel(float *S, const float *x, const float *cc, const float *U, int64_t B,
                                                              int64_t K, float inv_t, int64_t *ids, int D, int cosine) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    float xx = (!cosine && lane < D) ? x[row * D + lane] * x[row * D + lane] : 0.0f;
    xx = hv_wave_sum(xx);
    float *s = S + row * K;
    float best = INFINITY;
    int64_t bi = 0;
    float mx = -INFINITY;
    for (int64_t k = lane; k < K; k += 64) {
        const float dist = cosine ? -s[k] : (xx + cc[k]) - 2.0f * s[k];
        if (dist < best) { best = dist; bi = k; }
        const float g = -logf(-logf(U[row * K + k] + 1e-20f) + 1e-20f);
        const float l = (-dist + g) * inv_t;
        s[k] = l;
        mx = fmaxf(mx, l);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int64_t oi = __shfl_xor(bi, o);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    mx = hv_wave_max(mx);
    float sum = 0.0f;
    for (int64_t k = lane; k < K; k += 64) sum += expf(s[k] - mx);
    sum = hv_wave_sum(sum);
    for (int64_t k = lane; k < K; k += 64) s[k] = expf(s[k] - mx) / sum;
    if (lane == 0) ids[row] = bi;
}

// loss[b] = |x - emb|^2 + beta |x - emb|^2  (loss.py:41-44; both terms have the same value)
__global__ __launch_bounds__(256) void gumbel_loss_kernel(const float *x, const float *emb, int64_t B, float beta, float *loss, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    float d = lane < D ? x[row * D + lane] - emb[row * D + lane] : 0.0f;
    d = hv_wave_sum(d * d);
    if (lane == 0) loss[row] = d + beta * d;
}

// g_emb_total = g_out + g_l * 2 (emb - x)   (the |sg(x) - emb|^2 term); g_out / g_l may be null
__global__ __launch_bounds__(256) void gumbel_gemb_kernel(const float *g_out, const float *g_l, int64_t gl_stride, const float *x,
                                                          const float *emb, int64_t B, float *g_emb, int D) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * D) return;
    const int64_t row = i / D;
    const float gl = g_l != nullptr ? g_l[row * gl_stride] : 0.0f;
    g_emb[i] = (g_out != nullptr ? g_out[i] : 0.0f) + gl * 2.0f * (emb[i] - x[i]);
}

// gP [B,K] (= g_emb cb^T) and P -> gS = 2/T * P (gP - sum_k P gP)  in place of gP;  g_xx[row] = -(1/T) sum_k g_logit
//   logit = (-dist + G)/T, dist = xx + cc - 2 S  =>  g_dist = -g_logit/T,  g_S = -2 g_dist,  g_xx = sum_k g_dist
__global__ __launch_bounds__(256) void gumbel_rows_bwd_kernel(const float *P, float *gP, int64_t B, int64_t K, float inv_t,
                                                              float *g_xx) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float *p = P + row * K;
    float *g = gP + row * K;
    float dotv = 0.0f;
    for (int64_t k = lane; k < K; k += 64) dotv += p[k] * g[k];
    dotv = hv_wave_sum(dotv);
    float sx = 0.0f;
    for (int64_t k = lane; k < K; k += 64) {
        const float g_logit = p[k] * (g[k] - dotv);
        const float g_dist = -g_logit * inv_t;
        g[k] = -2.0f * g_dist;  // g_S
        sx += g_dist;
    }
    sx = hv_wave_sum(sx);
    if (lane == 0) g_xx[row] = sx;
}

// g_x (+)= 2 x g_xx + g_l 2 beta (x - emb)      (g_x already holds g_S cb)
__global__ __launch_bounds__(256) void gumbel_gx_kernel(float *g_x, const float *x, const float *emb, const float *g_xx,
                                                        const float *g_l, int64_t gl_stride, float beta, int64_t B, int D) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * D) return;
    const int64_t row = i / D;
    const float gl = g_l != nullptr ? g_l[row * gl_stride] : 0.0f;
    g_x[i] = g_x[i] + 2.0f * x[i] * g_xx[row] + gl * 2.0f * beta * (x[i] - emb[i]);
}

// g_cb (+)= 2 cb * g_cc[k]      (g_cc = column sums of g_dist = -g_S/2 summed over rows)
__global__ __launch_bounds__(256) void gumbel_gcb_kernel(float *g_cb, const float *cb, const float *gS_colsum, int64_t K, int D) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= K * D) return;
    g_cb[i] = g_cb[i] + 2.0f * cb[i] * (-0.5f * gS_colsum[i / D]);
}

}  // namespace

extern "C" int hidvae_gumbel_rows_fwd(float *S, const float *x, const float *cc, const float *U, int64_t B, int64_t K,
                                      float temperature, int64_t *ids, int embed_dim, int cosine, void *stream) {
    HV_REQUIRE(S && U && ids && B >= 1 && K >= 1 && temperature > 0.0f && (cosine || (x && cc)), "gumbel_rows_fwd: bad arguments");
    HV_REQUIRE(embed_dim >= 1 && embed_dim <= 64, "gumbel_rows_fwd: embed_dim %d (1 .. 64)", embed_dim);
    hipLaunchKernelGGL(gumbel_rows_fwd_kernel, dim3((unsigned)hv_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, S, x, cc, U, B, K,
                       1.0f / temperature, ids, embed_dim, cosine);
    HV_LAUNCH_CHECK("gumbel_rows_fwd");
    return HIDVAE_OK;
}

extern "C" int hidvae_gumbel_loss(const float *x, const float *emb, int64_t B, float beta, float *loss, int embed_dim, void *stream) {
    HV_REQUIRE(x && emb && loss && B >= 1 && embed_dim >= 1 && embed_dim <= 64, "gumbel_loss: bad arguments");
    const int D = embed_dim;
    hipLaunchKernelGGL(gumbel_loss_kernel, dim3((unsigned)hv_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, x, emb, B, beta, loss, D);
    HV_LAUNCH_CHECK("gumbel_loss");
    return HIDVAE_OK;
}

extern "C" int hidvae_gumbel_gemb(const float *g_out, const float *g_l, int64_t gl_stride, const float *x, const float *emb, int64_t B,
                                  float *g_emb, int embed_dim, void *stream) {
    HV_REQUIRE(x && emb && g_emb && B >= 1 && embed_dim >= 1 && embed_dim <= 64, "gumbel_gemb: bad arguments");
    const int D = embed_dim;
    hipLaunchKernelGGL(gumbel_gemb_kernel, dim3((unsigned)hv_cdiv(B * D, 256)), dim3(256), 0, (hipStream_t)stream, g_out, g_l,
                       gl_stride, x, emb, B, g_emb, D);
    HV_LAUNCH_CHECK("gumbel_gemb");
    return HIDVAE_OK;
}

extern "C" int hidvae_gumbel_rows_bwd(const float *P, float *gP, int64_t B, int64_t K, float temperature, float *g_xx, void *stream) {
    HV_REQUIRE(P && gP && g_xx && B >= 1 && K >= 1 && temperature > 0.0f, "gumbel_rows_bwd: bad arguments");
    hipLaunchKernelGGL(gumbel_rows_bwd_kernel, dim3((unsigned)hv_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, P, gP, B, K,
                       1.0f / temperature, g_xx);
    HV_LAUNCH_CHECK("gumbel_rows_bwd");
    return HIDVAE_OK;
}

extern "C" int hidvae_gumbel_finish(float *g_x, const float *x, const float *emb, const float *g_xx, const float *g_l,
                                    int64_t gl_stride, float beta, int64_t B, float *g_cb, const float *cb, const float *gS_colsum,
                                    int64_t K, int embed_dim, void *stream) {
    HV_REQUIRE(g_x && x && emb && g_xx && g_cb && cb && gS_colsum && B >= 1 && K >= 1 && embed_dim >= 1 && embed_dim <= 64,
               "gumbel_finish: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int D = embed_dim;
    hipLaunchKernelGGL(gumbel_gx_kernel, dim3((unsigned)hv_cdiv(B * D, 256)), dim3(256), 0, s, g_x, x, emb, g_xx, g_l, gl_stride, beta, B, D);
    HV_LAUNCH_CHECK("gumbel_gx");
    hipLaunchKernelGGL(gumbel_gcb_kernel, dim3((unsigned)hv_cdiv(K * D, 256)), dim3(256), 0, s, g_cb, cb, gS_colsum, K, D);
    HV_LAUNCH_CHECK("gumbel_gcb");
    return HIDVAE_OK;
}
