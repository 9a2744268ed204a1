// Stand-alone forms of the reference's small loss / sampling modules -- the pieces the fused step carries inside bigger kernels,
// offered as their own entry points so the mirrored nn.Modules (modules/loss.py: ReconstructionLoss, QuantizeLoss;
// distributions/gumbel.py: sample_gumbel, gumbel_softmax_sample) are callable on their own exactly like the reference's.
// One wave per row, fixed summation order (lane-strided partial sums, then the xor butterfly): bit-reproducible.
#include <math.h>
#include "common.h"

namespace {

// s = sum_j (a[m,j] - b[m,j])^2;  out[m] = s + extra * s               (loss.py:11-12 with extra = 0; loss.py:41-44 with
// extra = commitment_weight: emb_loss + cw * query_loss, two terms of equal value)
__global__ __launch_bounds__(256) void sqdiff_rows_kernel(const float *a, int64_t lda, const float *b, int64_t ldb, int64_t M, int64_t N,
                                                          float extra, float *out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float *pa = a + row * lda, *pb = b + row * ldb;
    float s = 0.0f;
    for (int64_t j = lane; j < N; j += 64) {
        const float d = pa[j] - pb[j];
        s = fmaf(d, d, s);
    }
    s = hv_wave_sum(s);
    if (lane == 0) out[row] = extra != 0.0f ? s + extra * s : s;
}

// ga[m,j] = g[m] * sa * 2 (a - b),  gb[m,j] = g[m] * sb * 2 (b - a)     (either output may be null)
__global__ __launch_bounds__(256) void sqdiff_rows_bwd_kernel(const float *g, int64_t g_stride, const float *a, int64_t lda, const float *b,
                                                              int64_t ldb, int64_t M, int64_t N, float sa, float sb, float *ga, float *gb) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    const int64_t row = i / N, col = i - row * N;
    const float d = a[row * lda + col] - b[row * ldb + col];
    const float gr = g[row * g_stride];
    if (ga != nullptr) ga[i] = gr * sa * 2.0f * d;
    if (gb != nullptr) gb[i] = gr * sb * 2.0f * (-d);
}

// G = -log(-log(U + eps) + eps)                                          (distributions/gumbel.py:8-11)
__global__ __launch_bounds__(256) void gumbel_noise_kernel(const float *U, int64_t n, float eps, float *out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = -logf(-logf(U[i] + eps) + eps);
}

// out[m,:] = softmax((logits[m,:] + G(U[m,:])) / T)                      (distributions/gumbel.py:14-18)
__global__ __launch_bounds__(256) void gumbel_softmax_rows_kernel(const float *logits, const float *U, int64_t B, int64_t K, float inv_t,
                                                                  float *out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float *l = logits + row * K, *u = U + row * K;
    float *o = out + row * K;
    float mx = -INFINITY;
    for (int64_t k = lane; k < K; k += 64) {
        const float v = (l[k] + (-logf(-logf(u[k] + 1e-20f) + 1e-20f))) * inv_t;
        o[k] = v;
        mx = fmaxf(mx, v);
    }
    mx = hv_wave_max(mx);
    float sum = 0.0f;
    for (int64_t k = lane; k < K; k += 64) sum += expf(o[k] - mx);
    sum = hv_wave_sum(sum);
    for (int64_t k = lane; k < K; k += 64) o[k] = expf(o[k] - mx) / sum;
}

// one lane stores the device's constant-rate wall clock (100 MHz on gfx950: 10 ns ticks)
__global__ void timestamp_kernel(int64_t *slot) {
    if (threadIdx.x == 0) *slot = (int64_t)wall_clock64();
}

}  // namespace

extern "C" int hidvae_timestamp(int64_t *slot, void *stream) {
    HV_REQUIRE(slot, "timestamp: null slot");
    hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, slot);
    HV_LAUNCH_CHECK("timestamp");
    return HIDVAE_OK;
}

extern "C" int hidvae_sqdiff_rows(const float *a, int64_t lda, const float *b, int64_t ldb, int64_t M, int64_t N, float extra,
                                  float *out, void *stream) {
    HV_REQUIRE(a && b && out && M >= 1 && N >= 1 && lda >= N && ldb >= N, "sqdiff_rows: bad arguments");
    hipLaunchKernelGGL(sqdiff_rows_kernel, dim3((unsigned)hv_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, M, N, extra, out);
    HV_LAUNCH_CHECK("sqdiff_rows");
    return HIDVAE_OK;
}

extern "C" int hidvae_sqdiff_rows_bwd(const float *g, int64_t g_stride, const float *a, int64_t lda, const float *b, int64_t ldb, int64_t M,
                                      int64_t N, float scale_a, float scale_b, float *ga, float *gb, void *stream) {
    HV_REQUIRE(g && a && b && (ga || gb) && M >= 1 && N >= 1 && lda >= N && ldb >= N, "sqdiff_rows_bwd: bad arguments");
    hipLaunchKernelGGL(sqdiff_rows_bwd_kernel, dim3((unsigned)hv_cdiv(M * N, 256)), dim3(256), 0, (hipStream_t)stream, g, g_stride, a, lda, b,
                       ldb, M, N, scale_a, scale_b, ga, gb);
    HV_LAUNCH_CHECK("sqdiff_rows_bwd");
    return HIDVAE_OK;
}

extern "C" int hidvae_gumbel_noise(const float *U, int64_t n, float eps, float *out, void *stream) {
    HV_REQUIRE(U && out && n >= 1, "gumbel_noise: bad arguments");
    hipLaunchKernelGGL(gumbel_noise_kernel, dim3((unsigned)hv_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, U, n, eps, out);
    HV_LAUNCH_CHECK("gumbel_noise");
    return HIDVAE_OK;
}

extern "C" int hidvae_gumbel_softmax_rows(const float *logits, const float *U, int64_t B, int64_t K, float temperature, float *out,
                                          void *stream) {
    HV_REQUIRE(logits && U && out && B >= 1 && K >= 1 && temperature > 0.0f, "gumbel_softmax_rows: bad arguments");
    hipLaunchKernelGGL(gumbel_softmax_rows_kernel, dim3((unsigned)hv_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, logits, U, B, K,
                       1.0f / temperature, out);
    HV_LAUNCH_CHECK("gumbel_softmax_rows");
    return HIDVAE_OK;
}

// ---- HRqVae.predict_tags (reference modules/h_rqvae.py:716-722): per row, the arg max of the logits and its softmax probability --
// conf = max softmax(logits) = 1 / sum_j exp(l_j - l_max), pred = the FIRST maximal column (torch.max's CPU tie rule, as the reference
// runs it).  One wave per row; replaces torch.softmax(...).max(...) (two torch kernels and a [B, C] intermediate per level).
namespace {
__global__ __launch_bounds__(256) void softmax_argmax_rows_kernel(const float *logits, int64_t B, int64_t C, int64_t ld, int64_t *pred,
                                                                  int64_t pred_stride, float *conf, int64_t conf_stride) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float *l = logits + row * ld;
    float best = -INFINITY;
    int64_t bi = 0;
    for (int64_t c = lane; c < C; c += 64) {
        const float v = l[c];
        if (v > best) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int64_t oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    float sum = 0.0f;
    for (int64_t c = lane; c < C; c += 64) sum += expf(l[c] - best);
    sum = hv_wave_sum(sum);
    if (lane == 0) {
        pred[row * pred_stride] = bi;
        conf[row * conf_stride] = 1.0f / sum;
    }
}
}  // namespace

extern "C" int hidvae_softmax_argmax_rows(const float *logits, int64_t B, int64_t C, int64_t ld_logits, int64_t *pred, int64_t pred_stride,
                                          float *conf, int64_t conf_stride, void *stream) {
    HV_REQUIRE(logits && pred && conf && B >= 1 && C >= 1 && ld_logits >= C && pred_stride >= 1 && conf_stride >= 1,
               "softmax_argmax_rows: bad arguments");
    hipLaunchKernelGGL(softmax_argmax_rows_kernel, dim3((unsigned)hv_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, logits, B, C, ld_logits,
                       pred, pred_stride, conf, conf_stride);
    HV_LAUNCH_CHECK("softmax_argmax_rows");
    return HIDVAE_OK;
}
