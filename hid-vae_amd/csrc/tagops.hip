// Row / elementwise kernels of the tag heads (reference modules/h_rqvae.py:108-227 TagPredictor, :322-331 tag projector)
// and their losses (modules/loss.py:48-85 InfoNCE, :89-265 focal / CE with mixup) for gfx950.
// One wave per row wherever a row reduction is needed (rows are <= 768 wide); column reductions (BatchNorm, the affine
// gradients of LayerNorm) use one workgroup per 64 columns walking the rows in a fixed order, so every result is
// bit-reproducible run to run.  Everything is fp32.
#include <math.h>
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// out = g * act'(ref) * mask * scale          (backward through activation + dropout of a Linear)
//   RELU: ref = forward OUTPUT (after dropout): y > 0 <=> pre > 0 and kept, so the mask is implied
//   GELU / SILU: ref = pre-activation;  SIGMOID: ref = forward output
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_bwd_kernel(const float *g, const float *ref, int64_t n, int act, const float *mask,
                                                      float scale, float *out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v = g[i];
        const float r = ref != nullptr ? ref[i] : 0.0f;
        const float ms = mask != nullptr ? mask[i] * scale : 1.0f;
        switch (act) {
            case HIDVAE_EPI_RELU: v = r > 0.0f ? v * (mask != nullptr ? scale : 1.0f) : 0.0f; break;
            case HIDVAE_EPI_GELU: v = v * hv_dgelu(r) * ms; break;
            case HIDVAE_EPI_SIGMOID: v = v * (r * (1.0f - r)) * ms; break;
            case HIDVAE_EPI_SILU: v = v * hv_dsilu(r) * ms; break;
            default: v = v * ms; break;
        }
        out[i] = v;
    }
}

// out = a * b  (op 0)  or  a + b  (op 1), elementwise, row strides allowed
__global__ __launch_bounds__(256) void mul_kernel(const float *a, int64_t lda, const float *b, int64_t ldb, int64_t M, int64_t N,
                                                  float *out, int64_t ldo, int op) {
    const int64_t n = M * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / N, c = i - r * N;
        const float av = a[r * lda + c], bv = b[r * ldb + c];
        out[r * ldo + c] = op == 0 ? av * bv : (op == 1 ? av + bv : av - bv);
    }
}

// dst[:, c] = sum over sources s with c < width[s] of src_s[:, c]   (gradient of the concatenated-embedding views)
constexpr int MAX_SLICES = 24;
struct SliceArgs {
    const float *src[MAX_SLICES];
    int width[MAX_SLICES];
    int n;
    int64_t M, N;
    float *dst;
};
__global__ __launch_bounds__(256) void sum_prefix_slices_kernel(SliceArgs a) {
    const int64_t tot = a.M * a.N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / a.N;
        const int c = (int)(i - r * a.N);
        float v = 0.0f;
        for (int s = 0; s < a.n; s++)
            if (c < a.width[s]) v += a.src[s][r * a.width[s] + c];
        a.dst[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm (biased variance) + optional ReLU + dropout mask + residual:  y = drop(relu(LN(x))) + res
// ------------------------------------------------------------------------------------------------
template <bool RESIDENT>  // RESIDENT: N <= 1024, the row is read once and kept in registers between the three passes
__device__ __forceinline__ void layernorm_fwd_body(int64_t blk, const float *x, int64_t M, int64_t N, const float *gamma,
                                                   const float *beta, float eps, float *y, float *mean, float *rstd,
                                                   int relu, const float *mask, float scale, const float *res, HvDrop drop) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blk * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float *xr = x + row * N;
    float xv[RESIDENT ? 16 : 1];
    float s = 0.0f;
    if (RESIDENT) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int64_t i = lane + 64 * j;
            xv[j] = i < N ? xr[i] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 16; j++) s += xv[j];  // (columns past N hold 0: same sum, same order as the strided loop)
    } else {
        for (int64_t i = lane; i < N; i += 64) s += xr[i];
    }
    const float mu = hv_wave_sum(s) / (float)N;
    float v = 0.0f;
    if (RESIDENT) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float d = xv[j] - mu;
            if (lane + 64 * j < N) v += d * d;
        }
    } else {
        for (int64_t i = lane; i < N; i += 64) { const float d = xr[i] - mu; v += d * d; }
    }
    const float rs = 1.0f / sqrtf(hv_wave_sum(v) / (float)N + eps);
    if (RESIDENT) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int64_t i = lane + 64 * j;
            if (i < N) {
                float o = (xv[j] - mu) * rs * gamma[i] + beta[i];
                if (relu) o = fmaxf(o, 0.0f);
                if (mask != nullptr) o = o * (mask[row * N + i] * scale);
                else if (drop.state != nullptr) o = hv_drop_keep(drop, (unsigned long long)(row * N + i)) ? o * scale : 0.0f;
                if (res != nullptr) o = o + res[row * N + i];
                y[row * N + i] = o;
            }
        }
    } else {
        for (int64_t i = lane; i < N; i += 64) {
            float o = (xr[i] - mu) * rs * gamma[i] + beta[i];
            if (relu) o = fmaxf(o, 0.0f);
            if (mask != nullptr) o = o * (mask[row * N + i] * scale);
            else if (drop.state != nullptr) o = hv_drop_keep(drop, (unsigned long long)(row * N + i)) ? o * scale : 0.0f;
            if (res != nullptr) o = o + res[row * N + i];
            y[row * N + i] = o;
        }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

template <bool RESIDENT>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float *x, int64_t M, int64_t N, const float *gamma,
                                                            const float *beta, float eps, float *y, float *mean, float *rstd,
                                                            int relu, const float *mask, float scale, const float *res, HvDrop drop) {
    layernorm_fwd_body<RESIDENT>((int64_t)blockIdx.x, x, M, N, gamma, beta, eps, y, mean, rstd, relu, mask, scale, res, drop);
}

// The same forward with a row shared by S waves (S = 2 or 4; a workgroup is 4 rows x S waves): at the tag heads' sizes (1024-2048 rows)
// one wave per row leaves one wave per SIMD and nothing to hide its three dependent phases behind.  Each wave keeps its slice of the row
// (columns q*64*NV + lane + 64 j) in registers; the row sums are added over the S slices in ascending order through LDS.
template <int NV, int S>
__global__ __launch_bounds__(256 * S) void layernorm_fwd_split_kernel(const float *x, int64_t M, int64_t N, const float *gamma, const float *beta,
                                                                      float eps, float *y, float *mean, float *rstd, int relu,
                                                                      const float *mask, float scale, const float *res, HvDrop drop) {
    __shared__ float part[2][4][S];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = wave / S, q = wave % S;
    const int64_t row = (int64_t)blockIdx.x * 4 + r;
    const bool live = row < M;
    const int64_t c0 = (int64_t)q * 64 * NV + lane;
    float xv[NV];
    float sm = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int64_t i = c0 + 64 * j;
        xv[j] = (live && i < N) ? x[row * N + i] : 0.0f;
        sm += xv[j];
    }
    sm = hv_wave_sum(sm);
    if (lane == 0) part[0][r][q] = sm;
    __syncthreads();
    float tot = part[0][r][0];
#pragma unroll
    for (int k = 1; k < S; k++) tot += part[0][r][k];
    const float mu = tot / (float)N;
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const float d = xv[j] - mu;
        if (c0 + 64 * j < N) v += d * d;
    }
    v = hv_wave_sum(v);
    if (lane == 0) part[1][r][q] = v;
    __syncthreads();
    float vt = part[1][r][0];
#pragma unroll
    for (int k = 1; k < S; k++) vt += part[1][r][k];
    const float rs = 1.0f / sqrtf(vt / (float)N + eps);
    if (!live) return;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int64_t i = c0 + 64 * j;
        if (i < N) {
            float o = (xv[j] - mu) * rs * gamma[i] + beta[i];
            if (relu) o = fmaxf(o, 0.0f);
            if (mask != nullptr) o = o * (mask[row * N + i] * scale);
            else if (drop.state != nullptr) o = hv_drop_keep(drop, (unsigned long long)(row * N + i)) ? o * scale : 0.0f;
            if (res != nullptr) o = o + res[row * N + i];
            y[row * N + i] = o;
        }
    }
    if (q == 0 && lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// gh = gy * mask*scale * (h > 0 if relu), h = xhat*gamma+beta;  dy = gh*gamma
// gx = rstd * (dy - mean(dy) - xhat * mean(dy*xhat))
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float *gy, const float *x, const float *gamma, const float *beta,
                                                            const float *mean, const float *rstd, int64_t M, int64_t N, int relu,
                                                            const float *mask, float scale, float *gx) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.0f, s2 = 0.0f;
    for (int64_t i = lane; i < N; i += 64) {
        const float xh = (x[row * N + i] - mu) * rs;
        float g = gy[row * N + i];
        if (mask != nullptr) g = g * (mask[row * N + i] * scale);
        if (relu && !(xh * gamma[i] + beta[i] > 0.0f)) g = 0.0f;
        const float dy = g * gamma[i];
        s1 += dy;
        s2 += dy * xh;
    }
    s1 = hv_wave_sum(s1) / (float)N;
    s2 = hv_wave_sum(s2) / (float)N;
    for (int64_t i = lane; i < N; i += 64) {
        const float xh = (x[row * N + i] - mu) * rs;
        float g = gy[row * N + i];
        if (mask != nullptr) g = g * (mask[row * N + i] * scale);
        if (relu && !(xh * gamma[i] + beta[i] > 0.0f)) g = 0.0f;
        gx[row * N + i] = rs * ((g * gamma[i] - s1) - xh * s2);
    }
}

// affine gradients of LayerNorm, two fixed-order passes: (1) one workgroup per (64 columns x 128 rows) writes partial
// sums, (2) the row chunks are added in ascending order.  Bit-reproducible, and parallel over rows as well as columns.
constexpr int LN_CHUNK = 128;
__global__ __launch_bounds__(256) void layernorm_param_partial_kernel(const float *gy, const float *x, const float *gamma,
                                                                      const float *beta, const float *mean, const float *rstd,
                                                                      int64_t M, int64_t N, int relu, const float *mask, float scale,
                                                                      float *part) {
    __shared__ float red[2][4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 64 + c;
    const int64_t m0 = (int64_t)blockIdx.y * LN_CHUNK;
    float sg = 0.0f, sb = 0.0f;
    if (n < N) {
        const float ga = gamma[n], be = beta[n];
#pragma unroll 4
        for (int j = 0; j < LN_CHUNK / 4; j++) {
            const int64_t m = m0 + rl + 4 * j;
            if (m < M) {
                const float xh = (x[m * N + n] - mean[m]) * rstd[m];
                float g = gy[m * N + n];
                if (mask != nullptr) g = g * (mask[m * N + n] * scale);
                if (relu && !(xh * ga + be > 0.0f)) g = 0.0f;
                sg += g * xh;
                sb += g;
            }
        }
    }
    red[0][rl][c] = sg;
    red[1][rl][c] = sb;
    __syncthreads();
    if (rl == 0 && n < N) {
        part[((int64_t)blockIdx.y * 2 + 0) * N + n] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        part[((int64_t)blockIdx.y * 2 + 1) * N + n] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    }
}

// Input AND affine gradients in one pass over (gy, x): a workgroup owns LNF_ROWS rows (one per wave),
// keeps the row in registers between the two row reductions and the gx write, and adds g*xhat / g into per-lane column sums;
// the four waves' sums are combined in LDS and written as one partial per workgroup (summed by layernorm_param_final_kernel in
// ascending order).  N <= 64*LNF_NV.
constexpr int LNF_ROWS = 4, LNF_NV = 16;  // one row per wave: rows in flight, not in sequence, hide the two dependent phases
// yout (optional, replaces mask): the forward OUTPUT y = relu(h) * keep * scale -- y > 0 exactly where the unit was active and kept, so
//   the gate of ReLU -> Dropout needs neither the keep-mask nor h.
// in_relu_scale (0 = off): this LayerNorm's INPUT x was itself relu(.) * keep * in_relu_scale (Linear -> ReLU -> Dropout -> LayerNorm,
//   h_rqvae.py:157-162): gx leaves the launch already taken through that ReLU / Dropout, gx * (x > 0 ? in_relu_scale : 0).
// NV: columns per lane (N <= 64 NV); instantiated for 4, 8, 12, 16 so that the common widths keep a small register footprint -- the one
//   NV = 16 form took 256 VGPRs + 42 AGPRs, ONE wave per SIMD: it could start only on a SIMD with nothing else on it, which beside the
//   other streams' GEMM launches meant waiting for them (25 us in-step for a 4.5 us kernel)
template <int NV>
__device__ __forceinline__ void layernorm_bwd_fused_body(int64_t blk, float *red, const float *gy, const float *x, const float *gamma,
                                                         const float *beta, const float *mean, const float *rstd, int64_t M, int64_t N,
                                                         int relu, const float *mask, float scale, float *gx, float *part,
                                                         const float *yout, float in_relu_scale, const float *gy2, float *gsum) {
    // gy2 (optional): a second gradient of the same output, added on the way in (the residual path of TagPredictor's blocks,
    //   h_rqvae.py:165-186: f_{n+1} = LN(..) + f_n hands f_n's producer two gradients); gsum (optional): that sum, written out for the
    //   next residual hop
    // red: [2][3][64 NV] floats of LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = blk * LNF_ROWS + wave;  // one row per wave
    float g[NV], xh[NV], ga[NV];
    unsigned xpos = 0u;  // bit j: the LayerNorm input of column lane + 64 j is > 0 (in_relu_scale)
    float s1 = 0.0f, s2 = 0.0f, rs = 0.0f;
    const bool live = row < M;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int64_t c = lane + 64 * j;
        g[j] = 0.0f;
        xh[j] = 0.0f;
        ga[j] = c < N ? gamma[c] : 0.0f;
    }
    if (live) {
        const float mu = mean[row];
        rs = rstd[row];
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int64_t c = lane + 64 * j;
            if (c < N) {
                const float xraw = x[row * N + c];
                if (xraw > 0.0f) xpos |= 1u << j;
                xh[j] = (xraw - mu) * rs;
                float gv = gy[row * N + c];
                if (gy2 != nullptr) gv += gy2[row * N + c];
                if (gsum != nullptr) gsum[row * N + c] = gv;
                if (yout != nullptr) {
                    if (relu) gv = yout[row * N + c] > 0.0f ? gv * scale : 0.0f;
                } else {
                    if (mask != nullptr) gv = gv * (mask[row * N + c] * scale);
                    if (relu && !(xh[j] * ga[j] + beta[c] > 0.0f)) gv = 0.0f;
                }
                g[j] = gv;
                const float dy = gv * ga[j];
                s1 += dy;
                s2 += dy * xh[j];
            }
        }
    }
    s1 = hv_wave_sum(s1) / (float)N;
    s2 = hv_wave_sum(s2) / (float)N;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int64_t c = lane + 64 * j;
        if (live && c < N && gx != nullptr) {
            float o = rs * ((g[j] * ga[j] - s1) - xh[j] * s2);
            if (in_relu_scale != 0.0f) o = ((xpos >> j) & 1u) ? o * in_relu_scale : 0.0f;
            gx[row * N + c] = o;
        }
        // the row's terms of the affine gradients: waves 1..3 hand theirs to wave 0, which adds the four in a fixed order
        const float tg = 0.0f + g[j] * xh[j], tb = 0.0f + g[j];
        if (wave > 0) {
            red[(0 * 3 + wave - 1) * 64 * NV + lane + 64 * j] = tg;
            red[(1 * 3 + wave - 1) * 64 * NV + lane + 64 * j] = tb;
        } else {
            g[j] = tg;
            xh[j] = tb;
        }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int64_t c = lane + 64 * j;
            if (c < N) {
                part[(blk * 2 + 0) * N + c] = (g[j] + red[(0 * 3 + 0) * 64 * NV + c]) + (red[(0 * 3 + 1) * 64 * NV + c] + red[(0 * 3 + 2) * 64 * NV + c]);
                part[(blk * 2 + 1) * N + c] = (xh[j] + red[(1 * 3 + 0) * 64 * NV + c]) + (red[(1 * 3 + 1) * 64 * NV + c] + red[(1 * 3 + 2) * 64 * NV + c]);
            }
        }
    }
}

template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_fused_kernel(const float *gy, const float *x, const float *gamma, const float *beta,
                                                                  const float *mean, const float *rstd, int64_t M, int64_t N, int relu,
                                                                  const float *mask, float scale, float *gx, float *part,
                                                                  const float *yout, float in_relu_scale, const float *gy2, float *gsum) {
    __shared__ float red[2 * 3 * 64 * NV];
    layernorm_bwd_fused_body<NV>((int64_t)blockIdx.x, red, gy, x, gamma, beta, mean, rstd, M, N, relu, mask, scale, gx, part, yout, in_relu_scale,
                                 gy2, gsum);
}

// the fused backward with a row shared by S waves (see layernorm_fwd_split_kernel): a workgroup is 4 rows x S waves; the two row sums are
// added over the S slices in ascending order through LDS, the column partials over the 4 rows exactly as in the one-wave-per-row form
template <int NV, int S>
__global__ __launch_bounds__(256 * S) void layernorm_bwd_split_kernel(const float *gy, const float *x, const float *gamma, const float *beta,
                                                                      const float *mean, const float *rstd, int64_t M, int64_t N, int relu,
                                                                      const float *mask, float scale, float *gx, float *part,
                                                                      const float *yout, float in_relu_scale, const float *gy2, float *gsum) {
    constexpr int W = 64 * NV * S;  // columns the workgroup covers
    __shared__ float red[2 * 3 * W];
    __shared__ float rsum[2][4][S];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = wave / S, q = wave % S;
    const int64_t blk = blockIdx.x, row = blk * LNF_ROWS + r;
    const bool live = row < M;
    const int c0 = q * 64 * NV + lane;
    float g[NV], xh[NV], ga[NV];
    unsigned xpos = 0u;
    float s1 = 0.0f, s2 = 0.0f, rs = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int c = c0 + 64 * j;
        g[j] = 0.0f;
        xh[j] = 0.0f;
        ga[j] = c < N ? gamma[c] : 0.0f;
    }
    if (live) {
        const float mu = mean[row];
        rs = rstd[row];
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int c = c0 + 64 * j;
            if (c < N) {
                const float xraw = x[row * N + c];
                if (xraw > 0.0f) xpos |= 1u << j;
                xh[j] = (xraw - mu) * rs;
                float gv = gy[row * N + c];
                if (gy2 != nullptr) gv += gy2[row * N + c];
                if (gsum != nullptr) gsum[row * N + c] = gv;
                if (yout != nullptr) {
                    if (relu) gv = yout[row * N + c] > 0.0f ? gv * scale : 0.0f;
                } else {
                    if (mask != nullptr) gv = gv * (mask[row * N + c] * scale);
                    if (relu && !(xh[j] * ga[j] + beta[c] > 0.0f)) gv = 0.0f;
                }
                g[j] = gv;
                const float dy = gv * ga[j];
                s1 += dy;
                s2 += dy * xh[j];
            }
        }
    }
    s1 = hv_wave_sum(s1);
    s2 = hv_wave_sum(s2);
    if (lane == 0) { rsum[0][r][q] = s1; rsum[1][r][q] = s2; }
    __syncthreads();
    s1 = rsum[0][r][0];
    s2 = rsum[1][r][0];
#pragma unroll
    for (int k = 1; k < S; k++) { s1 += rsum[0][r][k]; s2 += rsum[1][r][k]; }
    s1 = s1 / (float)N;
    s2 = s2 / (float)N;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int c = c0 + 64 * j;
        if (live && c < N && gx != nullptr) {
            float o = rs * ((g[j] * ga[j] - s1) - xh[j] * s2);
            if (in_relu_scale != 0.0f) o = ((xpos >> j) & 1u) ? o * in_relu_scale : 0.0f;
            gx[row * N + c] = o;
        }
        const float tg = 0.0f + g[j] * xh[j], tb = 0.0f + g[j];
        if (r > 0) {
            red[(0 * 3 + r - 1) * W + c] = tg;
            red[(1 * 3 + r - 1) * W + c] = tb;
        } else {
            g[j] = tg;
            xh[j] = tb;
        }
    }
    __syncthreads();
    if (r == 0) {
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int c = c0 + 64 * j;
            if (c < N) {
                part[(blk * 2 + 0) * N + c] = (g[j] + red[(0 * 3 + 0) * W + c]) + (red[(0 * 3 + 1) * W + c] + red[(0 * 3 + 2) * W + c]);
                part[(blk * 2 + 1) * N + c] = (xh[j] + red[(1 * 3 + 0) * W + c]) + (red[(1 * 3 + 1) * W + c] + red[(1 * 3 + 2) * W + c]);
            }
        }
    }
}

static void launch_layernorm_bwd_fused(int64_t chunks, hipStream_t s, const float *gy, const float *x, const float *gamma, const float *beta,
                                       const float *mean, const float *rstd, int64_t M, int64_t N, int relu, const float *mask, float scale,
                                       float *gx, float *part, const float *yout, float in_relu_scale, const float *gy2, float *gsum) {
    const dim3 grid((unsigned)chunks), block(256);
#define HV_LNB(NVV) hipLaunchKernelGGL(layernorm_bwd_fused_kernel<NVV>, grid, block, 0, s, gy, x, gamma, beta, mean, rstd, M, N, relu, mask, scale, gx, \
                                       part, yout, in_relu_scale, gy2, gsum)
#define HV_LNS(NVV, SS) hipLaunchKernelGGL((layernorm_bwd_split_kernel<NVV, SS>), grid, dim3(256 * SS), 0, s, gy, x, gamma, beta, mean, rstd, M, N, relu, \
                                           mask, scale, gx, part, yout, in_relu_scale, gy2, gsum)
    const bool split = M <= 16384;  // few rows: share a row between waves (more waves in flight); many rows: one wave per row fills the chip
    if (N <= 256) HV_LNB(4);
    else if (N <= 512) { if (split) HV_LNS(4, 2); else HV_LNB(8); }
    else if (N <= 768) { if (split) HV_LNS(3, 4); else HV_LNB(12); }
    else { if (split) HV_LNS(4, 4); else HV_LNB(16); }
#undef HV_LNS
#undef HV_LNB
}
// 32 columns x 32 chunk groups per workgroup: group q adds chunks q, q+32, ... in ascending order, then the groups are added
// in ascending order (the fused backward leaves one partial per 4 rows, so depth matters more than work here)
__device__ __forceinline__ void layernorm_param_final_body(int64_t blk, float (*red)[32][33], const float *part, int64_t chunks, int64_t N,
                                                           float *ggamma, float *gbeta, int accumulate) {
    const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int64_t n = blk * 32 + c;
    float a = 0.0f, b = 0.0f;
    if (n < N)
        for (int64_t k0 = q; k0 < chunks; k0 += 8 * 32) {  // eight chunks' loads in flight, added in the same ascending order
            float va[8], vb[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int64_t k = k0 + 32 * j;
                va[j] = k < chunks ? part[(k * 2 + 0) * N + n] : 0.0f;
                vb[j] = k < chunks ? part[(k * 2 + 1) * N + n] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) { a += va[j]; b += vb[j]; }
        }
    red[0][q][c] = a;
    red[1][q][c] = b;
    __syncthreads();
    if (q < 2 && n < N) {  // wave 0: lanes 0-31 finish gamma, lanes 32-63 finish beta
        float v = red[q][0][c];
        for (int j = 1; j < 32; j++) v += red[q][j][c];
        float *dst = q == 0 ? ggamma : gbeta;
        dst[n] = accumulate ? dst[n] + v : v;
    }
}

__global__ __launch_bounds__(1024) void layernorm_param_final_kernel(const float *part, int64_t chunks, int64_t N, float *ggamma,
                                                                     float *gbeta, int accumulate) {
    __shared__ float red[2][32][33];
    layernorm_param_final_body((int64_t)blockIdx.x, red, part, chunks, N, ggamma, gbeta, accumulate);
}
// the affine-gradient finish of MANY LayerNorms in one launch (all the tag heads' LayerNorms of a step: their partials are complete
// long before anyone needs gamma / beta gradients, so the 21 finishing launches of a step -- each a 5 us node on its level's chain --
// become one at the end of the backward)
constexpr int LN_FINAL_MAX = 32;
struct LnFinalArgs {
    int n;
    int nb[LN_FINAL_MAX];
    hidvae_ln_final p[LN_FINAL_MAX];
};
__global__ __launch_bounds__(1024) void layernorm_param_final_many_kernel(LnFinalArgs a) {
    __shared__ float red[2][32][33];
    int bid = blockIdx.x, p = 0;
    while (p + 1 < a.n && bid >= a.nb[p]) {
        bid -= a.nb[p];
        p++;
    }
    const hidvae_ln_final &q = a.p[p];
    layernorm_param_final_body((int64_t)bid, red, q.partials, (q.M + LNF_ROWS - 1) / LNF_ROWS, q.N, q.ggamma, q.gbeta, q.accumulate);
}
// ------------------------------------------------------------------------------------------------
// BatchNorm1d (h_rqvae.py:325).  One workgroup (1024 threads = 32 columns x 32 row-lanes) per 32 columns.
// training: batch mean / biased variance (two passes), running stats <- (1-mom)*old + mom*{mean, unbiased var};
// eval: the running stats.  y = drop(relu(xhat*gamma+beta)).
// ------------------------------------------------------------------------------------------------
constexpr int BN_COLS = 32, BN_LANES = 32;
__device__ __forceinline__ float colreduce16(float v, float (*red)[BN_COLS], int c, int rl) {
    red[rl][c] = v;
    __syncthreads();
    float s = 0.0f;
    if (rl == 0) {
#pragma unroll
        for (int j = 0; j < BN_LANES; j++) s += red[j][c];
        red[0][c] = s;
    }
    __syncthreads();
    s = red[0][c];
    __syncthreads();
    return s;
}

__global__ __launch_bounds__(1024) void batchnorm_fwd_kernel(const float *x, int64_t ldx, int64_t M, int64_t N, const float *gamma,
                                                             const float *beta, float eps, float momentum, int training,
                                                             float *running_mean, float *running_var, int64_t *num_batches_tracked,
                                                             float *y, float *save_mean, float *save_rstd, int relu,
                                                             const float *mask, float scale, HvDrop drop) {
    __shared__ float red[BN_LANES][BN_COLS];
    if (training && num_batches_tracked != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    const int c = threadIdx.x & (BN_COLS - 1), rl = threadIdx.x / BN_COLS;
    const int64_t n = (int64_t)blockIdx.x * BN_COLS + c;
    const bool ok = n < N;
    float mu, rs;
    if (training) {
        float s = 0.0f;
        if (ok) for (int64_t m = rl; m < M; m += BN_LANES) s += x[m * ldx + n];
        mu = colreduce16(s, red, c, rl) / (float)M;
        float v = 0.0f;
        if (ok) for (int64_t m = rl; m < M; m += BN_LANES) { const float d = x[m * ldx + n] - mu; v += d * d; }
        const float var = colreduce16(v, red, c, rl) / (float)M;
        rs = 1.0f / sqrtf(var + eps);
        if (ok && rl == 0) {
            save_mean[n] = mu;
            save_rstd[n] = rs;
            if (running_mean != nullptr) {
                const float unbiased = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
                running_mean[n] = (1.0f - momentum) * running_mean[n] + momentum * mu;
                running_var[n] = (1.0f - momentum) * running_var[n] + momentum * unbiased;
            }
        }
    } else {
        mu = ok ? running_mean[n] : 0.0f;
        rs = ok ? 1.0f / sqrtf(running_var[n] + eps) : 0.0f;
    }
    if (!ok) return;
    const float ga = gamma != nullptr ? gamma[n] : 1.0f, be = beta != nullptr ? beta[n] : 0.0f;
    for (int64_t m = rl; m < M; m += BN_LANES) {
        float o = (x[m * ldx + n] - mu) * rs * ga + be;
        if (relu) o = fmaxf(o, 0.0f);
        if (mask != nullptr) o = o * (mask[m * N + n] * scale);
        else if (drop.state != nullptr) o = hv_drop_keep(drop, (unsigned long long)(m * N + n)) ? o * scale : 0.0f;
        y[m * N + n] = o;
    }
}

// the gradient entering the normalisation, taken back through Dropout and ReLU: either from the keep-mask and h = xhat*gamma+beta, or
// (yout != NULL) read off the forward OUTPUT y = relu(h) * keep * scale, which is > 0 exactly where the unit was active and kept
__device__ __forceinline__ float bn_gate(float g, float xh, float ga, float be, int relu, const float *mask, const float *yout, int64_t off,
                                         float scale) {
    if (yout != nullptr) return relu ? (yout[off] > 0.0f ? g * scale : 0.0f) : g;
    if (mask != nullptr) g = g * (mask[off] * scale);
    if (relu && !(xh * ga + be > 0.0f)) g = 0.0f;
    return g;
}

// training-mode backward: gh = gy*mask*scale*(h>0); ggamma = sum gh*xhat; gbeta = sum gh;
// gx = gamma*rstd/M * (M*gh - gbeta - xhat*ggamma)
__global__ __launch_bounds__(1024) void batchnorm_bwd_kernel(const float *gy, const float *x, int64_t ldx, const float *gamma,
                                                             const float *beta, const float *save_mean, const float *save_rstd,
                                                             int64_t M, int64_t N, int relu, const float *mask, float scale,
                                                             float *gx, float *ggamma, float *gbeta, int accumulate, const float *yout) {
    __shared__ float red[BN_LANES][BN_COLS];
    const int c = threadIdx.x & (BN_COLS - 1), rl = threadIdx.x / BN_COLS;
    const int64_t n = (int64_t)blockIdx.x * BN_COLS + c;
    const bool ok = n < N;
    const float mu = ok ? save_mean[n] : 0.0f, rs = ok ? save_rstd[n] : 0.0f;
    const float ga = ok ? gamma[n] : 0.0f, be = ok ? beta[n] : 0.0f;
    float sg = 0.0f, sb = 0.0f;
    if (ok)
        for (int64_t m = rl; m < M; m += BN_LANES) {
            const float xh = (x[m * ldx + n] - mu) * rs;
            const float g = bn_gate(gy[m * N + n], xh, ga, be, relu, mask, yout, m * N + n, scale);
            sg += g * xh;
            sb += g;
        }
    sg = colreduce16(sg, red, c, rl);
    sb = colreduce16(sb, red, c, rl);
    if (!ok) return;
    if (rl == 0) {
        ggamma[n] = accumulate ? ggamma[n] + sg : sg;
        gbeta[n] = accumulate ? gbeta[n] + sb : sb;
    }
    if (gx != nullptr) {
        const float k = ga * rs / (float)M;
        for (int64_t m = rl; m < M; m += BN_LANES) {
            const float xh = (x[m * ldx + n] - mu) * rs;
            const float g = bn_gate(gy[m * N + n], xh, ga, be, relu, mask, yout, m * N + n, scale);
            gx[m * N + n] = k * (((float)M * g - sb) - xh * sg);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm in two launches each way, parallel over rows as well as columns (the one-workgroup-per-32-columns kernels above
// keep a 1024 x 512 activation on 16 CUs: 37 us forward, 23 us backward).
//   stats : one workgroup per (64 columns x 64 rows) -> per-chunk (mean, M2) by the two-pass formula on the register-resident
//           chunk, the four row lanes merged pairwise with Chan's update
//   apply : every workgroup merges the chunks of its 64 columns in ascending order (same update), then normalises 16 rows
// Chan: n = na + nb; d = mb - ma; mean = ma + d*nb/n; M2 = M2a + M2b + d*d*na*nb/n
// ------------------------------------------------------------------------------------------------
constexpr int BN2_ROWS = 64;
__device__ __forceinline__ void chan_merge(float &na, float &ma, float &qa, float nb, float mb, float qb) {
    if (nb == 0.0f) return;
    if (na == 0.0f) { na = nb; ma = mb; qa = qb; return; }
    const float n = na + nb, d = mb - ma;
    ma = ma + d * (nb / n);
    qa = (qa + qb) + d * d * (na * nb / n);
    na = n;
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const float *x, int64_t ldx, int64_t M, int64_t N, float *part) {
    __shared__ float red[3][4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 64 + c;
    const int64_t m0 = (int64_t)blockIdx.y * BN2_ROWS;
    float v[BN2_ROWS / 4];
    float cnt = 0.0f, sum = 0.0f;
#pragma unroll
    for (int j = 0; j < BN2_ROWS / 4; j++) {
        const int64_t m = m0 + rl + 4 * j;
        const bool ok = n < N && m < M;
        v[j] = ok ? x[m * ldx + n] : 0.0f;
        if (ok) { cnt += 1.0f; sum += v[j]; }
    }
    const float mu = cnt > 0.0f ? sum / cnt : 0.0f;
    float m2 = 0.0f;
#pragma unroll
    for (int j = 0; j < BN2_ROWS / 4; j++) {
        const int64_t m = m0 + rl + 4 * j;
        if (n < N && m < M) { const float d = v[j] - mu; m2 += d * d; }
    }
    red[0][rl][c] = cnt; red[1][rl][c] = mu; red[2][rl][c] = m2;
    __syncthreads();
    if (rl == 0 && n < N) {
        float na = red[0][0][c], ma = red[1][0][c], qa = red[2][0][c];
        float nb = red[0][1][c], mb = red[1][1][c], qb = red[2][1][c];
        chan_merge(na, ma, qa, nb, mb, qb);
        float nc = red[0][2][c], mc = red[1][2][c], qc = red[2][2][c];
        chan_merge(nc, mc, qc, red[0][3][c], red[1][3][c], red[2][3][c]);
        chan_merge(na, ma, qa, nc, mc, qc);
        part[((int64_t)blockIdx.y * 3 + 0) * N + n] = na;
        part[((int64_t)blockIdx.y * 3 + 1) * N + n] = ma;
        part[((int64_t)blockIdx.y * 3 + 2) * N + n] = qa;
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float *x, int64_t ldx, int64_t M, int64_t N, const float *gamma,
                                                       const float *beta, float eps, float momentum, int training, float *running_mean,
                                                       float *running_var, int64_t *num_batches_tracked, const float *part,
                                                       int64_t chunks, float *y, float *save_mean, float *save_rstd, int relu,
                                                       const float *mask, float scale, HvDrop drop) {
    __shared__ float s_mu[64], s_rs[64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 64 + c;
    if (training && num_batches_tracked != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    if (rl == 0 && n < N) {
        float mu, rs;
        if (training) {
            float na = 0.0f, ma = 0.0f, qa = 0.0f;
            for (int64_t k0 = 0; k0 < chunks; k0 += 8) {  // eight chunks' loads in flight, merged in the same ascending order
                float pn[8], pm[8], pq[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int64_t k = k0 + j < chunks ? k0 + j : chunks - 1;
                    pn[j] = part[(k * 3 + 0) * N + n]; pm[j] = part[(k * 3 + 1) * N + n]; pq[j] = part[(k * 3 + 2) * N + n];
                }
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if (k0 + j < chunks) chan_merge(na, ma, qa, pn[j], pm[j], pq[j]);
            }
            const float var = qa / (float)M;
            mu = ma;
            rs = 1.0f / sqrtf(var + eps);
            if (blockIdx.y == 0) {
                save_mean[n] = mu;
                save_rstd[n] = rs;
                if (running_mean != nullptr) {
                    const float unbiased = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
                    running_mean[n] = (1.0f - momentum) * running_mean[n] + momentum * mu;
                    running_var[n] = (1.0f - momentum) * running_var[n] + momentum * unbiased;
                }
            }
        } else {
            mu = running_mean[n];
            rs = 1.0f / sqrtf(running_var[n] + eps);
        }
        s_mu[c] = mu;
        s_rs[c] = rs;
    }
    __syncthreads();
    if (n >= N) return;
    const float mu = s_mu[c], rs = s_rs[c];
    const float ga = gamma != nullptr ? gamma[n] : 1.0f, be = beta != nullptr ? beta[n] : 0.0f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int64_t m = (int64_t)blockIdx.y * 16 + rl + 4 * j;
        if (m >= M) continue;
        float o = (x[m * ldx + n] - mu) * rs * ga + be;
        if (relu) o = fmaxf(o, 0.0f);
        if (mask != nullptr) o = o * (mask[m * N + n] * scale);
        else if (drop.state != nullptr) o = hv_drop_keep(drop, (unsigned long long)(m * N + n)) ? o * scale : 0.0f;
        y[m * N + n] = o;
    }
}

// backward: partial (sum gh*xhat, sum gh) per (64 columns x 64 rows), then every workgroup adds the chunks of its columns in
// ascending order and writes gx for 16 rows
constexpr int BN2_BWD_ROWS = 32;  // rows per partial chunk of the backward: every load of a thread's 8 rows is in flight at once (64-row
                                  // chunks walked four rows at a time: 14.2 us in the step at 1024 x 512)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float *gy, const float *x, int64_t ldx, const float *gamma,
                                                             const float *beta, const float *save_mean, const float *save_rstd,
                                                             int64_t M, int64_t N, int relu, const float *mask, float scale,
                                                             float *part, const float *yout) {
    __shared__ float red[2][4][64];
    constexpr int R = BN2_BWD_ROWS / 4;
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 64 + c;
    const int64_t m0 = (int64_t)blockIdx.y * BN2_BWD_ROWS;
    float sg = 0.0f, sb = 0.0f;
    if (n < N) {
        float xv[R], gv[R], ov[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            const int64_t m = m0 + rl + 4 * j;
            const bool ok = m < M;
            xv[j] = ok ? x[m * ldx + n] : 0.0f;
            gv[j] = ok ? gy[m * N + n] : 0.0f;
            ov[j] = 0.0f;
            if (ok && yout != nullptr) ov[j] = yout[m * N + n];
            else if (ok && mask != nullptr) ov[j] = mask[m * N + n];
        }
        const float mu = save_mean[n], rs = save_rstd[n], ga = gamma[n], be = beta[n];
#pragma unroll
        for (int j = 0; j < R; j++) {
            if (m0 + rl + 4 * j < M) {
                const float xh = (xv[j] - mu) * rs;
                float g = gv[j];  // bn_gate on the values already loaded
                if (yout != nullptr) g = relu ? (ov[j] > 0.0f ? g * scale : 0.0f) : g;
                else {
                    if (mask != nullptr) g = g * (ov[j] * scale);
                    if (relu && !(xh * ga + be > 0.0f)) g = 0.0f;
                }
                sg += g * xh;
                sb += g;
            }
        }
    }
    red[0][rl][c] = sg;
    red[1][rl][c] = sb;
    __syncthreads();
    if (rl == 0 && n < N) {
        part[((int64_t)blockIdx.y * 2 + 0) * N + n] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        part[((int64_t)blockIdx.y * 2 + 1) * N + n] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *gy, const float *x, int64_t ldx, const float *gamma,
                                                           const float *beta, const float *save_mean, const float *save_rstd,
                                                           int64_t M, int64_t N, int relu, const float *mask, float scale,
                                                           const float *part, int64_t chunks, float *gx, float *ggamma, float *gbeta,
                                                           int accumulate, const float *yout) {
    __shared__ float s_g[64], s_b[64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 64 + c;
    if (rl == 0 && n < N) {
        float a = 0.0f, b = 0.0f;
        for (int64_t k0 = 0; k0 < chunks; k0 += 8) {  // eight chunks' loads in flight, added in the same ascending order
            float va[8], vb[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                va[j] = k0 + j < chunks ? part[((k0 + j) * 2 + 0) * N + n] : 0.0f;
                vb[j] = k0 + j < chunks ? part[((k0 + j) * 2 + 1) * N + n] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) { a += va[j]; b += vb[j]; }
        }
        s_g[c] = a;
        s_b[c] = b;
        if (blockIdx.y == 0) {
            ggamma[n] = accumulate ? ggamma[n] + a : a;
            gbeta[n] = accumulate ? gbeta[n] + b : b;
        }
    }
    __syncthreads();
    if (n >= N || gx == nullptr) return;
    const float sg = s_g[c], sb = s_b[c];
    const float mu = save_mean[n], rs = save_rstd[n], ga = gamma[n], be = beta[n];
    const float k = ga * rs / (float)M;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int64_t m = (int64_t)blockIdx.y * 16 + rl + 4 * j;
        if (m >= M) continue;
        const float xh = (x[m * ldx + n] - mu) * rs;
        const float g = bn_gate(gy[m * N + n], xh, ga, be, relu, mask, yout, m * N + n, scale);
        gx[m * N + n] = k * (((float)M * g - sb) - xh * sg);
    }
}

// ------------------------------------------------------------------------------------------------
// InfoNCE rows (loss.py:70-77): S = cn tn^T already in `S` [B,B]; per row b: loss_b = logsumexp_j(S_bj/tau) - S_bb/tau,
// and S is overwritten with the softmax P (the backward needs it).  row_loss [B].
// ------------------------------------------------------------------------------------------------
// B <= 64 * NV: the row lives in registers (one pass over memory, one expf per element; the generic kernel below reads the row three
// times and takes the exponential twice, each of its loops paying the load latency per trip).  Same operations in the same order.
template <int NV>
__global__ __launch_bounds__(256) void infonce_rows_resident_kernel(float *S, int64_t B, float inv_tau, float *row_loss) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    float *s = S + row * B;
    float v[NV];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NV; t++) {
        const int64_t j = lane + 64 * t;
        v[t] = j < B ? s[j] * inv_tau : -INFINITY;
    }
#pragma unroll
    for (int t = 0; t < NV; t++)
        if (lane + 64 * t < B) mx = fmaxf(mx, v[t]);
    mx = hv_wave_max(mx);
    float sum = 0.0f;
#pragma unroll
    for (int t = 0; t < NV; t++)
        if (lane + 64 * t < B) { v[t] = expf(v[t] - mx); sum += v[t]; }
    sum = hv_wave_sum(sum);
    const float lse = mx + logf(sum);
    const float diag = s[row] * inv_tau;
#pragma unroll
    for (int t = 0; t < NV; t++)
        if (lane + 64 * t < B) s[lane + 64 * t] = v[t] / sum;
    if (lane == 0) row_loss[row] = lse - diag;
}

// A row shared by four waves (B <= 256 NV): one wave per row is one wave per SIMD at 1024 rows, its load -> max -> exp -> sum -> divide
// chain exposed end to end (12 us alone, 15-20 us in the step for 8 MB of traffic).  Wave w owns the columns [256 w NV', ...) in
// 64-column stripes; the row maximum and the row sum are combined over the four waves in wave order through LDS.
template <int NV>
__global__ __launch_bounds__(256) void infonce_rows_split_kernel(float *S, int64_t B, float inv_tau, float *row_loss) {
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t row = blockIdx.x;
    float *s = S + row * B;
    float v[NV];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NV; t++) {
        const int64_t j = 64 * (wave * NV + t) + lane;
        v[t] = j < B ? s[j] * inv_tau : -INFINITY;
        mx = fmaxf(mx, v[t]);
    }
    const float diag = s[row] * inv_tau;
    mx = hv_wave_max(mx);
    if (lane == 0) red[0][wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    float sum = 0.0f;
#pragma unroll
    for (int t = 0; t < NV; t++)
        if (64 * (wave * NV + t) + lane < B) { v[t] = expf(v[t] - mx); sum += v[t]; }
    sum = hv_wave_sum(sum);
    if (lane == 0) red[1][wave] = sum;
    __syncthreads();
    sum = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
#pragma unroll
    for (int t = 0; t < NV; t++) {
        const int64_t j = 64 * (wave * NV + t) + lane;
        if (j < B) s[j] = v[t] / sum;
    }
    if (threadIdx.x == 0) row_loss[row] = (mx + logf(sum)) - diag;
}

__global__ __launch_bounds__(256) void infonce_rows_kernel(float *S, int64_t B, float inv_tau, float *row_loss) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    float *s = S + row * B;
    float mx = -INFINITY;
    for (int64_t j = lane; j < B; j += 64) mx = fmaxf(mx, s[j] * inv_tau);
    mx = hv_wave_max(mx);
    float sum = 0.0f;
    for (int64_t j = lane; j < B; j += 64) sum += expf(s[j] * inv_tau - mx);
    sum = hv_wave_sum(sum);
    const float lse = mx + logf(sum);
    const float diag = s[row] * inv_tau;
    for (int64_t j = lane; j < B; j += 64) s[j] = expf(s[j] * inv_tau - mx) / sum;
    if (lane == 0) row_loss[row] = lse - diag;
}

// ---- InfoNCE without the B x B matrix (large batches; the 20,000-item k-means warm-up forward would hold 3 x 1.6 GB of softmax) ----
// The similarity matrix is produced in column chunks Sc [B, C] = cn tn[col0:col0+C]^T and consumed at once: an online
// logsumexp per row (running maximum m and sum l, rescaled when the maximum moves; chunks in ascending column order, so the result
// is a fixed function of the chunk width), the diagonal logit captured when its column passes by.
__global__ __launch_bounds__(256) void infonce_lse_chunk_kernel(const float *Sc, int64_t B, int64_t C, int64_t ldc, int64_t col0,
                                                                float inv_tau, float *m, float *l, float *diag, int first) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float *s = Sc + row * ldc;
    float mx = -INFINITY;
    for (int64_t j = lane; j < C; j += 64) mx = fmaxf(mx, s[j] * inv_tau);
    mx = hv_wave_max(mx);
    const float m_old = first ? -INFINITY : m[row];
    const float m_new = fmaxf(m_old, mx);
    float sum = 0.0f;
    for (int64_t j = lane; j < C; j += 64) sum += expf(s[j] * inv_tau - m_new);
    sum = hv_wave_sum(sum);
    if (lane == 0) {
        const float l_old = first ? 0.0f : l[row] * expf(m_old - m_new);
        m[row] = m_new;
        l[row] = l_old + sum;
        if (row >= col0 && row < col0 + C) diag[row] = s[row - col0];
    }
}
__global__ __launch_bounds__(256) void infonce_lse_finish_kernel(const float *m, const float *l, const float *diag, int64_t B, float inv_tau,
                                                                 float *row_loss, float *lse) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const float v = m[i] + logf(l[i]);
    lse[i] = v;
    row_loss[i] = v - diag[i] * inv_tau;
}
// backward of a chunk: Sc -> dS = (g * coef) * (softmax - I) restricted to columns [col0, col0+C), softmax = exp(Sc/tau - lse)
__global__ __launch_bounds__(256) void infonce_dlogits_chunk_kernel(float *Sc, int64_t B, int64_t C, int64_t ldc, int64_t col0, float inv_tau,
                                                                    const float *lse, const float *g, float coef) {
    const float k = *g * coef;
    const int64_t n = B * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / C, c = i - r * C;
        const float p = expf(Sc[r * ldc + c] * inv_tau - lse[r]);
        Sc[r * ldc + c] = k * (p - (r == col0 + c ? 1.0f : 0.0f));
    }
}

// out[0] = scale * mean(v)   (single workgroup, fixed order)
__global__ __launch_bounds__(256) void vec_mean_kernel(const float *v, int64_t n, float scale, float *out) {
    __shared__ float red[4];
    float s = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += v[i];
    s = hv_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = scale * (((red[0] + red[1]) + (red[2] + red[3])) / (float)n);
}

// dS = (g * coef) * (P - I)     coef = scale / (B * tau), g = upstream device scalar
__global__ __launch_bounds__(256) void infonce_dlogits_kernel(float *P, int64_t B, const float *g, float coef) {
    const float k = *g * coef;
    const int64_t n = B * B;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / B, c = i - r * B;
        P[i] = k * (P[i] - (r == c ? 1.0f : 0.0f));
    }
}

// ------------------------------------------------------------------------------------------------
// Tag prediction loss rows (loss.py:107-265, layer_idx = 0 semantics).  One wave per row.
//   partner[b]: mixup partner row (original indexing) or -1 => no mixup for the whole call when partner == nullptr
//   mode 1 = focal with smoothing, 0 = CE(label_smoothing=0.05) + 0.05*KL(uniform || softmax(unmixed)+1e-8)
// Writes per row: row_loss (0 for invalid rows), row_hit (argmax(unmixed) == target), and -- if dmix != nullptr --
// dmix[b,:] = d row_loss / d mixed_logits[b,:] and (CE mode) dkl[b,:] = d kl_row / d logits[b,:].
// ------------------------------------------------------------------------------------------------
struct TagLossArgs {
    const float *logits;
    int64_t B, C;
    const int64_t *target;
    const int64_t *partner;
    const float *lam_dev;  // mixup weight (device scalar: it changes every step, also under graph replay)
    int focal;
    float gamma, alpha, smooth, ce_ls;
    float *row_loss, *row_hit, *dmix, *dkl;
};

__device__ __forceinline__ float focal_row(const float *z, int64_t C, int64_t cls, float smooth, float gamma, float alpha, int lane,
                                           float mx, float sum, float coef, float *dz) {
    // p_k = exp(z_k - mx)/sum; oh_k = (k==cls)(1-s) + s/C
    float pt = 0.0f, ce = 0.0f;
    const float lsum = logf(sum);
    for (int64_t k = lane; k < C; k += 64) {
        const float oh = (k == cls ? 1.0f - smooth : 0.0f) + smooth / (float)C;
        const float logp = (z[k] - mx) - lsum;
        pt += oh * expf(logp);
        ce -= oh * logp;
    }
    pt = hv_wave_sum(pt);
    ce = hv_wave_sum(ce);
    const float om = 1.0f - pt;
    const float w = alpha * powf(om, gamma);
    if (dz != nullptr) {
        const float dw = -alpha * gamma * powf(om, gamma - 1.0f);  // d w / d pt
        for (int64_t k = lane; k < C; k += 64) {
            const float oh = (k == cls ? 1.0f - smooth : 0.0f) + smooth / (float)C;
            const float p = expf((z[k] - mx) - lsum);
            dz[k] += coef * (dw * (p * (oh - pt)) * ce + w * (p - oh));
        }
    }
    return w * ce;
}

__device__ __forceinline__ float ce_row(const float *z, int64_t C, int64_t cls, float ls, int lane, float mx, float sum, float coef,
                                        float *dz) {
    const float lsum = logf(sum);
    float nll = 0.0f, all = 0.0f;
    for (int64_t k = lane; k < C; k += 64) {
        const float logp = (z[k] - mx) - lsum;
        if (k == cls) nll = -logp;
        all -= logp;
    }
    nll = hv_wave_sum(nll);
    all = hv_wave_sum(all);
    if (dz != nullptr)
        for (int64_t k = lane; k < C; k += 64) {
            const float p = expf((z[k] - mx) - lsum);
            const float oh = (k == cls ? 1.0f - ls : 0.0f) + ls / (float)C;
            dz[k] += coef * (p - oh);
        }
    return (1.0f - ls) * nll + ls * (all / (float)C);
}

__global__ __launch_bounds__(256) void tag_loss_rows_kernel(TagLossArgs a, float *zbuf) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.B) return;
    const int64_t cls = a.target[row];
    const int64_t C = a.C;
    float *dz = a.dmix != nullptr ? a.dmix + row * C : nullptr;
    if (dz != nullptr)
        for (int64_t k = lane; k < C; k += 64) dz[k] = 0.0f;
    if (a.dkl != nullptr)
        for (int64_t k = lane; k < C; k += 64) a.dkl[row * C + k] = 0.0f;
    if (cls < 0) {
        if (lane == 0) { a.row_loss[row] = 0.0f; a.row_hit[row] = 0.0f; }
        return;
    }
    const float *z0 = a.logits + row * C;
    // accuracy on the un-mixed logits (first maximum, like torch.argmax)
    float bm = -INFINITY;
    int64_t bi = 0;
    for (int64_t k = lane; k < C; k += 64)
        if (z0[k] > bm) { bm = z0[k]; bi = k; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(bm, o);
        const int64_t oi = __shfl_xor(bi, o);
        if (om > bm || (om == bm && oi < bi)) { bm = om; bi = oi; }
    }
    // mixed logits (loss.py:150) into this row's scratch
    float *z = zbuf + row * C;
    const int64_t pr = a.partner != nullptr ? a.partner[row] : -1;
    const bool mixed = pr >= 0;
    const float lam = mixed ? *a.lam_dev : 1.0f;
    for (int64_t k = lane; k < C; k += 64) z[k] = mixed ? lam * z0[k] + (1.0f - lam) * a.logits[pr * C + k] : z0[k];
    float mx = -INFINITY;
    for (int64_t k = lane; k < C; k += 64) mx = fmaxf(mx, z[k]);
    mx = hv_wave_max(mx);
    float sum = 0.0f;
    for (int64_t k = lane; k < C; k += 64) sum += expf(z[k] - mx);
    sum = hv_wave_sum(sum);
    float loss;
    if (a.focal) {
        loss = focal_row(z, C, cls, a.smooth, a.gamma, a.alpha, lane, mx, sum, lam, dz);
        if (mixed) loss = lam * loss + (1.0f - lam) * focal_row(z, C, a.target[pr], a.smooth, a.gamma, a.alpha, lane, mx, sum, 1.0f - lam, dz);
    } else {
        loss = ce_row(z, C, cls, a.ce_ls, lane, mx, sum, lam, dz);
        if (mixed) loss = lam * loss + (1.0f - lam) * ce_row(z, C, a.target[pr], a.ce_ls, lane, mx, sum, 1.0f - lam, dz);
        // + 0.05 * KL(uniform || softmax(unmixed) + 1e-8), batchmean  (loss.py:222-223)
        float m0 = -INFINITY;
        for (int64_t k = lane; k < C; k += 64) m0 = fmaxf(m0, z0[k]);
        m0 = hv_wave_max(m0);
        float s0 = 0.0f;
        for (int64_t k = lane; k < C; k += 64) s0 += expf(z0[k] - m0);
        s0 = hv_wave_sum(s0);
        const float u = 1.0f / (float)C, logu = logf(u);
        float kl = 0.0f, t = 0.0f;
        for (int64_t k = lane; k < C; k += 64) {
            const float p = expf(z0[k] - m0) / s0;
            kl += u * (logu - logf(p + 1e-8f));
            t += p / (p + 1e-8f) * p;  // sum_j w_j p_j, w_j = p_j/(p_j+eps)
        }
        kl = hv_wave_sum(kl);
        t = hv_wave_sum(t);
        loss += 0.05f * kl;
        if (a.dkl != nullptr) {
            // d/dz_k [ -u sum_j log(p_j+eps) ] = -u ( w_k p_k - p_k sum_j w_j p_j )... with w_j = p_j/(p_j+eps):
            // d log(p_j+eps)/dz_k = (p_j/(p_j+eps)) (delta_jk - p_k)
            float wsum = 0.0f;
            for (int64_t k = lane; k < C; k += 64) { const float p = expf(z0[k] - m0) / s0; wsum += p / (p + 1e-8f); }
            wsum = hv_wave_sum(wsum);
            for (int64_t k = lane; k < C; k += 64) {
                const float p = expf(z0[k] - m0) / s0;
                a.dkl[row * C + k] = 0.05f * (-u) * (p / (p + 1e-8f) - p * wsum);
            }
        }
        (void)t;
    }
    if (lane == 0) { a.row_loss[row] = loss; a.row_hit[row] = (bi == cls) ? 1.0f : 0.0f; }
}

// The same row with its C <= 64 * NV logits held in REGISTERS: lane l owns classes l, l + 64, ...  The kernel above walks the row in
// ~10 dependent passes through memory (the mixed row goes to scratch and comes back, dz is zeroed, read and rewritten: 12 us in the
// step for 1024 x 348); here every pass is register arithmetic in the SAME per-lane order of the same operations, so the results are
// bit-identical, and memory is touched once per operand.
template <int NV>
__device__ __forceinline__ float focal_row_reg(const float (&z)[NV], int64_t C, int64_t cls, float smooth, float gamma, float alpha, int lane,
                                               float mx, float sum, float coef, float (&dz)[NV], bool want_dz) {
    float pt = 0.0f, ce = 0.0f;
    const float lsum = logf(sum);
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int64_t k = lane + 64 * j;
        if (k < C) {
            const float oh = (k == cls ? 1.0f - smooth : 0.0f) + smooth / (float)C;
            const float logp = (z[j] - mx) - lsum;
            pt += oh * expf(logp);
            ce -= oh * logp;
        }
    }
    pt = hv_wave_sum(pt);
    ce = hv_wave_sum(ce);
    const float om = 1.0f - pt;
    const float w = alpha * powf(om, gamma);
    if (want_dz) {
        const float dw = -alpha * gamma * powf(om, gamma - 1.0f);
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int64_t k = lane + 64 * j;
            if (k < C) {
                const float oh = (k == cls ? 1.0f - smooth : 0.0f) + smooth / (float)C;
                const float p = expf((z[j] - mx) - lsum);
                dz[j] += coef * (dw * (p * (oh - pt)) * ce + w * (p - oh));
            }
        }
    }
    return w * ce;
}

template <int NV>
__device__ __forceinline__ float ce_row_reg(const float (&z)[NV], int64_t C, int64_t cls, float ls, int lane, float mx, float sum, float coef,
                                            float (&dz)[NV], bool want_dz) {
    const float lsum = logf(sum);
    float nll = 0.0f, all = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int64_t k = lane + 64 * j;
        if (k < C) {
            const float logp = (z[j] - mx) - lsum;
            if (k == cls) nll = -logp;
            all -= logp;
        }
    }
    nll = hv_wave_sum(nll);
    all = hv_wave_sum(all);
    if (want_dz) {
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int64_t k = lane + 64 * j;
            if (k < C) {
                const float p = expf((z[j] - mx) - lsum);
                const float oh = (k == cls ? 1.0f - ls : 0.0f) + ls / (float)C;
                dz[j] += coef * (p - oh);
            }
        }
    }
    return (1.0f - ls) * nll + ls * (all / (float)C);
}

template <int NV>
__global__ __launch_bounds__(256) void tag_loss_rows_reg_kernel(TagLossArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.B) return;
    const int64_t cls = a.target[row];
    const int64_t C = a.C;
    const bool want_dz = a.dmix != nullptr;
    if (cls < 0) {
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int64_t k = lane + 64 * j;
            if (k < C) {
                if (want_dz) a.dmix[row * C + k] = 0.0f;
                if (a.dkl != nullptr) a.dkl[row * C + k] = 0.0f;
            }
        }
        if (lane == 0) { a.row_loss[row] = 0.0f; a.row_hit[row] = 0.0f; }
        return;
    }
    const int64_t pr = a.partner != nullptr ? a.partner[row] : -1;
    const bool mixed = pr >= 0;
    const float lam = mixed ? *a.lam_dev : 1.0f;
    const int64_t cls2 = mixed ? a.target[pr] : 0;
    float z0[NV], zp[NV], z[NV], dz[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int64_t k = lane + 64 * j;
        z0[j] = k < C ? a.logits[row * C + k] : 0.0f;
        zp[j] = (mixed && k < C) ? a.logits[pr * C + k] : 0.0f;
        dz[j] = 0.0f;
    }
    // accuracy on the un-mixed logits (first maximum, like torch.argmax)
    float bm = -INFINITY;
    int64_t bi = 0;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int64_t k = lane + 64 * j;
        if (k < C && z0[j] > bm) { bm = z0[j]; bi = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(bm, o);
        const int64_t oi = __shfl_xor(bi, o);
        if (om > bm || (om == bm && oi < bi)) { bm = om; bi = oi; }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        z[j] = mixed ? lam * z0[j] + (1.0f - lam) * zp[j] : z0[j];  // loss.py:150
        if (lane + 64 * j < C) mx = fmaxf(mx, z[j]);
    }
    mx = hv_wave_max(mx);
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; j++)
        if (lane + 64 * j < C) sum += expf(z[j] - mx);
    sum = hv_wave_sum(sum);
    float loss;
    if (a.focal) {
        loss = focal_row_reg<NV>(z, C, cls, a.smooth, a.gamma, a.alpha, lane, mx, sum, lam, dz, want_dz);
        if (mixed) loss = lam * loss + (1.0f - lam) * focal_row_reg<NV>(z, C, cls2, a.smooth, a.gamma, a.alpha, lane, mx, sum, 1.0f - lam, dz, want_dz);
    } else {
        loss = ce_row_reg<NV>(z, C, cls, a.ce_ls, lane, mx, sum, lam, dz, want_dz);
        if (mixed) loss = lam * loss + (1.0f - lam) * ce_row_reg<NV>(z, C, cls2, a.ce_ls, lane, mx, sum, 1.0f - lam, dz, want_dz);
        float m0 = -INFINITY;
#pragma unroll
        for (int j = 0; j < NV; j++)
            if (lane + 64 * j < C) m0 = fmaxf(m0, z0[j]);
        m0 = hv_wave_max(m0);
        float s0 = 0.0f;
#pragma unroll
        for (int j = 0; j < NV; j++)
            if (lane + 64 * j < C) s0 += expf(z0[j] - m0);
        s0 = hv_wave_sum(s0);
        const float u = 1.0f / (float)C, logu = logf(u);
        float kl = 0.0f;
#pragma unroll
        for (int j = 0; j < NV; j++)
            if (lane + 64 * j < C) {
                const float p = expf(z0[j] - m0) / s0;
                kl += u * (logu - logf(p + 1e-8f));
            }
        kl = hv_wave_sum(kl);
        loss += 0.05f * kl;
        if (a.dkl != nullptr) {
            float wsum = 0.0f;
#pragma unroll
            for (int j = 0; j < NV; j++)
                if (lane + 64 * j < C) { const float p = expf(z0[j] - m0) / s0; wsum += p / (p + 1e-8f); }
            wsum = hv_wave_sum(wsum);
#pragma unroll
            for (int j = 0; j < NV; j++) {
                const int64_t k = lane + 64 * j;
                if (k < C) {
                    const float p = expf(z0[j] - m0) / s0;
                    a.dkl[row * C + k] = 0.05f * (-u) * (p / (p + 1e-8f) - p * wsum);
                }
            }
        }
    } 
    if (want_dz) {
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int64_t k = lane + 64 * j;
            if (k < C) a.dmix[row * C + k] = dz[j];
        }
    }
    if (a.focal && a.dkl != nullptr) {
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int64_t k = lane + 64 * j;
            if (k < C) a.dkl[row * C + k] = 0.0f;
        }
    }
    if (lane == 0) { a.row_loss[row] = loss; a.row_hit[row] = (bi == cls) ? 1.0f : 0.0f; }
}

// loss = sum(row_loss)/n_valid, acc = sum(row_hit)/n_valid (0,0 if no valid row: loss.py:123-125); single workgroup
__global__ __launch_bounds__(256) void tag_loss_reduce_kernel(const float *row_loss, const float *row_hit, const int64_t *target,
                                                              int64_t B, float *loss, float *acc, float *n_valid_out) {
    __shared__ float red[3][4];
    float l = 0.0f, h = 0.0f, n = 0.0f;
    for (int64_t i = threadIdx.x; i < B; i += 256) {
        if (target[i] >= 0) { l += row_loss[i]; h += row_hit[i]; n += 1.0f; }
    }
    l = hv_wave_sum(l); h = hv_wave_sum(h); n = hv_wave_sum(n);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = l; red[1][threadIdx.x >> 6] = h; red[2][threadIdx.x >> 6] = n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float ls = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const float hs = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const float ns = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
        *loss = ns > 0.0f ? ls / ns : 0.0f;
        *acc = ns > 0.0f ? hs / ns : 0.0f;
        *n_valid_out = ns;
    }
}

// L = (1/n) sum_b row_loss_b with mixed_b = lam z_b + (1-lam) z_partner(b)  =>
// g_logits[r,:] = (g / n_valid) * ( lam*dmix[r,:] + (1-lam)*dmix[inverse[r],:] + dkl[r,:] )   (invalid rows: 0)
__global__ __launch_bounds__(256) void tag_loss_bwd_kernel(const float *dmix, const float *dkl, const int64_t *target,
                                                           const int64_t *inverse, const float *lam_dev, int64_t B, int64_t C,
                                                           const float *g, const float *n_valid, float *g_logits) {
    const float nv = *n_valid;
    const float lam = inverse != nullptr ? *lam_dev : 1.0f;
    const float k = nv > 0.0f ? *g / nv : 0.0f;
    const int64_t n = B * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / C, c = i - r * C;
        float v = 0.0f;
        if (target[r] >= 0) {
            if (inverse != nullptr) v = lam * dmix[i] + (1.0f - lam) * dmix[inverse[r] * C + c];
            else v = dmix[i];
            if (dkl != nullptr) v += dkl[i];
        }
        g_logits[i] = k * v;
    }
}

}  // namespace

static inline unsigned ew_grid(int64_t n) {
    int64_t g = hv_cdiv(n, 256);
    return (unsigned)(g < 2048 ? (g < 1 ? 1 : g) : 2048);
}

extern "C" int hidvae_act_bwd(const float *g, const float *ref, int64_t numel, int act, const float *mask, float mask_scale,
                              float *out, void *stream) {
    HV_REQUIRE(g && out && numel >= 1, "act_bwd: bad arguments");
    HV_REQUIRE(act == HIDVAE_EPI_NONE || ref != nullptr, "act_bwd: activation %d needs its reference tensor", act);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(numel)), dim3(256), 0, (hipStream_t)stream, g, ref, numel, act, mask, mask_scale, out);
    HV_LAUNCH_CHECK("act_bwd");
    return HIDVAE_OK;
}

extern "C" int hidvae_binary(int op, const float *a, int64_t lda, const float *b, int64_t ldb, int64_t M, int64_t N, float *out,
                             int64_t ldo, void *stream) {
    HV_REQUIRE(a && b && out && M >= 1 && N >= 1 && lda >= N && ldb >= N && ldo >= N && op >= 0 && op <= 2, "binary: bad arguments");
    hipLaunchKernelGGL(mul_kernel, dim3(ew_grid(M * N)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, M, N, out, ldo, op);
    HV_LAUNCH_CHECK("binary");
    return HIDVAE_OK;
}

extern "C" int hidvae_sum_prefix_slices(const float *const *src_host, const int32_t *width_host, int n, int64_t M, int64_t N,
                                        float *dst, void *stream) {
    HV_REQUIRE(src_host && width_host && dst && n >= 0 && n <= MAX_SLICES && M >= 1 && N >= 1, "sum_prefix_slices: bad arguments");
    SliceArgs a{};
    a.n = 0;
    for (int i = 0; i < n; i++) {
        if (src_host[i] == nullptr) continue;
        HV_REQUIRE(width_host[i] >= 1 && width_host[i] <= N, "sum_prefix_slices: width %d", width_host[i]);
        a.src[a.n] = src_host[i];
        a.width[a.n] = width_host[i];
        a.n++;
    }
    a.M = M; a.N = N; a.dst = dst;
    hipLaunchKernelGGL(sum_prefix_slices_kernel, dim3(ew_grid(M * N)), dim3(256), 0, (hipStream_t)stream, a);
    HV_LAUNCH_CHECK("sum_prefix_slices");
    return HIDVAE_OK;
}

extern "C" int hidvae_layernorm_fwd(const float *x, int64_t M, int64_t N, const float *gamma, const float *beta, float eps, float *y,
                                    float *mean, float *rstd, int relu, const float *keep_mask, float keep_scale,
                                    const float *residual, const unsigned long long *rng_state, unsigned rng_site, unsigned drop_threshold,
                                    void *stream) {
    HV_REQUIRE(x && gamma && beta && y && mean && rstd && M >= 1 && N >= 1, "layernorm_fwd: bad arguments");
    HV_REQUIRE(!(keep_mask && rng_state), "layernorm_fwd: a keep-mask OR the in-kernel generator, not both");
    const HvDrop drop{rng_state, rng_site, drop_threshold};
    const dim3 grid4((unsigned)hv_cdiv(M, 4));
    if (N > 256 && N <= 512 && M <= 16384)
        hipLaunchKernelGGL((layernorm_fwd_split_kernel<4, 2>), grid4, dim3(512), 0, (hipStream_t)stream, x, M, N, gamma, beta, eps, y, mean, rstd,
                           relu, keep_mask, keep_scale, residual, drop);
    else if (N > 512 && N <= 768 && M <= 16384)
        hipLaunchKernelGGL((layernorm_fwd_split_kernel<3, 4>), grid4, dim3(1024), 0, (hipStream_t)stream, x, M, N, gamma, beta, eps, y, mean, rstd,
                           relu, keep_mask, keep_scale, residual, drop);
    else if (N > 768 && N <= 1024 && M <= 16384)
        hipLaunchKernelGGL((layernorm_fwd_split_kernel<4, 4>), grid4, dim3(1024), 0, (hipStream_t)stream, x, M, N, gamma, beta, eps, y, mean, rstd,
                           relu, keep_mask, keep_scale, residual, drop);
    else if (N <= 1024)
        hipLaunchKernelGGL(layernorm_fwd_kernel<true>, dim3((unsigned)hv_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, M, N, gamma, beta,
                           eps, y, mean, rstd, relu, keep_mask, keep_scale, residual, drop);
    else
    hipLaunchKernelGGL(layernorm_fwd_kernel<false>, dim3((unsigned)hv_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, M, N, gamma, beta,
                       eps, y, mean, rstd, relu, keep_mask, keep_scale, residual, drop);
    HV_LAUNCH_CHECK("layernorm_fwd");
    return HIDVAE_OK;
}

extern "C" int hidvae_layernorm_bwd(const float *gy, const float *x, const float *gamma, const float *beta, const float *mean,
                                    const float *rstd, int64_t M, int64_t N, int relu, const float *keep_mask, float keep_scale,
                                    float *gx, void *stream) {
    HV_REQUIRE(gy && x && gamma && beta && mean && rstd && gx && M >= 1 && N >= 1, "layernorm_bwd: bad arguments");
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)hv_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, gy, x, gamma, beta, mean,
                       rstd, M, N, relu, keep_mask, keep_scale, gx);
    HV_LAUNCH_CHECK("layernorm_bwd");
    return HIDVAE_OK;
}

extern "C" int hidvae_layernorm_param_grad(const float *gy, const float *x, const float *gamma, const float *beta, const float *mean,
                                           const float *rstd, int64_t M, int64_t N, int relu, const float *keep_mask,
                                           float keep_scale, float *ggamma, float *gbeta, int accumulate, float *workspace,
                                           void *stream) {
    HV_REQUIRE(gy && x && gamma && beta && mean && rstd && ggamma && gbeta && workspace && M >= 1 && N >= 1,
               "layernorm_param_grad: bad arguments");
    const int64_t chunks = hv_cdiv(M, LN_CHUNK);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(layernorm_param_partial_kernel, dim3((unsigned)hv_cdiv(N, 64), (unsigned)chunks), dim3(256), 0, s, gy, x, gamma,
                       beta, mean, rstd, M, N, relu, keep_mask, keep_scale, workspace);
    HV_LAUNCH_CHECK("layernorm_param_partial");
    hipLaunchKernelGGL(layernorm_param_final_kernel, dim3((unsigned)hv_cdiv(N, 32)), dim3(1024), 0, s, workspace, chunks, N, ggamma,
                       gbeta, accumulate);
    HV_LAUNCH_CHECK("layernorm_param_final");
    return HIDVAE_OK;
}

extern "C" int hidvae_layernorm_bwd_all(const float *gy, const float *x, const float *gamma, const float *beta, const float *mean,
                                        const float *rstd, int64_t M, int64_t N, int relu, const float *keep_mask, float keep_scale,
                                        float *gx, float *ggamma, float *gbeta, int accumulate, float *workspace, void *stream) {
    HV_REQUIRE(gy && x && gamma && beta && mean && rstd && ggamma && gbeta && workspace && M >= 1 && N >= 1,
               "layernorm_bwd_all: bad arguments");
    if (N > 64 * LNF_NV) {  // rows too wide for the register-resident form: the separate kernels
        if (gx != nullptr) {
            const int rc = hidvae_layernorm_bwd(gy, x, gamma, beta, mean, rstd, M, N, relu, keep_mask, keep_scale, gx, stream);
            if (rc != HIDVAE_OK) return rc;
        }
        return hidvae_layernorm_param_grad(gy, x, gamma, beta, mean, rstd, M, N, relu, keep_mask, keep_scale, ggamma, gbeta, accumulate,
                                           workspace, stream);
    }
    const int64_t chunks = hv_cdiv(M, LNF_ROWS);
    hipStream_t s = (hipStream_t)stream;
    launch_layernorm_bwd_fused(chunks, s, gy, x, gamma, beta, mean, rstd, M, N, relu, keep_mask, keep_scale, gx, workspace, nullptr, 0.0f, nullptr,
                               nullptr);
    HV_LAUNCH_CHECK("layernorm_bwd_fused");
    hipLaunchKernelGGL(layernorm_param_final_kernel, dim3((unsigned)hv_cdiv(N, 32)), dim3(1024), 0, s, workspace, chunks, N, ggamma,
                       gbeta, accumulate);
    HV_LAUNCH_CHECK("layernorm_param_final");
    return HIDVAE_OK;
}

extern "C" int hidvae_layernorm_bwd_partial(const float *gy, const float *x, const float *gamma, const float *beta, const float *mean,
                                            const float *rstd, int64_t M, int64_t N, int relu, const float *y_out, float keep_scale,
                                            float in_relu_scale, const float *gy2, float *gsum, float *gx, float *partials, void *stream) {
    HV_REQUIRE(gy && x && gamma && beta && mean && rstd && partials && M >= 1 && N >= 1, "layernorm_bwd_partial: bad arguments");
    HV_REQUIRE(N <= 64 * LNF_NV, "layernorm_bwd_partial: N=%lld exceeds the register-resident form (%d)", (long long)N, 64 * LNF_NV);
    HV_REQUIRE(!relu || y_out != nullptr, "layernorm_bwd_partial: the ReLU gate is read off the forward output");
    const int64_t chunks = hv_cdiv(M, LNF_ROWS);
    launch_layernorm_bwd_fused(chunks, (hipStream_t)stream, gy, x, gamma, beta, mean, rstd, M, N, relu, nullptr, keep_scale, gx, partials, y_out,
                               in_relu_scale, gy2, gsum);
    HV_LAUNCH_CHECK("layernorm_bwd_partial");
    return HIDVAE_OK;
}

extern "C" int hidvae_layernorm_param_final_many(const hidvae_ln_final *pr, int n, void *stream) {
    HV_REQUIRE(pr != nullptr && n >= 1, "layernorm_param_final_many: bad arguments");
    for (int i0 = 0; i0 < n; i0 += LN_FINAL_MAX) {
        LnFinalArgs a{};
        a.n = (n - i0 < LN_FINAL_MAX) ? n - i0 : LN_FINAL_MAX;
        int blocks = 0;
        for (int i = 0; i < a.n; i++) {
            const hidvae_ln_final &q = pr[i0 + i];
            HV_REQUIRE(q.partials && q.ggamma && q.gbeta && q.M >= 1 && q.N >= 1 && q.N <= 64 * LNF_NV, "layernorm_param_final_many: problem %d is malformed", i0 + i);
            a.p[i] = q;
            a.nb[i] = (int)hv_cdiv(q.N, 32);
            blocks += a.nb[i];
        }
        hipLaunchKernelGGL(layernorm_param_final_many_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, a);
        HV_LAUNCH_CHECK("layernorm_param_final_many");
    }
    return HIDVAE_OK;
}

extern "C" int hidvae_batchnorm_fwd(const float *x, int64_t ldx, int64_t M, int64_t N, const float *gamma, const float *beta, float eps,
                                    float momentum, int training, float *running_mean, float *running_var,
                                    int64_t *num_batches_tracked, float *y, float *save_mean, float *save_rstd, int relu,
                                    const float *keep_mask, float keep_scale, const unsigned long long *rng_state, unsigned rng_site,
                                    unsigned drop_threshold, float *workspace, void *stream) {
    HV_REQUIRE(x && y && M >= 1 && N >= 1 && ldx >= N, "batchnorm_fwd: bad arguments");
    HV_REQUIRE(!(keep_mask && rng_state), "batchnorm_fwd: a keep-mask OR the in-kernel generator, not both");
    const HvDrop drop{rng_state, rng_site, drop_threshold};
    HV_REQUIRE(training ? (save_mean && save_rstd) : (running_mean && running_var), "batchnorm_fwd: statistics buffers missing");
    if (workspace != nullptr || !training) {  // row-parallel form
        hipStream_t s = (hipStream_t)stream;
        const int64_t chunks = hv_cdiv(M, BN2_ROWS);
        if (training) {
            hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)hv_cdiv(N, 64), (unsigned)chunks), dim3(256), 0, s, x, ldx, M, N, workspace);
            HV_LAUNCH_CHECK("batchnorm_fwd stats");
        }
        hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)hv_cdiv(N, 64), (unsigned)hv_cdiv(M, 16)), dim3(256), 0, s, x, ldx, M, N, gamma, beta,
                           eps, momentum, training, running_mean, running_var, num_batches_tracked, workspace, chunks, y, save_mean,
                           save_rstd, relu, keep_mask, keep_scale, drop);
        HV_LAUNCH_CHECK("batchnorm_fwd apply");
        return HIDVAE_OK;
    }
    hipLaunchKernelGGL(batchnorm_fwd_kernel, dim3((unsigned)hv_cdiv(N, BN_COLS)), dim3(1024), 0, (hipStream_t)stream, x, ldx, M, N, gamma,
                       beta, eps, momentum, training, running_mean, running_var, num_batches_tracked, y, save_mean, save_rstd, relu, keep_mask,
                       keep_scale, drop);
    HV_LAUNCH_CHECK("batchnorm_fwd");
    return HIDVAE_OK;
}

extern "C" int hidvae_batchnorm_bwd(const float *gy, const float *x, int64_t ldx, const float *gamma, const float *beta,
                                    const float *save_mean, const float *save_rstd, int64_t M, int64_t N, int relu,
                                    const float *keep_mask, float keep_scale, const float *y_out, float *gx, float *ggamma, float *gbeta,
                                    int accumulate, float *workspace, void *stream) {
    HV_REQUIRE(gy && x && gamma && beta && save_mean && save_rstd && ggamma && gbeta && M >= 1 && N >= 1 && ldx >= N,
               "batchnorm_bwd: bad arguments");
    HV_REQUIRE(!(keep_mask && y_out), "batchnorm_bwd: the gate comes from the keep-mask OR from the forward output, not both");
    if (workspace != nullptr) {  // row-parallel form
        hipStream_t s = (hipStream_t)stream;
        const int64_t chunks = hv_cdiv(M, BN2_BWD_ROWS);
        hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3((unsigned)hv_cdiv(N, 64), (unsigned)chunks), dim3(256), 0, s, gy, x, ldx, gamma, beta,
                           save_mean, save_rstd, M, N, relu, keep_mask, keep_scale, workspace, y_out);
        HV_LAUNCH_CHECK("batchnorm_bwd partial");
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)hv_cdiv(N, 64), (unsigned)hv_cdiv(M, 16)), dim3(256), 0, s, gy, x, ldx, gamma,
                           beta, save_mean, save_rstd, M, N, relu, keep_mask, keep_scale, workspace, chunks, gx, ggamma, gbeta, accumulate, y_out);
        HV_LAUNCH_CHECK("batchnorm_bwd apply");
        return HIDVAE_OK;
    }
    hipLaunchKernelGGL(batchnorm_bwd_kernel, dim3((unsigned)hv_cdiv(N, BN_COLS)), dim3(1024), 0, (hipStream_t)stream, gy, x, ldx, gamma, beta,
                       save_mean, save_rstd, M, N, relu, keep_mask, keep_scale, gx, ggamma, gbeta, accumulate, y_out);
    HV_LAUNCH_CHECK("batchnorm_bwd");
    return HIDVAE_OK;
}

extern "C" int hidvae_infonce_rows(float *S, int64_t B, float tau, float scale, float *row_loss, float *loss, void *stream) {
    HV_REQUIRE(S && row_loss && loss && B >= 1 && tau > 0.0f, "infonce_rows: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)hv_cdiv(B, 4));
    const dim3 rows((unsigned)B);
    if (B <= 256) hipLaunchKernelGGL((infonce_rows_split_kernel<1>), rows, dim3(256), 0, s, S, B, 1.0f / tau, row_loss);
    else if (B <= 512) hipLaunchKernelGGL((infonce_rows_split_kernel<2>), rows, dim3(256), 0, s, S, B, 1.0f / tau, row_loss);
    else if (B <= 1024) hipLaunchKernelGGL((infonce_rows_split_kernel<4>), rows, dim3(256), 0, s, S, B, 1.0f / tau, row_loss);
    else if (B <= 2048) hipLaunchKernelGGL((infonce_rows_split_kernel<8>), rows, dim3(256), 0, s, S, B, 1.0f / tau, row_loss);
    else hipLaunchKernelGGL(infonce_rows_kernel, grid, dim3(256), 0, s, S, B, 1.0f / tau, row_loss);
    HV_LAUNCH_CHECK("infonce_rows");
    hipLaunchKernelGGL(vec_mean_kernel, dim3(1), dim3(256), 0, s, row_loss, B, scale, loss);
    HV_LAUNCH_CHECK("infonce mean");
    return HIDVAE_OK;
}

extern "C" int hidvae_infonce_lse_chunk(const float *Sc, int64_t B, int64_t C, int64_t ldc, int64_t col0, float tau, float *m, float *l,
                                        float *diag, int first, void *stream) {
    HV_REQUIRE(Sc && m && l && diag && B >= 1 && C >= 1 && ldc >= C && col0 >= 0 && col0 + C <= B && tau > 0.0f, "infonce_lse_chunk: bad arguments");
    hipLaunchKernelGGL(infonce_lse_chunk_kernel, dim3((unsigned)hv_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, Sc, B, C, ldc, col0,
                       1.0f / tau, m, l, diag, first);
    HV_LAUNCH_CHECK("infonce_lse_chunk");
    return HIDVAE_OK;
}

extern "C" int hidvae_infonce_lse_finish(const float *m, const float *l, const float *diag, int64_t B, float tau, float scale, float *row_loss,
                                         float *lse, float *loss, void *stream) {
    HV_REQUIRE(m && l && diag && row_loss && lse && loss && B >= 1 && tau > 0.0f, "infonce_lse_finish: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(infonce_lse_finish_kernel, dim3((unsigned)hv_cdiv(B, 256)), dim3(256), 0, s, m, l, diag, B, 1.0f / tau, row_loss, lse);
    HV_LAUNCH_CHECK("infonce_lse_finish");
    hipLaunchKernelGGL(vec_mean_kernel, dim3(1), dim3(256), 0, s, row_loss, B, scale, loss);
    HV_LAUNCH_CHECK("infonce mean");
    return HIDVAE_OK;
}

extern "C" int hidvae_infonce_dlogits_chunk(float *Sc, int64_t B, int64_t C, int64_t ldc, int64_t col0, float tau, float scale,
                                            const float *lse, const float *g_dev, void *stream) {
    HV_REQUIRE(Sc && lse && g_dev && B >= 1 && C >= 1 && ldc >= C && col0 >= 0 && col0 + C <= B && tau > 0.0f, "infonce_dlogits_chunk: bad arguments");
    hipLaunchKernelGGL(infonce_dlogits_chunk_kernel, dim3(ew_grid(B * C)), dim3(256), 0, (hipStream_t)stream, Sc, B, C, ldc, col0, 1.0f / tau,
                       lse, g_dev, scale / ((float)B * tau));
    HV_LAUNCH_CHECK("infonce_dlogits_chunk");
    return HIDVAE_OK;
}

extern "C" int hidvae_infonce_dlogits(float *P, int64_t B, float tau, float scale, const float *g_dev, void *stream) {
    HV_REQUIRE(P && g_dev && B >= 1 && tau > 0.0f, "infonce_dlogits: bad arguments");
    hipLaunchKernelGGL(infonce_dlogits_kernel, dim3(ew_grid(B * B)), dim3(256), 0, (hipStream_t)stream, P, B, g_dev,
                       scale / ((float)B * tau));
    HV_LAUNCH_CHECK("infonce_dlogits");
    return HIDVAE_OK;
}

// ------------------------------------------------------------------------------------------------
// Mixup plan of a training step (loss.py:139-147: perm = randperm(n_valid), lam ~ Beta(alpha, alpha)), all levels in one launch.
// The torch composition (rand / where / argsort / cumsum / gather / scatter_ / Beta.sample) is ~40 launches, ~190 us of the tagged
// step; here one workgroup per level sorts (uniform key, row) pairs of the valid rows in LDS (bitonic, ties by row), so
//   partner[b] = the row mixed into b  (a uniformly random permutation of the valid rows among themselves, -1 on invalid rows)
//   inverse[partner[b]] = b
// and one thread draws lam from two Marsaglia-Tsang gamma variates.  The randomness comes in as uniforms u [L, B + 64] from the
// caller's generator (B keys + 64 spare draws per level for the rejection sampler), so graph replay semantics stay torch's.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int MIX_MAXB = 16384, MIX_SPARE = 64;  // (rows: the sort's keys and row numbers live in LDS, 6 bytes per row)

__device__ float mix_gamma(float shape, const float *u, int &k) {  // Gamma(shape, 1), shape > 0
    const float boost = shape < 1.0f ? powf(fmaxf(u[k++ % MIX_SPARE], 1e-30f), 1.0f / shape) : 1.0f;  // G(a) = G(a+1) U^(1/a)
    const float a = shape < 1.0f ? shape + 1.0f : shape;
    const float d = a - 1.0f / 3.0f, c = 1.0f / sqrtf(9.0f * d);
    float v = 1.0f;
    for (int attempt = 0; attempt < 16; attempt++) {
        const float u1 = fmaxf(u[k++ % MIX_SPARE], 1e-30f), u2 = u[k++ % MIX_SPARE], u3 = fmaxf(u[k++ % MIX_SPARE], 1e-30f);
        const float x = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);  // Box-Muller
        v = 1.0f + c * x;
        if (v <= 0.0f) { v = 1.0f; continue; }
        v = v * v * v;
        const float x2 = x * x;
        if (u3 < 1.0f - 0.0331f * x2 * x2 || logf(u3) < 0.5f * x2 + d * (1.0f - v + logf(v))) break;
    }
    return d * v * boost;
}

constexpr unsigned MIX_SITE = 0x4d495800u;  // the mixup plan's randomness site (+ level): apart from every dropout site
__global__ __launch_bounds__(1024) void mixup_plan_kernel(const int64_t *targets, int64_t B, int64_t ldt, const float *u, float alpha,
                                                          int64_t *partner, int64_t *inverse, float *lam,
                                                          const unsigned long long *rng_state) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mix_lds[];  // key float[n2], then idx unsigned short[n2]
    __shared__ int part[1024];
    __shared__ float spare[MIX_SPARE];
    const int lvl = blockIdx.x, tid = threadIdx.x;
    // the level's uniforms: handed in (u, tests / injected draws) or drawn here from the counter-based generator (rng_state)
    const float *ul = u != nullptr ? u + (int64_t)lvl * (B + MIX_SPARE) : nullptr;
    if (tid < MIX_SPARE) spare[tid] = ul != nullptr ? ul[B + tid] : hv_rng_uniform(rng_state, MIX_SITE + (unsigned)lvl, (unsigned long long)(B + tid));
    int n2 = 1024;
    while (n2 < B) n2 <<= 1;
    float *key = reinterpret_cast<float *>(mix_lds);
    unsigned short *idx = reinterpret_cast<unsigned short *>(mix_lds + 4 * (size_t)n2);
    int64_t *inv = inverse + (int64_t)lvl * B;  // (written in place: -1 everywhere first, the valid rows' entries after the barrier below)
    for (int i = tid; i < n2; i += 1024) {
        const bool valid = i < B && targets[(int64_t)i * ldt + lvl] >= 0;
        const float ui = !valid ? 0.0f : (ul != nullptr ? ul[i] : hv_rng_uniform(rng_state, MIX_SITE + (unsigned)lvl, (unsigned long long)i));
        key[i] = valid ? ui : (i < B ? 2.0f : 3.0f);  // valid rows first, then the invalid ones, then the padding
        idx[i] = (unsigned short)i;
        if (i < B) inv[i] = -1;
    }
    __syncthreads();
    // (a stage of distance j < 64 exchanges inside 64-element blocks, and a block is one wave's in every pass of the i loop: between two
    //  such stages the wave's own program order through LDS is all the ordering there is to keep -- 14 workgroup barriers at 1024 rows
    //  instead of 55)
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += 1024) {
                const int p = i ^ j;
                if (p > i) {
                    const bool up = (i & k) == 0;
                    const float ka = key[i], kb = key[p];
                    const int ia = idx[i], ib = idx[p];
                    const bool gt = ka > kb || (ka == kb && ia > ib);
                    if (gt == up) { key[i] = kb; key[p] = ka; idx[i] = ib; idx[p] = ia; }
                }
            }
            const int nj = j > 1 ? j >> 1 : k;  // the following stage's distance (k: the first stage of the next k; past the last: >= 64)
            if (j >= 64 || nj >= 64) {
                __syncthreads();
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    // rank of each valid row among the valid rows: `per` consecutive rows per thread, scan of the per-thread counts
    const int per = n2 / 1024;
    int cnt = 0;
    for (int r = 0; r < per; r++) {
        const int i = tid * per + r;
        cnt += (i < B && targets[(int64_t)i * ldt + lvl] >= 0) ? 1 : 0;
    }
    // exclusive scan of the per-thread counts: inside the wave by shuffles, the sixteen wave totals through LDS
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if ((tid & 63) >= off) incl += v;
    }
    if ((tid & 63) == 63) part[tid >> 6] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < (tid >> 6); w++) before += part[w];
    int rank = before + incl - cnt;
    for (int r = 0; r < per; r++) {
        const int i = tid * per + r;
        if (i >= B) break;
        const bool valid = targets[(int64_t)i * ldt + lvl] >= 0;
        const int pr = valid ? (int)idx[rank] : -1;
        partner[(int64_t)lvl * B + i] = pr;
        if (valid) { inv[pr] = i; rank++; }
    }
    if (tid == 0) {
        int k = 0;
        const float x = mix_gamma(alpha, spare, k), y = mix_gamma(alpha, spare, k);
        const float s = x + y;
        lam[lvl] = s > 0.0f ? x / s : 0.5f;
    }
}
}  // namespace

extern "C" int hidvae_mixup_plan(const int64_t *targets, int64_t B, int L, int64_t ld_targets, const float *uniforms, float alpha,
                                 int64_t *partner, int64_t *inverse, float *lam, const unsigned long long *rng_state, void *stream) {
    HV_REQUIRE(targets && (uniforms || rng_state) && partner && inverse && lam && B >= 1 && L >= 1 && ld_targets >= L && alpha > 0.0f,
               "mixup_plan: bad arguments");
    HV_REQUIRE(B <= MIX_MAXB, "mixup_plan: B=%lld rows do not fit the in-LDS sort (max %d)", (long long)B, MIX_MAXB);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&mixup_plan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       6 * MIX_MAXB);
    HV_REQUIRE(attr == hipSuccess, "mixup_plan: could not size the LDS of mixup_plan_kernel");
    size_t n2 = 1024;
    while ((int64_t)n2 < B) n2 <<= 1;
    hipLaunchKernelGGL(mixup_plan_kernel, dim3((unsigned)L), dim3(1024), 6 * n2, (hipStream_t)stream, targets, B, ld_targets, uniforms, alpha,
                       partner, inverse, lam, rng_state);
    HV_LAUNCH_CHECK("mixup_plan");
    return HIDVAE_OK;
}

namespace {
__global__ void rng_advance_kernel(unsigned long long *state) { state[1] = state[1] + 1ull; }
__global__ __launch_bounds__(256) void dropout_mask_kernel(float *out, int64_t n, HvDrop d) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = hv_drop_keep(d, (unsigned long long)i) ? 1.0f : 0.0f;
}
}  // namespace

extern "C" int hidvae_rng_advance(unsigned long long *state, void *stream) {
    HV_REQUIRE(state != nullptr, "rng_advance: null state");
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
    HV_LAUNCH_CHECK("rng_advance");
    return HIDVAE_OK;
}

extern "C" int hidvae_dropout_mask(float *out, int64_t numel, const unsigned long long *rng_state, unsigned site, unsigned threshold,
                                   void *stream) {
    HV_REQUIRE(out && rng_state && numel >= 1, "dropout_mask: bad arguments");
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(ew_grid(numel)), dim3(256), 0, (hipStream_t)stream, out, numel, HvDrop{rng_state, site, threshold});
    HV_LAUNCH_CHECK("dropout_mask");
    return HIDVAE_OK;
}

extern "C" int hidvae_tag_loss_fwd(const float *logits, int64_t B, int64_t C, const int64_t *target, const int64_t *partner,
                                   const float *lam_dev,
                                   int focal, float gamma, float alpha, float smooth, float ce_label_smoothing, float *loss,
                                   float *acc, float *n_valid, float *row_loss, float *row_hit, float *zbuf, float *dmix, float *dkl,
                                   void *stream) {
    HV_REQUIRE(logits && target && loss && acc && n_valid && row_loss && row_hit && zbuf && B >= 1 && C >= 1, "tag_loss_fwd: bad arguments");
    HV_REQUIRE(partner == nullptr || lam_dev != nullptr, "tag_loss_fwd: mixup needs lam");
    TagLossArgs a{logits, B, C, target, partner, lam_dev, focal, gamma, alpha, smooth, ce_label_smoothing, row_loss, row_hit, dmix, dkl};
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)hv_cdiv(B, 4)), block(256);
    if (C <= 64) hipLaunchKernelGGL(tag_loss_rows_reg_kernel<1>, grid, block, 0, s, a);
    else if (C <= 128) hipLaunchKernelGGL(tag_loss_rows_reg_kernel<2>, grid, block, 0, s, a);
    else if (C <= 192) hipLaunchKernelGGL(tag_loss_rows_reg_kernel<3>, grid, block, 0, s, a);
    else if (C <= 256) hipLaunchKernelGGL(tag_loss_rows_reg_kernel<4>, grid, block, 0, s, a);
    else if (C <= 384) hipLaunchKernelGGL(tag_loss_rows_reg_kernel<6>, grid, block, 0, s, a);
    else if (C <= 512) hipLaunchKernelGGL(tag_loss_rows_reg_kernel<8>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(tag_loss_rows_kernel, grid, block, 0, s, a, zbuf);  // wider rows: through the scratch rows
    HV_LAUNCH_CHECK("tag_loss_rows");
    hipLaunchKernelGGL(tag_loss_reduce_kernel, dim3(1), dim3(256), 0, s, row_loss, row_hit, target, B, loss, acc, n_valid);
    HV_LAUNCH_CHECK("tag_loss_reduce");
    return HIDVAE_OK;
}

extern "C" int hidvae_tag_loss_bwd(const float *dmix, const float *dkl, const int64_t *target, const int64_t *inverse,
                                   const float *lam_dev,
                                   int64_t B, int64_t C, const float *g_dev, const float *n_valid, float *g_logits, void *stream) {
    HV_REQUIRE(dmix && target && g_dev && n_valid && g_logits && B >= 1 && C >= 1, "tag_loss_bwd: bad arguments");
    hipLaunchKernelGGL(tag_loss_bwd_kernel, dim3(ew_grid(B * C)), dim3(256), 0, (hipStream_t)stream, dmix, dkl, target, inverse, lam_dev, B,
                       C, g_dev, n_valid, g_logits);
    HV_LAUNCH_CHECK("tag_loss_bwd");
    return HIDVAE_OK;
}
