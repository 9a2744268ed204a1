// Lloyd k-means for the first-batch codebook initialisation (reference init/kmeans.py:34-77, called from
// modules/quantize.py:91-95,103-104) on gfx950; kernels instantiated for embed_dim 32 and 64 (narrower multiples of 4: zero-padded reads).
//   assign : nearest centroid by the DIRECT sum_d (x_d - c_d)^2 (the reference's formula), first minimum;
//   update : centroid_k = (sum of its items in ascending item order) / count -- fixed order, bit-reproducible;
//            an empty cluster is re-seeded from x[reseed_idx[k]] (the reference draws torch.randint);
//   shift  : max_k |c_new - c_old|_2, what the reference compares with its 1e-10 stop threshold.
#include <math.h>
#include "common.h"

namespace {

constexpr int KT = 128;  // centroids staged in LDS per tile

// DP: the padded width the kernel is compiled for (32 or 64); D: the rows' real width (a multiple of 4, <= DP): the padding
// components are zeros on both sides, so they add exact zeros at the end of the ascending-d sum
template <int DP>
__global__ __launch_bounds__(256) void kmeans_assign_kernel(const float *x, int64_t N, const float *c, int64_t K, int32_t *assign, int D) {
    __shared__ float cs[KT][DP + 1];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float xv[DP];
    const int64_t src = i < N ? i : N - 1;
#pragma unroll
    for (int d4 = 0; d4 < DP / 4; d4++) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (4 * d4 < D) v = *reinterpret_cast<const float4 *>(x + src * D + 4 * d4);
        xv[4 * d4] = v.x; xv[4 * d4 + 1] = v.y; xv[4 * d4 + 2] = v.z; xv[4 * d4 + 3] = v.w;
    }
    float best = INFINITY;
    int bi = 0;
    for (int64_t k0 = 0; k0 < K; k0 += KT) {
        __syncthreads();
        for (int e = threadIdx.x; e < KT * DP; e += 256) {
            const int kk = e / DP, d = e - kk * DP;
            cs[kk][d] = (k0 + kk < K && d < D) ? c[(k0 + kk) * D + d] : 0.0f;
        }
        __syncthreads();
        const int lim = (int)((K - k0) < KT ? (K - k0) : KT);
        for (int kk = 0; kk < lim; kk++) {
            float s = 0.0f;
#pragma unroll
            for (int d = 0; d < DP; d++) { const float t = xv[d] - cs[kk][d]; s += t * t; }
            if (s < best) { best = s; bi = (int)(k0 + kk); }
        }
    }
    if (i < N) assign[i] = bi;
}

// one wave per cluster; same scan-then-accumulate structure as the codebook gradient (rq.hip)
// D <= 32: the two half-waves hold the same component (d = lane & 31) and the squared shift is summed over 32 lanes, as ever;
// wider rows: a lane per component, summed over the wave (lanes past D contribute zeros)
__global__ __launch_bounds__(256) void kmeans_update_kernel(const float *x, int64_t N, const int32_t *assign, const float *c_old,
                                                            int64_t K, const int64_t *reseed_idx, float *c_new, float *shift_k, int D) {
    __shared__ int hits[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t k = (int64_t)blockIdx.x * 4 + wave;
    if (k >= K) return;
    const bool wide = D > 32;
    const int d = wide ? lane : (lane & 31);
    const bool on = d < D;
    int *list = hits[wave];
    float acc = 0.0f;
    int64_t count = 0;
    for (int64_t c0 = 0; c0 < N; c0 += 1024) {
        int n = 0;
#pragma unroll 4
        for (int j = 0; j < 16; j++) {
            const int64_t b = c0 + j * 64 + lane;
            const bool hit = b < N && assign[b] == (int32_t)k;
            const unsigned long long m = __ballot(hit);
            if (hit) list[n + __popcll(m & ((1ull << lane) - 1ull))] = (int)(j * 64 + lane);
            n += __popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        if (on)
            for (int e = 0; e < n; e++) acc += x[(c0 + list[e]) * D + d];
        count += n;
        __builtin_amdgcn_wave_barrier();
    }
    float v = 0.0f;
    if (on) v = count > 0 ? acc / (float)count : x[reseed_idx[k] * D + d];
    const float df = on ? v - c_old[k * D + d] : 0.0f;
    float s = df * df;
    if (wide) s = s + __shfl_xor(s, 32);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (on && (wide || lane < 32)) c_new[k * D + d] = v;
    if (lane == 0) shift_k[k] = sqrtf(s);
}

__global__ __launch_bounds__(256) void kmeans_shift_kernel(const float *shift_k, int64_t K, float *shift) {
    __shared__ float red[4];
    float m = 0.0f;
    for (int64_t i = threadIdx.x; i < K; i += 256) m = fmaxf(m, shift_k[i]);
    m = hv_wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) *shift = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

}  // namespace

extern "C" int hidvae_kmeans_iter(const float *x, int64_t N, const float *centroids, int64_t K, int32_t *assign,
                                  const int64_t *reseed_idx, float *new_centroids, float *shift_scratch, float *shift, int embed_dim, void *stream) {
    HV_REQUIRE(x && centroids && assign && reseed_idx && new_centroids && shift_scratch && shift && N >= 1 && K >= 1,
               "kmeans_iter: bad arguments");
    HV_REQUIRE(embed_dim >= 4 && embed_dim <= 64 && embed_dim % 4 == 0, "kmeans_iter: embed_dim=%d (a multiple of 4, at most 64)", embed_dim);
    hipStream_t s = (hipStream_t)stream;
    if (embed_dim <= 32) hipLaunchKernelGGL(kmeans_assign_kernel<32>, dim3((unsigned)hv_cdiv(N, 256)), dim3(256), 0, s, x, N, centroids, K, assign, embed_dim);
    else hipLaunchKernelGGL(kmeans_assign_kernel<64>, dim3((unsigned)hv_cdiv(N, 256)), dim3(256), 0, s, x, N, centroids, K, assign, embed_dim);
    HV_LAUNCH_CHECK("kmeans_assign");
    hipLaunchKernelGGL(kmeans_update_kernel, dim3((unsigned)hv_cdiv(K, 4)), dim3(256), 0, s, x, N, assign, centroids, K, reseed_idx,
                       new_centroids, shift_scratch, embed_dim);
    HV_LAUNCH_CHECK("kmeans_update");
    hipLaunchKernelGGL(kmeans_shift_kernel, dim3(1), dim3(256), 0, s, shift_scratch, K, shift);
    HV_LAUNCH_CHECK("kmeans_shift");
    return HIDVAE_OK;
}
