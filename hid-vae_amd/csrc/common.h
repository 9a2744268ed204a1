// Shared host/device helpers for the HiD-VAE gfx950 kernels (wave = 64 lanes, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hidvae.h"

int hv_fail(int code, const char *fmt, ...);  // records the thread's last error, returns `code`

#define HV_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return hv_fail(HIDVAE_EINVAL, __VA_ARGS__); \
    } while (0)

#define HV_LAUNCH_CHECK(name)                                                                     \
    do {                                                                                          \
        hipError_t e__ = hipGetLastError();                                                       \
        if (e__ != hipSuccess) return hv_fail(HIDVAE_ELAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

static inline int64_t hv_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define HV_WAVE 64

// ---- device math with a FIXED operation order (the files are built with -ffp-contract=off, so only
// explicit fmaf fuses).  expE / silu mirror oracle/exact.c's specification bit for bit.
__device__ __forceinline__ float hv_expE(float x) {
    x = fminf(fmaxf(x, -87.3f), 88.7f);
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.428606765330187e-06f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p, r * r, r) + 1.0f;
    return ldexpf(y, (int)n);
}
__device__ __forceinline__ float hv_sigmoid(float a) { return 1.0f / (1.0f + hv_expE(-a)); }
__device__ __forceinline__ float hv_silu(float a) { return a / (1.0f + hv_expE(-a)); }
__device__ __forceinline__ float hv_dsilu(float a) {
    float s = hv_sigmoid(a);
    return s * (1.0f + a * (1.0f - s));
}
__device__ __forceinline__ float hv_gelu(float a) { return 0.5f * a * (1.0f + erff(a * 0.70710678118654752f)); }
__device__ __forceinline__ float hv_dgelu(float a) {
    float cdf = 0.5f * (1.0f + erff(a * 0.70710678118654752f));
    float pdf = 0.3989422804014327f * __expf(-0.5f * a * a);
    return cdf + a * pdf;
}

// GEMM epilogues (include/hidvae.h HIDVAE_EPI_*).  `scale` matters to DRELU only: the backward through ReLU -> Dropout(keep_scale) read off
// the layer's OUTPUT y = relu(.) * keep * scale (y > 0 exactly where the unit was active AND kept, so neither the keep-mask nor the
// pre-activation is needed)
__device__ __forceinline__ float hv_apply_epilogue(int epi, float v, const float *aux, int64_t off, float scale = 1.0f) {
    switch (epi) {
        case HIDVAE_EPI_SILU: return hv_silu(v);
        case HIDVAE_EPI_RELU: return fmaxf(v, 0.0f);
        case HIDVAE_EPI_GELU: return hv_gelu(v);
        case HIDVAE_EPI_SIGMOID: return hv_sigmoid(v);
        case HIDVAE_EPI_DSILU: return v * hv_dsilu(aux[off]);
        case HIDVAE_EPI_DRELU: return aux[off] > 0.0f ? v * scale : 0.0f;
        case HIDVAE_EPI_DGELU: return v * hv_dgelu(aux[off]);
        case HIDVAE_EPI_DSIGMOID: { const float s = aux[off]; return v * (s * (1.0f - s)); }
        default: return v;
    }
}

// per-step AdamW scalars in double precision, once per tensor (torch computes them on the host in double), and the step counter
// itself.  Executed by ONE whole workgroup (adamw_prepare_kernel, or a spare workgroup of a launch that carries it).
struct HvAdamPrepare {
    int64_t *step;
    const float *base_lr, *wd;
    int n;
    float beta1, beta2, eta_min;
    int64_t T_max, step_size;
    float gamma;
    float *hyper;
};
__device__ __forceinline__ void hv_adamw_prepare(const HvAdamPrepare &a) {
    const int64_t t1 = a.step[0] + 1;  // torch counts the step being taken from 1
    const int64_t ts = t1 - 1 + a.step[1];  // scheduler steps taken so far: the optimizer's own + those of a run resumed without its state
    for (int t = threadIdx.x; t < a.n; t += blockDim.x) {
        double lr = (double)a.base_lr[t];
        if (a.T_max > 0)  // CosineAnnealingLR after (t1 - 1) scheduler steps, closed form
            lr = (double)a.eta_min + (lr - (double)a.eta_min) * (1.0 + cos(M_PI * (double)ts / (double)a.T_max)) * 0.5;
        else if (a.step_size > 0)  // StepLR after (t1 - 1) scheduler steps: base * gamma^floor((t1-1)/step_size)
            lr = lr * pow((double)a.gamma, (double)(ts / a.step_size));
        const double bc1 = 1.0 - pow((double)a.beta1, (double)t1);
        const double bc2 = 1.0 - pow((double)a.beta2, (double)t1);
        a.hyper[3 * t + 0] = (float)(1.0 - lr * (double)a.wd[t]);
        a.hyper[3 * t + 1] = (float)(lr / bc1);
        a.hyper[3 * t + 2] = (float)sqrt(bc2);
    }
    __syncthreads();
    if (threadIdx.x == 0) a.step[0] = t1;
}

// ---- counter-based randomness for dropout (and the mixup plan): Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as
// 1, 2, 3", SC'11; the generator torch's CUDA dropout uses too).  key = the provider's 64-bit seed, counter = (element index lo, hi,
// site, step): `site` numbers the dropout layers of a forward pass in the order the reference reaches them (h_rqvae.py:147-186,327),
// `step` is a device counter advanced once per forward pass -- so a replayed HIP graph draws fresh masks every step, every site and
// element gets an independent stream, and nothing but the 16-byte state lives in memory: the keep-decision is evaluated where the
// activation is produced, and the backward never needs it (it reads the gate off the forward OUTPUT, y > 0).
struct HvDrop {
    const unsigned long long *state;  // [0] seed, [1] step; nullptr: no in-kernel dropout
    unsigned site;
    unsigned threshold;               // drop iff word < threshold, threshold = round(p_drop * 2^32)
};
__device__ __forceinline__ unsigned hv_philox4x32_10_word0(unsigned long long key, unsigned c0, unsigned c1, unsigned c2, unsigned c3) {
    unsigned k0 = (unsigned)key, k1 = (unsigned)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}
__device__ __forceinline__ unsigned hv_rng_word(const unsigned long long *state, unsigned site, unsigned long long idx) {
    return hv_philox4x32_10_word0(state[0], (unsigned)idx, (unsigned)(idx >> 32), site, (unsigned)state[1]);
}
__device__ __forceinline__ bool hv_drop_keep(const HvDrop &d, unsigned long long idx) { return hv_rng_word(d.state, d.site, idx) >= d.threshold; }
__device__ __forceinline__ float hv_rng_uniform(const unsigned long long *state, unsigned site, unsigned long long idx) {  // [0, 1)
    return (float)(hv_rng_word(state, site, idx) >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float hv_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float hv_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---- residual quantisation at embedding widths other than 32 (rq_generic.hip), called by the entry points of rq.hip ------------------
bool hv_rqg_dim_ok(int D);
int hv_rqg_prepare(const float *const *E_host, const int32_t *normalize_host, int L, int64_t K, int D, float *cb_eff, float *cc, hipStream_t s);
int hv_rqg_forward(const float *y, int64_t B, int normalize_input, const float *cb_eff, const float *cc, int L, int64_t K, int D, int mode,
                   int training, float beta, float *z, int64_t *ids, float *emb_cat, int64_t ld_cat, float *emb_sum, float *res_cat,
                   float *qloss, int cosine, hipStream_t s);
int hv_rqg_backward(const float *y, const float *z, int64_t B, int normalize_input, const float *cb_eff, const float *cc, int L, int64_t K, int D,
                    int mode, float beta, const int64_t *ids, const float *g_cat, int64_t ld_gcat, const float *g_sum, const float *g_z_in,
                    int64_t g_z_rows, float gq, const float *gq_items, int64_t gq_stride, float *g_y, float *dE_rows, hipStream_t s);
int hv_rqg_codebook_grad(const int64_t *ids, const float *dE_rows, int64_t B, int L, int64_t K, int D, const float *const *E_host,
                         const float *cb_eff, const int32_t *normalize_host, float *const *gE_host, int accumulate, hipStream_t s);
