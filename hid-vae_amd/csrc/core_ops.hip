// Core-path row kernels, optimizer and id statistics for the HiD-VAE tokenizer step (gfx950).
//   recon_fwd_bwd : decoder tail  l2norm -> sum (x_hat - x)^2 (+ gradient)   encoder.py:32, loss.py:11-12
//   adamw_step    : multi-tensor AdamW with an on-device cosine schedule     train_hidvae.py:533-563,636-640
//   id_stats      : embs_norm and p_unique_ids                               h_rqvae.py:643-648
#include <math.h>
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// one wave per row; rows are short (768 floats), so a row is read once into registers (float4 x MAXV)
// ---------------------------------------------------------------------------------------------------
constexpr int MAXV = 4;  // N <= 64 lanes * 4 floats * MAXV = 1024 on the vector path

// the decoder tail with n_cat categorical columns at the end of the row (h_rqvae.py:610-613, rqvae.py:146-149, loss.py:15-33):
//   u = y / max(|y|, eps)            the decoder's closing L2NormalizationLayer, over the WHOLE row (encoder.py:32)
//   v = u[:H] / max(|u[:H]|, eps)    the forward's l2norm of the non-categorical head, H = N - n_cat
//   recon = sum_{i<H} (v_i - x_i)^2 + sum_{i>=H} BCE-with-logits(u_i, x_i)
// and its gradient through both normalisations.  Not on any shipped config's path (all bind n_cat = 0): one wave per row, scalar loads.
__device__ __noinline__ void recon_cat_row(const float *y, const float *x, int64_t row, int64_t N, int n_cat, float gscale, float *x_hat,
                                           float *recon, float *g_y) {
    const int lane = threadIdx.x & 63;
    const float *yr = y + row * N, *xr = x + row * N;
    const int64_t H = N - n_cat;
    float ss = 0.0f;
    for (int64_t i = lane; i < N; i += 64) ss += yr[i] * yr[i];
    const float n1 = sqrtf(hv_wave_sum(ss)), den1 = fmaxf(n1, 1e-12f);
    float sh = 0.0f;
    for (int64_t i = lane; i < H; i += 64) { const float u = yr[i] / den1; sh += u * u; }
    const float n2 = sqrtf(hv_wave_sum(sh)), den2 = fmaxf(n2, 1e-12f);
    float rs = 0.0f, t2 = 0.0f;
    for (int64_t i = lane; i < N; i += 64) {
        const float u = yr[i] / den1;
        if (i < H) {
            const float v = u / den2, d = v - xr[i];
            rs += d * d;
            t2 += v * d;
            if (x_hat != nullptr) x_hat[row * N + i] = v;
        } else {  // max(u, 0) - u x + log(1 + exp(-|u|))
            rs += (fmaxf(u, 0.0f) - u * xr[i]) + log1pf(expf(-fabsf(u)));
            if (x_hat != nullptr) x_hat[row * N + i] = u;
        }
    }
    rs = hv_wave_sum(rs);
    if (lane == 0 && recon != nullptr) recon[row] = rs;
    if (g_y == nullptr) return;
    t2 = n2 > 1e-12f ? hv_wave_sum(t2) : 0.0f;
    // g_u: head 2 (d - v (v.d)) / den2, tail sigmoid(u) - x;  g_y = gscale (g_u - u (u.g_u)) / den1
    float t1 = 0.0f;
    for (int64_t i = lane; i < N; i += 64) {
        const float u = yr[i] / den1;
        float gu;
        if (i < H) { const float v = u / den2; gu = 2.0f * ((v - xr[i]) - v * t2) / den2; }
        else gu = 1.0f / (1.0f + expf(-u)) - xr[i];
        t1 += u * gu;
    }
    t1 = n1 > 1e-12f ? hv_wave_sum(t1) : 0.0f;
    for (int64_t i = lane; i < N; i += 64) {
        const float u = yr[i] / den1;
        float gu;
        if (i < H) { const float v = u / den2; gu = 2.0f * ((v - xr[i]) - v * t2) / den2; }
        else gu = 1.0f / (1.0f + expf(-u)) - xr[i];
        g_y[row * N + i] = gscale * (gu - u * t1) / den1;
    }
}

// one wave, one row: x_hat = y/max(|y|,eps), recon = |x_hat - x|^2, g_y = gscale * d recon / d y   (each output optional)
template <bool VEC>
__device__ __forceinline__ void recon_row(const float *y, const float *x, int64_t row, int64_t N, int n_cat, float gscale, float *x_hat,
                                          float *recon, float *g_y) {
    if (n_cat > 0) return recon_cat_row(y, x, row, N, n_cat, gscale, x_hat, recon, g_y);
    const int lane = threadIdx.x & 63;
    const float *yr = y + row * N, *xr = x + row * N;
    if (VEC) {
        const int nv = (int)(N / 4);
        float4 yv[MAXV], xv[MAXV];
        float ss = 0.0f;
#pragma unroll
        for (int s = 0; s < MAXV; s++) {
            const int i = lane + 64 * s;
            yv[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            xv[s] = yv[s];
            if (i < nv) {
                yv[s] = reinterpret_cast<const float4 *>(yr)[i];
                xv[s] = reinterpret_cast<const float4 *>(xr)[i];
            }
            ss += (yv[s].x * yv[s].x + yv[s].y * yv[s].y) + (yv[s].z * yv[s].z + yv[s].w * yv[s].w);
        }
        ss = hv_wave_sum(ss);
        const float nrm = sqrtf(ss), den = fmaxf(nrm, 1e-12f);
        float rs = 0.0f, t = 0.0f;
        float4 dv[MAXV];
#pragma unroll
        for (int s = 0; s < MAXV; s++) {
            const float4 h = make_float4(yv[s].x / den, yv[s].y / den, yv[s].z / den, yv[s].w / den);
            dv[s] = make_float4(h.x - xv[s].x, h.y - xv[s].y, h.z - xv[s].z, h.w - xv[s].w);
            rs += (dv[s].x * dv[s].x + dv[s].y * dv[s].y) + (dv[s].z * dv[s].z + dv[s].w * dv[s].w);
            t += (h.x * dv[s].x + h.y * dv[s].y) + (h.z * dv[s].z + h.w * dv[s].w);
            yv[s] = h;
            const int i = lane + 64 * s;
            if (x_hat != nullptr && i < nv) reinterpret_cast<float4 *>(x_hat + row * N)[i] = h;
        }
        rs = hv_wave_sum(rs);
        t = hv_wave_sum(t);
        if (lane == 0 && recon != nullptr) recon[row] = rs;
        if (g_y != nullptr) {
            // g_xhat = 2 gscale (x_hat - x);  g_y = (g_xhat - x_hat (x_hat . g_xhat)) / den   (den = |y| unless clamped)
            const float c = 2.0f * gscale / den;
            const float proj = nrm > 1e-12f ? t : 0.0f;
#pragma unroll
            for (int s = 0; s < MAXV; s++) {
                const int i = lane + 64 * s;
                if (i < nv)
                    reinterpret_cast<float4 *>(g_y + row * N)[i] =
                        make_float4(c * (dv[s].x - yv[s].x * proj), c * (dv[s].y - yv[s].y * proj),
                                    c * (dv[s].z - yv[s].z * proj), c * (dv[s].w - yv[s].w * proj));
            }
        }
    } else {
        float ss = 0.0f;
        for (int64_t i = lane; i < N; i += 64) ss += yr[i] * yr[i];
        ss = hv_wave_sum(ss);
        const float nrm = sqrtf(ss), den = fmaxf(nrm, 1e-12f);
        float rs = 0.0f, t = 0.0f;
        for (int64_t i = lane; i < N; i += 64) {
            const float h = yr[i] / den, d = h - xr[i];
            rs += d * d;
            t += h * d;
            if (x_hat != nullptr) x_hat[row * N + i] = h;
        }
        rs = hv_wave_sum(rs);
        t = hv_wave_sum(t);
        if (lane == 0 && recon != nullptr) recon[row] = rs;
        if (g_y != nullptr) {
            const float c = 2.0f * gscale / den, proj = nrm > 1e-12f ? t : 0.0f;
            for (int64_t i = lane; i < N; i += 64) {
                const float h = yr[i] / den;
                g_y[row * N + i] = c * ((h - xr[i]) - h * proj);
            }
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void recon_kernel(const float *y, const float *x, int64_t B, int64_t N, int n_cat, float gscale_all,
                                                    const float *gscale_items, int64_t gs_stride, float *x_hat, float *recon, float *g_y) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float gscale = gscale_items != nullptr ? gscale_all * gscale_items[row * gs_stride] : gscale_all;
    recon_row<VEC>(y, x, row, N, n_cat, gscale, x_hat, recon, g_y);
}

// generic row L2 normalise: out = x / max(|x|, eps); saves |x| for the backward.  One wave per row.
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float *x, int64_t M, int64_t N, int64_t ldx, float eps,
                                                         float *out, int64_t ldo, float *norms) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float ss = 0.0f;
    for (int64_t i = lane; i < N; i += 64) ss += x[row * ldx + i] * x[row * ldx + i];
    ss = hv_wave_sum(ss);
    const float nrm = sqrtf(ss), den = fmaxf(nrm, eps);
    for (int64_t i = lane; i < N; i += 64) out[row * ldo + i] = x[row * ldx + i] / den;
    if (lane == 0 && norms != nullptr) norms[row] = nrm;
}

// gx (+)= (g - out (out . g)) / |x|   (or g / eps where the norm was clamped)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float *g, int64_t ldg, const float *out, int64_t ldo,
                                                         const float *norms, int64_t M, int64_t N, float eps, float *gx,
                                                         int64_t ldgx, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float t = 0.0f;
    for (int64_t i = lane; i < N; i += 64) t += out[row * ldo + i] * g[row * ldg + i];
    t = hv_wave_sum(t);
    const float nrm = norms[row];
    const float den = fmaxf(nrm, eps), proj = nrm > eps ? t : 0.0f;
    for (int64_t i = lane; i < N; i += 64) {
        const float v = (g[row * ldg + i] - out[row * ldo + i] * proj) / den;
        float *d = gx + row * ldgx + i;
        *d = accumulate ? *d + v : v;
    }
}

// two independent row normalisations of M rows each in ONE launch (InfoNCE normalises the codebook embeddings and the projected tags,
// loss.py:66-67; their backward likewise): problem 0 on the first ceil(M/4) workgroups, problem 1 on the rest; per row exactly the
// arithmetic of l2norm_fwd_kernel / l2norm_bwd_kernel
struct L2Pair {
    const float *a[2]; int64_t lda[2];   // fwd: x;  bwd: g
    const float *b[2]; int64_t ldb[2];   // bwd: out (the normalised rows)
    const float *norms_in[2];            // bwd
    float *out[2]; int64_t ldo[2];       // fwd: out;  bwd: gx
    float *norms_out[2];                 // fwd
    int64_t N[2];
    int64_t M;
    float eps;
};
__global__ __launch_bounds__(256) void l2norm_fwd_pair_kernel(L2Pair p) {
    const int lane = threadIdx.x & 63;
    const int64_t per = (p.M + 3) / 4;
    const int k = (int64_t)blockIdx.x >= per ? 1 : 0;
    const int64_t row = ((int64_t)blockIdx.x - k * per) * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const float *x = p.a[k] + row * p.lda[k];
    const int64_t N = p.N[k];
    float ss = 0.0f;
    for (int64_t i = lane; i < N; i += 64) ss += x[i] * x[i];
    ss = hv_wave_sum(ss);
    const float nrm = sqrtf(ss), den = fmaxf(nrm, p.eps);
    for (int64_t i = lane; i < N; i += 64) p.out[k][row * p.ldo[k] + i] = x[i] / den;
    if (lane == 0) p.norms_out[k][row] = nrm;
}
__global__ __launch_bounds__(256) void l2norm_bwd_pair_kernel(L2Pair p) {
    const int lane = threadIdx.x & 63;
    const int64_t per = (p.M + 3) / 4;
    const int k = (int64_t)blockIdx.x >= per ? 1 : 0;
    const int64_t row = ((int64_t)blockIdx.x - k * per) * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const float *g = p.a[k] + row * p.lda[k], *o = p.b[k] + row * p.ldb[k];
    const int64_t N = p.N[k];
    float t = 0.0f;
    for (int64_t i = lane; i < N; i += 64) t += o[i] * g[i];
    t = hv_wave_sum(t);
    const float nrm = p.norms_in[k][row];
    const float den = fmaxf(nrm, p.eps), proj = nrm > p.eps ? t : 0.0f;
    for (int64_t i = lane; i < N; i += 64) p.out[k][row * p.ldo[k] + i] = (g[i] - o[i] * proj) / den;
}

// ---------------------------------------------------------------------------------------------------
// AdamW.  blockIdx.y = tensor, blockIdx.x strides over its elements.
// ---------------------------------------------------------------------------------------------------
constexpr int ADAM_CHUNK = 128;  // gradient pointers travel BY VALUE in the kernel arguments (they change every step
                                  // under autograd; parameters / moments / hyper-parameters are fixed device tables)
struct AdamArgs {
    float *const *p;
    const float *g[ADAM_CHUNK];
    int first;  // index of g[0] in the device tables
    float *const *m;
    float *const *v;
    const int64_t *numel;
    const float *hyper;  // [n][3] = {1 - lr*wd, lr / bias_correction1, sqrt(bias_correction2)} from adamw_prepare_kernel
    float beta1, beta2, eps, grad_scale;
};

// one tiny launch (or a spare workgroup of hidvae_codebook_prepare_adamw, which saves the launch): see hv_adamw_prepare
__global__ void adamw_prepare_kernel(HvAdamPrepare a) { hv_adamw_prepare(a); }

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
    const int t = a.first + blockIdx.y;
    const int64_t n = a.numel[t];
    const float decay = a.hyper[3 * t], step_size = a.hyper[3 * t + 1], bc2_sqrt = a.hyper[3 * t + 2];
    float *p = a.p[t], *m = a.m[t], *v = a.v[t];
    const float *g = a.g[blockIdx.y];
    const float w1 = 1.0f - a.beta1, w2 = 1.0f - a.beta2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 4 * stride) {  // 16 independent loads in flight
        float gv[4], pv[4], mv[4], vv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int64_t i = i0 + j * stride;
            const bool ok = i < n;
            gv[j] = ok ? g[i] : 0.0f; pv[j] = ok ? p[i] : 0.0f; mv[j] = ok ? m[i] : 0.0f; vv[j] = ok ? v[i] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int64_t i = i0 + j * stride;
            if (i >= n) break;
            const float gi = gv[j] * a.grad_scale;
            float pi = pv[j] * decay;
            const float mi = mv[j] + (gi - mv[j]) * w1;  // lerp
            const float vi = vv[j] * a.beta2 + (gi * gi) * w2;
            const float denom = sqrtf(vi) / bc2_sqrt + a.eps;
            pi = pi - step_size * (mi / denom);
            p[i] = pi;
            m[i] = mi;
            v[i] = vi;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// id statistics
// ---------------------------------------------------------------------------------------------------
// One launch: embs_norm, and a census of the distinct id tuples in an open-addressing table that is never cleared.
// scratch = tsize slots + {distinct<<32 | tickets, unused, call counter}; zero-filled ONCE by its owner.  A slot holds
// (generation << 40) | item; slots of older generations count as empty, so the table needs no per-call initialisation.  The
// workgroup that takes the last ticket publishes p_unique, resets the counters and moves to the next generation (and wipes
// the table when the 24-bit generation wraps).
constexpr unsigned long long GEN_MOD = 0xFFFFFEull;
__global__ __launch_bounds__(256) void id_stats_kernel(const float *emb_cat, int64_t ld_cat, const int64_t *ids, int64_t B, int L,
                                                       float *embs_norm, unsigned long long *table, int64_t tsize, float *p_unique, int D) {
    __shared__ int s_new;
    __shared__ int s_last;
    unsigned long long *ctrl = table + tsize;
    const unsigned long long calls = ctrl[2];
    const unsigned long long gen = calls % GEN_MOD + 1ull;
    if (threadIdx.x == 0) s_new = 0;
    __syncthreads();
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (embs_norm != nullptr && idx < B * L) {
        const int64_t b = idx / L;
        const int i = (int)(idx - b * L);
        const float4 *p = reinterpret_cast<const float4 *>(emb_cat + b * ld_cat + i * D);
        float s = 0.0f;
        for (int j = 0; j < D / 4; j++) {  // (D = 32: eight float4, the order the fused middle launch reproduces)
            const float4 v = p[j];
            s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        embs_norm[idx] = sqrtf(s);
    }
    if (idx < B) {
        const int64_t b = idx;
        unsigned long long h = 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < L; i++) {
            h ^= (unsigned long long)ids[b * L + i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
            h *= 0xBF58476D1CE4E5B9ull;
            h ^= h >> 29;
        }
        const unsigned long long mine = (gen << 40) | (unsigned long long)b;
        int64_t slot = (int64_t)(h % (unsigned long long)tsize);
        for (int64_t probe = 0; probe < tsize;) {
            unsigned long long cur = __atomic_load_n(table + slot, __ATOMIC_RELAXED);
            if ((cur >> 40) != gen) {
                const unsigned long long prev = atomicCAS(table + slot, cur, mine);
                if (prev == cur) {  // first item with this tuple
                    atomicAdd(&s_new, 1);
                    break;
                }
                cur = prev;
                if ((cur >> 40) != gen) continue;  // lost a race against a stale view of the slot: look again
            }
            const int64_t other = (int64_t)(cur & ((1ull << 40) - 1ull));
            bool same = true;
            for (int i = 0; i < L; i++) same = same && (ids[other * L + i] == ids[b * L + i]);
            if (same) break;  // duplicate of an already counted tuple
            slot = slot + 1 == tsize ? 0 : slot + 1;
            probe++;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // ONE atomic per workgroup: tickets in the low half, distinct tuples in the high half, so the workgroup that draws the
        // last ticket learns the final count from the value its own add returns (no fence, no second counter)
        const unsigned long long old = atomicAdd(ctrl, ((unsigned long long)s_new << 32) + 1ull);
        s_last = (old & 0xFFFFFFFFull) == (unsigned long long)gridDim.x - 1ull;
        s_new += (int)(old >> 32);
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) {
        *p_unique = (float)s_new / (float)B;
        ctrl[0] = 0ull;
        ctrl[1] = 0ull;
        ctrl[2] = calls + 1ull;
    }
    if ((calls + 1ull) % GEN_MOD == 0ull)
        for (int64_t i = threadIdx.x; i < tsize; i += 256) table[i] = 0ull;
}

// CategoricalReconstructionLoss.forward as a stand-alone module (loss.py:15-33): x_hat is taken as given.
// out[m] = sum_{j<H} (x_hat - x)^2 + sum_{j>=H} BCE-with-logits(x_hat, x);  g_xhat = g[m] * (2 (x_hat - x) | sigmoid(x_hat) - x)
__global__ __launch_bounds__(256) void cat_recon_rows_kernel(const float *xh, int64_t ldh, const float *x, int64_t ldx, int64_t M, int64_t N,
                                                             int n_cat, const float *g, int64_t g_stride, float *out, float *g_xhat) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int64_t H = N - n_cat;
    const float gr = g != nullptr ? g[row * g_stride] : 0.0f;
    float rs = 0.0f;
    for (int64_t i = lane; i < N; i += 64) {
        const float u = xh[row * ldh + i], t = x[row * ldx + i];
        if (i < H) {
            const float d = u - t;
            rs += d * d;
            if (g_xhat != nullptr) g_xhat[row * N + i] = gr * (2.0f * d);
        } else {
            rs += (fmaxf(u, 0.0f) - u * t) + log1pf(expf(-fabsf(u)));
            if (g_xhat != nullptr) g_xhat[row * N + i] = gr * (1.0f / (1.0f + expf(-u)) - t);
        }
    }
    rs = hv_wave_sum(rs);
    if (lane == 0 && out != nullptr) out[row] = rs;
}

}  // namespace

extern "C" int hidvae_cat_recon_rows(const float *x_hat, int64_t ldh, const float *x, int64_t ldx, int64_t M, int64_t N, int n_cat,
                                     const float *g, int64_t g_stride, float *out, float *g_xhat, void *stream) {
    HV_REQUIRE(x_hat && x && M >= 1 && N >= 1 && (out || g_xhat), "cat_recon_rows: bad arguments");
    HV_REQUIRE(n_cat >= 0 && n_cat <= N, "cat_recon_rows: n_cat=%d of %lld columns", n_cat, (long long)N);
    HV_REQUIRE(g_xhat == nullptr || g != nullptr, "cat_recon_rows: a gradient output needs the incoming per-row gradient");
    hipLaunchKernelGGL(cat_recon_rows_kernel, dim3((unsigned)hv_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x_hat, ldh, x, ldx, M, N, n_cat, g,
                       g_stride, out, g_xhat);
    HV_LAUNCH_CHECK("cat_recon_rows");
    return HIDVAE_OK;
}

extern "C" int hidvae_recon_fwd_bwd(const float *y, const float *x, int64_t B, int64_t N, int n_cat, float gscale,
                                    const float *gscale_items, int64_t gs_stride, float *x_hat, float *recon, float *g_y,
                                    void *stream) {
    HV_REQUIRE(y && x && B >= 1 && N >= 1, "recon: bad arguments");
    HV_REQUIRE(n_cat >= 0 && n_cat < N, "recon: n_cat=%d of %lld columns", n_cat, (long long)N);
    const unsigned grid = (unsigned)hv_cdiv(B, 4);
    const bool vec = (N % 4 == 0) && N <= 256 * MAXV && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(x) |
                                                          reinterpret_cast<uintptr_t>(x_hat) | reinterpret_cast<uintptr_t>(g_y)) & 15) == 0;
    if (vec) hipLaunchKernelGGL(recon_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, x, B, N, n_cat, gscale, gscale_items, gs_stride, x_hat, recon, g_y);
    else hipLaunchKernelGGL(recon_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, x, B, N, n_cat, gscale, gscale_items, gs_stride, x_hat, recon, g_y);
    HV_LAUNCH_CHECK("recon_fwd_bwd");
    return HIDVAE_OK;
}

extern "C" int hidvae_adamw_prepare(int64_t *step_dev, const float *base_lr_dev, const float *wd_dev, int n_tensors, float beta1,
                                    float beta2, float eta_min, int64_t T_max, int64_t step_size, float gamma, float *hyper_dev,
                                    void *stream) {
    HV_REQUIRE(step_dev && base_lr_dev && wd_dev && hyper_dev && n_tensors >= 1, "adamw_prepare: bad arguments");
    const HvAdamPrepare a{step_dev, base_lr_dev, wd_dev, n_tensors, beta1, beta2, eta_min, T_max, step_size, gamma, hyper_dev};
    hipLaunchKernelGGL(adamw_prepare_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    HV_LAUNCH_CHECK("adamw_prepare");
    return HIDVAE_OK;
}

extern "C" int hidvae_adamw_step(float *const *p_dev, const float *const *g_host, float *const *m_dev, float *const *v_dev,
                                 const int64_t *numel_dev, const float *hyper_dev, int n_tensors, int64_t max_numel, float beta1,
                                 float beta2, float eps, float grad_scale, void *stream) {
    HV_REQUIRE(p_dev && g_host && m_dev && v_dev && numel_dev && hyper_dev, "adamw: null pointer");
    HV_REQUIRE(n_tensors >= 1 && max_numel >= 1, "adamw: bad tensor count / size");
    int64_t gx = hv_cdiv(max_numel, 256 * 4);
    if (gx > 256) gx = 256;
    hipStream_t s = (hipStream_t)stream;
    for (int first = 0; first < n_tensors; first += ADAM_CHUNK) {
        const int cnt = n_tensors - first < ADAM_CHUNK ? n_tensors - first : ADAM_CHUNK;
        AdamArgs a{};
        a.p = p_dev; a.m = m_dev; a.v = v_dev; a.numel = numel_dev; a.hyper = hyper_dev;
        a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.grad_scale = grad_scale;
        a.first = first;
        for (int i = 0; i < cnt; i++) {
            HV_REQUIRE(g_host[first + i] != nullptr, "adamw: gradient %d is null", first + i);
            a.g[i] = g_host[first + i];
        }
        hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)gx, (unsigned)cnt), dim3(256), 0, s, a);
        HV_LAUNCH_CHECK("adamw");
    }
    return HIDVAE_OK;
}

extern "C" int hidvae_id_stats(const float *emb_cat, int64_t ld_cat, const int64_t *ids, int64_t B, int L, float *embs_norm,
                               float *p_unique, int64_t *scratch, int embed_dim, void *stream) {
    HV_REQUIRE(ids && p_unique && scratch && B >= 1 && L >= 1 && L <= HIDVAE_MAX_LEVELS, "id_stats: bad arguments");
    HV_REQUIRE(embed_dim >= 4 && embed_dim <= 64 && embed_dim % 4 == 0, "id_stats: embed_dim=%d (a multiple of 4, at most 64)", embed_dim);
    HV_REQUIRE(embs_norm == nullptr || (emb_cat != nullptr && ld_cat >= (int64_t)L * embed_dim && ld_cat % 4 == 0), "id_stats: emb_cat/ld_cat");
    const int64_t tsize = 4 * B;  // + 3 control words
    const int64_t n0 = B * (embs_norm != nullptr && L > 1 ? L : 1);
    hipLaunchKernelGGL(id_stats_kernel, dim3((unsigned)hv_cdiv(n0, 256)), dim3(256), 0, (hipStream_t)stream, emb_cat, ld_cat, ids, B, L,
                       embs_norm, reinterpret_cast<unsigned long long *>(scratch), tsize, p_unique, embed_dim);
    HV_LAUNCH_CHECK("id_stats");
    return HIDVAE_OK;
}

extern "C" int hidvae_l2norm_fwd(const float *x, int64_t M, int64_t N, int64_t ldx, float eps, float *out, int64_t ldo,
                                 float *norms, void *stream) {
    HV_REQUIRE(x && out && M >= 1 && N >= 1 && ldx >= N && ldo >= N, "l2norm_fwd: bad arguments");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)hv_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, M, N, ldx, eps, out,
                       ldo, norms);
    HV_LAUNCH_CHECK("l2norm_fwd");
    return HIDVAE_OK;
}

extern "C" int hidvae_l2norm_fwd_pair(const float *x0, int64_t ldx0, int64_t N0, float *out0, float *norms0, const float *x1, int64_t ldx1,
                                      int64_t N1, float *out1, float *norms1, int64_t M, float eps, void *stream) {
    HV_REQUIRE(x0 && x1 && out0 && out1 && norms0 && norms1 && M >= 1 && N0 >= 1 && N1 >= 1 && ldx0 >= N0 && ldx1 >= N1, "l2norm_fwd_pair: bad arguments");
    L2Pair p{};
    p.a[0] = x0; p.lda[0] = ldx0; p.a[1] = x1; p.lda[1] = ldx1;
    p.out[0] = out0; p.ldo[0] = N0; p.out[1] = out1; p.ldo[1] = N1;
    p.norms_out[0] = norms0; p.norms_out[1] = norms1;
    p.N[0] = N0; p.N[1] = N1; p.M = M; p.eps = eps;
    hipLaunchKernelGGL(l2norm_fwd_pair_kernel, dim3((unsigned)(2 * hv_cdiv(M, 4))), dim3(256), 0, (hipStream_t)stream, p);
    HV_LAUNCH_CHECK("l2norm_fwd_pair");
    return HIDVAE_OK;
}

extern "C" int hidvae_l2norm_bwd_pair(const float *g0, int64_t ldg0, const float *out0, const float *norms0, int64_t N0, float *gx0,
                                      const float *g1, int64_t ldg1, const float *out1, const float *norms1, int64_t N1, float *gx1, int64_t M,
                                      float eps, void *stream) {
    HV_REQUIRE(g0 && g1 && out0 && out1 && norms0 && norms1 && gx0 && gx1 && M >= 1 && N0 >= 1 && N1 >= 1 && ldg0 >= N0 && ldg1 >= N1,
               "l2norm_bwd_pair: bad arguments");
    L2Pair p{};
    p.a[0] = g0; p.lda[0] = ldg0; p.a[1] = g1; p.lda[1] = ldg1;
    p.b[0] = out0; p.ldb[0] = N0; p.b[1] = out1; p.ldb[1] = N1;
    p.norms_in[0] = norms0; p.norms_in[1] = norms1;
    p.out[0] = gx0; p.ldo[0] = N0; p.out[1] = gx1; p.ldo[1] = N1;
    p.N[0] = N0; p.N[1] = N1; p.M = M; p.eps = eps;
    hipLaunchKernelGGL(l2norm_bwd_pair_kernel, dim3((unsigned)(2 * hv_cdiv(M, 4))), dim3(256), 0, (hipStream_t)stream, p);
    HV_LAUNCH_CHECK("l2norm_bwd_pair");
    return HIDVAE_OK;
}

extern "C" int hidvae_l2norm_bwd(const float *g, int64_t ldg, const float *out, int64_t ldo, const float *norms, int64_t M,
                                 int64_t N, float eps, float *gx, int64_t ldgx, int accumulate, void *stream) {
    HV_REQUIRE(g && out && norms && gx && M >= 1 && N >= 1 && ldg >= N && ldo >= N && ldgx >= N, "l2norm_bwd: bad arguments");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)hv_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, g, ldg, out, ldo, norms,
                       M, N, eps, gx, ldgx, accumulate);
    HV_LAUNCH_CHECK("l2norm_bwd");
    return HIDVAE_OK;
}

// ---------------------------------------------------------------------------------------------------
// uniqueness loss exactly as HRqVae.forward calls it (h_rqvae.py:630-631 passes ids transposed to [L,B], so the
// LEVELS play the batch role of SemanticIdUniquenessLoss.forward, h_rqvae.py:41-105 -- SURVEY Q3), fused with the
// total loss (h_rqvae.py:634-640).  Tiny: one workgroup; wave 0 evaluates the uniqueness term.
// ---------------------------------------------------------------------------------------------------
namespace {

// executed by ONE full wave; returns the loss on every lane; g_rows [L,D] (optional) = d loss / d z[0:L].  A lane per component
// (lane < D <= 64; the lanes past D hold zeros, which the wave sums leave exact: at D = 32 the results are bit for bit those of
// the 32-lane form this replaces)
__device__ float uniq_loss_wave(const int64_t *ids, const float *z, int64_t B, int L, float weight, float margin,
                                float *g_rows, int D) {
    const int lane = threadIdx.x & 63;
    const int d = lane;
    const bool on = lane < D;
    if (g_rows != nullptr)
        for (int i = lane; i < L * D; i += 64) g_rows[i] = 0.0f;
    float total = 0.0f;
    int count = 0;
    unsigned long long flagged = 0ull;  // bit (a*8+b): levels a<b carry identical id vectors over the whole batch
    for (int a = 0; a < L; a++)
        for (int b = a + 1; b < L; b++) {
            // blocks of 8 x 64 items with independent loads (no short-circuit inside a block), leaving at the first block that
            // shows a difference -- which is the first block unless two levels really agree (a dependent one-load-at-a-time scan of
            // the whole batch cost 29 us at B = 8192)
            bool eq = true;
            for (int64_t i0 = 0; i0 < B && eq; i0 += 512) {
                bool e8 = true;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int64_t i = i0 + j * 64 + lane;
                    const int64_t ia = i < B ? ids[i * L + a] : 0, ib = i < B ? ids[i * L + b] : 0;
                    e8 = e8 & (ia == ib);
                }
                eq = __all(e8);
            }
            if (eq) { flagged |= 1ull << (a * 8 + b); count++; }
        }
    if (count > 0) {
        for (int a = 0; a < L; a++)
            for (int b = a + 1; b < L; b++) {
                if (!((flagged >> (a * 8 + b)) & 1ull) || b >= B) continue;  // (b >= B: the reference would raise IndexError)
                const float za = on ? z[(int64_t)a * D + d] : 0.0f, zb = on ? z[(int64_t)b * D + d] : 0.0f;
                const float na = hv_wave_sum(za * za), nb = hv_wave_sum(zb * zb);
                const float da = fmaxf(sqrtf(na), 1e-12f), db = fmaxf(sqrtf(nb), 1e-12f);
                const float ha = za / da, hb = zb / db;
                const float ab = hv_wave_sum(ha * hb);
                const float v = ab - margin;
                if (v > 0.0f) {
                    total += v;
                    if (g_rows != nullptr && on) {  // d cos / d za = (hb - ha cos) / |za|
                        const float c = weight / (float)count;
                        g_rows[a * D + d] += c * (hb - ha * ab) / da;
                        g_rows[b * D + d] += c * (ha - hb * ab) / db;
                    }
                }
            }
    }
    return count > 0 ? weight * (total / (float)count) : 0.0f;
}

__global__ __launch_bounds__(64) void uniq_loss_kernel(const int64_t *ids, const float *z, int64_t B, int L, float weight,
                                                       float margin, float *loss, float *g_rows, int D) {
    const float v = uniq_loss_wave(ids, z, B, L, weight, margin, g_rows, D);
    if (threadIdx.x == 0) *loss = v;
}

struct TotalArgs {
    const float *recon, *qloss;
    int64_t B;
    const float *align[HIDVAE_MAX_LEVELS], *pred[HIDVAE_MAX_LEVELS], *acc[HIDVAE_MAX_LEVELS];  // per-level device scalars
    int n_tag;        // levels that carry tag losses (0 = untagged batch)
    float tag_div;    // the reference divides the level sums by n_layers (h_rqvae.py:561-563)
    float *tagstats;  // [3 + 3*n_tag]: align, pred, acc means, then the three by-layer vectors
    const int64_t *ids;  // uniqueness term (nullptr = skip)
    const float *z;
    int L;
    float uniq_weight, uniq_margin;
    float w_a, w_p, w_u;
    float *loss, *uniq, *g_rows;
    int D;            // embedding width (rows of z / g_rows)
    float *summary;   // optional [6]: loss, mean recon, mean qloss, tag align, tag pred, tag accuracy (the training log's row)
};

// executed by one whole 256-thread workgroup
__device__ __forceinline__ void total_loss_body(const TotalArgs &a) {
    __shared__ float red[2][4];
    __shared__ float uq;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float r = 0.0f, q = 0.0f;
    for (int64_t i0 = tid; i0 < a.B; i0 += 8 * 256) {  // sixteen loads in flight, added in the same ascending order
        float rv[8], qv[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int64_t i = i0 + 256 * j;
            rv[j] = i < a.B ? a.recon[i] : 0.0f;
            qv[j] = i < a.B ? a.qloss[i] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) { r += rv[j]; q += qv[j]; }
    }
    r = hv_wave_sum(r);
    q = hv_wave_sum(q);
    if (lane == 0) { red[0][wave] = r; red[1][wave] = q; }
    if (wave == 3) {  // the last wave evaluates the uniqueness term meanwhile
        const float v = a.ids != nullptr ? uniq_loss_wave(a.ids, a.z, a.B, a.L, a.uniq_weight, a.uniq_margin, a.g_rows, a.D) : 0.0f;
        if (lane == 0) uq = v;
    }
    __syncthreads();
    if (tid == 0) {
        const float rm = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)a.B;
        const float qm = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)a.B;
        float al = 0.0f, pr = 0.0f, ac = 0.0f;
        for (int i = 0; i < a.n_tag; i++) {  // ascending level order, as the reference accumulates (h_rqvae.py:539-547)
            const float ai = *a.align[i], pi = *a.pred[i], ci = *a.acc[i];
            al += ai; pr += pi; ac += ci;
            if (a.tagstats != nullptr) {
                a.tagstats[3 + i] = ai;
                a.tagstats[3 + a.n_tag + i] = pi;
                a.tagstats[3 + 2 * a.n_tag + i] = ci;
            }
        }
        if (a.n_tag > 0) { al = al / a.tag_div; pr = pr / a.tag_div; ac = ac / a.tag_div; }
        if (a.tagstats != nullptr) { a.tagstats[0] = al; a.tagstats[1] = pr; a.tagstats[2] = ac; }
        float t = rm + qm;
        t = t + a.w_a * al;
        t = t + a.w_p * pr;
        t = t + a.w_u * uq;
        *a.loss = t;
        if (a.uniq != nullptr) *a.uniq = uq;
        if (a.summary != nullptr) {
            a.summary[0] = t; a.summary[1] = rm; a.summary[2] = qm; a.summary[3] = al; a.summary[4] = pr; a.summary[5] = ac;
        }
    }
}

__global__ __launch_bounds__(256) void total_loss_kernel(TotalArgs a) { total_loss_body(a); }

// backward of the same pair in ONE launch: g_y = (g/B) d recon/d y per row; scal / g_z as total_loss_bwd_kernel
template <bool VEC>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float *g_loss, const float *y, const float *x, int64_t B, int64_t N, int n_cat, int L,
                                                       float w_a, float w_p, float w_u, const float *g_rows, float *g_y, float *scal,
                                                       float *g_z, int D, float expect_g) {
    // expect_g != 0: the tag heads' backward already ran, seeded with that loss gradient; anything else arriving here would leave the
    // step with two different scalings -- every gradient this launch produces is made NaN instead, so the step fails visibly
    float g = *g_loss;
    if (expect_g != 0.0f && g != expect_g) g = NAN;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scal[0] = g / (float)B;
        scal[1] = g * w_a;
        scal[2] = g * w_p;
    }
    if (g_z != nullptr && threadIdx.x < 4 * D) {  // this workgroup's 4 rows of g_z
        const int64_t idx = (int64_t)blockIdx.x * (4 * D) + threadIdx.x;
        if (idx < B * D) g_z[idx] = (g_rows != nullptr && idx < (int64_t)L * D) ? (g * w_u) * g_rows[idx] : 0.0f;
    }
    if (row < B) recon_row<VEC>(y, x, row, N, n_cat, 1.0f * (g / (float)B), nullptr, nullptr, g_y);
}

// scal[0] = g/B, scal[1] = g*w_a, scal[2] = g*w_p ; g_z [B,D] = g*w_u*g_rows on the first L rows, 0 elsewhere
__global__ __launch_bounds__(256) void total_loss_bwd_kernel(const float *g_loss, int64_t B, int L, float w_a, float w_p, float w_u,
                                                             const float *g_rows, float *scal, float *g_z, int D) {
    const float g = *g_loss;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx == 0) {
        scal[0] = g / (float)B;
        scal[1] = g * w_a;
        scal[2] = g * w_p;
    }
    if (g_z != nullptr && idx < B * D) g_z[idx] = (g_rows != nullptr && idx < (int64_t)L * D) ? (g * w_u) * g_rows[idx] : 0.0f;
}

}  // namespace

extern "C" int hidvae_uniq_loss(const int64_t *ids, const float *z, int64_t B, int L, float weight, float margin, float *loss,
                                float *g_rows, int embed_dim, void *stream) {
    HV_REQUIRE(ids && z && loss && B >= 1 && L >= 1 && L <= HIDVAE_MAX_LEVELS, "uniq_loss: bad arguments");
    HV_REQUIRE(embed_dim >= 1 && embed_dim <= 64, "uniq_loss: embed_dim=%d (at most 64)", embed_dim);
    hipLaunchKernelGGL(uniq_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ids, z, B, L, weight, margin, loss, g_rows, embed_dim);
    HV_LAUNCH_CHECK("uniq_loss");
    return HIDVAE_OK;
}

extern "C" int hidvae_total_loss(const float *recon, const float *qloss, int64_t B, const float *const *align_host,
                                 const float *const *pred_host, const float *const *acc_host, int n_tag, float tag_div,
                                 const int64_t *ids, const float *z, int L, float uniq_weight, float uniq_margin, float w_a,
                                 float w_p, float w_u, float *loss, float *uniq, float *g_rows, float *tagstats, float *summary,
                                 int embed_dim, void *stream) {
    HV_REQUIRE(recon && qloss && loss && B >= 1, "total_loss: bad arguments");
    HV_REQUIRE(ids == nullptr || (z != nullptr && L >= 1 && L <= HIDVAE_MAX_LEVELS),
               "total_loss: uniqueness term needs z and 1 <= n_layers <= %d (L=%d)", HIDVAE_MAX_LEVELS, L);
    HV_REQUIRE(n_tag >= 0 && n_tag <= HIDVAE_MAX_LEVELS && (n_tag == 0 || (align_host && pred_host && acc_host && tag_div > 0.0f)),
               "total_loss: tag terms");
    TotalArgs a{};
    a.recon = recon; a.qloss = qloss; a.B = B; a.n_tag = n_tag; a.tag_div = tag_div; a.tagstats = tagstats;
    for (int i = 0; i < n_tag; i++) {
        HV_REQUIRE(align_host[i] && pred_host[i] && acc_host[i], "total_loss: null tag scalar at level %d", i);
        a.align[i] = align_host[i]; a.pred[i] = pred_host[i]; a.acc[i] = acc_host[i];
    }
    a.ids = ids; a.z = z; a.L = L; a.uniq_weight = uniq_weight; a.uniq_margin = uniq_margin;
    a.w_a = w_a; a.w_p = w_p; a.w_u = w_u; a.loss = loss; a.uniq = uniq; a.g_rows = g_rows; a.summary = summary;
    a.D = embed_dim;
    HV_REQUIRE(ids == nullptr || (embed_dim >= 1 && embed_dim <= 64), "loss: embed_dim=%d (at most 64)", embed_dim);
    hipLaunchKernelGGL(total_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    HV_LAUNCH_CHECK("total_loss");
    return HIDVAE_OK;
}

static bool recon_vec_ok(int64_t N, const void *a, const void *b, const void *c, const void *d) {
    return (N % 4 == 0) && N <= 256 * MAXV &&
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(d)) & 15) == 0;
}

extern "C" int hidvae_loss_fwd(const float *y, const float *x, int64_t B, int64_t N, int n_cat, const float *qloss,
                               const float *const *align_host, const float *const *pred_host, const float *const *acc_host, int n_tag,
                               float tag_div, const int64_t *ids, const float *z, int L, float uniq_weight, float uniq_margin, float w_a,
                               float w_p, float w_u, float *recon, float *loss, float *uniq, float *g_rows, float *tagstats,
                               float *summary, int embed_dim, void *stream) {
    HV_REQUIRE(y && x && qloss && recon && loss && B >= 1 && N >= 1, "loss_fwd: bad arguments");
    HV_REQUIRE(n_cat >= 0 && n_cat < N, "loss_fwd: n_cat=%d of %lld columns", n_cat, (long long)N);
    HV_REQUIRE(ids == nullptr || (z != nullptr && L >= 1 && L <= HIDVAE_MAX_LEVELS),
               "loss_fwd: uniqueness term needs z and 1 <= n_layers <= %d (L=%d)", HIDVAE_MAX_LEVELS, L);
    HV_REQUIRE(n_tag >= 0 && n_tag <= HIDVAE_MAX_LEVELS && (n_tag == 0 || (align_host && pred_host && acc_host && tag_div > 0.0f)),
               "loss_fwd: tag terms");
    TotalArgs a{};
    a.recon = recon; a.qloss = qloss; a.B = B; a.n_tag = n_tag; a.tag_div = tag_div; a.tagstats = tagstats;
    for (int i = 0; i < n_tag; i++) {
        HV_REQUIRE(align_host[i] && pred_host[i] && acc_host[i], "loss_fwd: null tag scalar at level %d", i);
        a.align[i] = align_host[i]; a.pred[i] = pred_host[i]; a.acc[i] = acc_host[i];
    }
    a.ids = ids; a.z = z; a.L = L; a.uniq_weight = uniq_weight; a.uniq_margin = uniq_margin;
    a.w_a = w_a; a.w_p = w_p; a.w_u = w_u; a.loss = loss; a.uniq = uniq; a.g_rows = g_rows; a.summary = summary;
    a.D = embed_dim;
    HV_REQUIRE(ids == nullptr || (embed_dim >= 1 && embed_dim <= 64), "loss: embed_dim=%d (at most 64)", embed_dim);
    // two launches: a grid-wide hand-off inside one launch needs an agent-scope release per workgroup, and on 8 XCDs that
    // L2 write-back costs several times the second launch (measured 25 us fused vs 5 + 6.6 us)
    const unsigned grid = (unsigned)hv_cdiv(B, 4);
    if (recon_vec_ok(N, y, x, nullptr, nullptr))
        hipLaunchKernelGGL(recon_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, x, B, N, n_cat, 0.0f, (const float *)nullptr,
                           (int64_t)0, (float *)nullptr, recon, (float *)nullptr);
    else
        hipLaunchKernelGGL(recon_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, x, B, N, n_cat, 0.0f, (const float *)nullptr,
                           (int64_t)0, (float *)nullptr, recon, (float *)nullptr);
    HV_LAUNCH_CHECK("loss_fwd recon");
    hipLaunchKernelGGL(total_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    HV_LAUNCH_CHECK("loss_fwd");
    return HIDVAE_OK;
}

extern "C" int hidvae_loss_bwd(const float *g_loss, const float *y, const float *x, int64_t B, int64_t N, int n_cat, int L, float w_a, float w_p,
                               float w_u, const float *g_rows, float *g_y, float *scal, float *g_z, int embed_dim, float expect_g, void *stream) {
    HV_REQUIRE(g_loss && y && x && g_y && scal && B >= 1 && N >= 1, "loss_bwd: bad arguments");
    HV_REQUIRE(n_cat >= 0 && n_cat < N, "loss_bwd: n_cat=%d of %lld columns", n_cat, (long long)N);
    HV_REQUIRE(g_z == nullptr || (embed_dim >= 1 && embed_dim <= 64), "loss_bwd: embed_dim=%d (at most 64)", embed_dim);
    const unsigned grid = (unsigned)hv_cdiv(B, 4);
    if (recon_vec_ok(N, y, x, g_y, nullptr))
        hipLaunchKernelGGL(loss_bwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, g_loss, y, x, B, N, n_cat, L, w_a, w_p, w_u, g_rows,
                           g_y, scal, g_z, embed_dim, expect_g);
    else
        hipLaunchKernelGGL(loss_bwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, g_loss, y, x, B, N, n_cat, L, w_a, w_p, w_u, g_rows,
                           g_y, scal, g_z, embed_dim, expect_g);
    HV_LAUNCH_CHECK("loss_bwd");
    return HIDVAE_OK;
}

extern "C" int hidvae_total_loss_bwd(const float *g_loss, int64_t B, int L, float w_a, float w_p, float w_u, const float *g_rows,
                                     float *scal, float *g_z, int embed_dim, void *stream) {
    HV_REQUIRE(g_loss && scal && B >= 1, "total_loss_bwd: bad arguments");
    HV_REQUIRE(g_z == nullptr || (embed_dim >= 1 && embed_dim <= 64), "total_loss_bwd: embed_dim=%d (at most 64)", embed_dim);
    const int64_t n = g_z != nullptr ? B * embed_dim : 1;
    hipLaunchKernelGGL(total_loss_bwd_kernel, dim3((unsigned)hv_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, g_loss, B, L, w_a,
                       w_p, w_u, g_rows, scal, g_z, embed_dim);
    HV_LAUNCH_CHECK("total_loss_bwd");
    return HIDVAE_OK;
}
