// L-level residual quantisation for embedding widths OTHER than 32 (embed_dim a multiple of 4, <= 64: the reference's
// configs/rqvae_ml32m.gin and configs/decoder_ml32m.gin use 64; modules/rqvae.py:37-88, modules/h_rqvae.py:231-256 take any width).
//
// The fused kernels of rq.hip are built around 32 = 4 quarter-lanes x 8 components.  This file is the same algorithm
// (modules/quantize.py:100-153 inside modules/h_rqvae.py:515-552) in a width-independent geometry, exact fp32 throughout, not tuned:
//   * a wave owns GI items; in the search phase a lane owns codes (k = 64 g + lane) and walks the components in ascending order --
//     one fmaf chain per (item, code), distance = fmaf(-2, dot, |r|^2 + |c|^2) as everywhere else, first minimum wins;
//   * in the per-level tail a lane owns component d (lane < D): winner row, rotation trick / STE / eval output, loss and residual
//     update are element-wise with wave sums (xor butterfly 32, 16, .. 1) for the five dot products.
// oracle/exact.c restates exactly this operation order (ORDER-GEN), so ids and floats are compared bit for bit in the tests, and the
// reference's own outputs at embed_dim = 64 pin the oracle (tests/golden/rqvae_rot_train_d64_*).
#include <math.h>
#include "common.h"

namespace {

constexpr int GI = 4;  // items per wave

struct GenFwd {
    const float *y; int64_t B; int normalize_input;
    const float *cb_eff, *cc;  // [L][K][D], [L][K]
    int L; int64_t K; int D;
    float beta;
    float *z; int64_t *ids; float *emb_cat; int64_t ld_cat; float *emb_sum, *res_cat, *qloss;
    int cosine;  // QuantizeDistance.COSINE (quantize.py:115-119): the search ranks by -(r/|r| . c) / |c|; outputs and loss as ever
};

template <int MODE, bool TRAIN>
__device__ __forceinline__ float gen_output(float r, float e, float xx, float cce) {
    if (!TRAIN) return e;
    if (MODE == HIDVAE_MODE_STE) return r + (e - r);
    const float inr = 1.0f / (sqrtf(xx) + 1e-8f), ine = 1.0f / (sqrtf(cce) + 1e-8f);
    const float u = r * inr, qv = e * ine, s = u + qv;
    const float inw = 1.0f / fmaxf(sqrtf(hv_wave_sum(s * s)), 1e-6f);
    const float w = s * inw;
    const float rw = hv_wave_sum(r * w), ru = hv_wave_sum(r * u);
    return (r - 2.0f * (rw * w)) + 2.0f * (ru * qv);
}

template <int MODE, bool TRAIN>
__global__ __launch_bounds__(256) void rq_generic_fwd_kernel(GenFwd a) {
    __shared__ float rbuf[4][GI][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = a.D;
    const int64_t item0 = ((int64_t)blockIdx.x * 4 + wave) * GI;
    float r[GI], esum[GI], loss[GI];
    bool live[GI];
#pragma unroll
    for (int i = 0; i < GI; i++) {
        const int64_t item = item0 + i;
        live[i] = item < a.B;
        const int64_t src = live[i] ? item : a.B - 1;
        r[i] = lane < D ? a.y[src * D + lane] : 0.0f;
        if (a.normalize_input) {
            const float den = fmaxf(sqrtf(hv_wave_sum(r[i] * r[i])), 1e-12f);
            r[i] = r[i] / den;
        }
        if (a.z != nullptr && live[i] && lane < D) a.z[item * D + lane] = r[i];
        esum[i] = 0.0f;
        loss[i] = 0.0f;
    }
    for (int lvl = 0; lvl < a.L; lvl++) {
        const float *C = a.cb_eff + (int64_t)lvl * a.K * D;
        const float *cc = a.cc + (int64_t)lvl * a.K;
        float xx[GI], best[GI];
        int bidx[GI];
#pragma unroll
        for (int i = 0; i < GI; i++) {
            if (a.res_cat != nullptr && live[i] && lane < D) a.res_cat[(item0 + i) * ((int64_t)a.L * D) + lvl * D + lane] = r[i];
            xx[i] = hv_wave_sum(r[i] * r[i]);
            rbuf[wave][i][lane] = a.cosine ? r[i] / sqrtf(xx[i]) : r[i];  // (no epsilon: x / x.norm(), quantize.py:117)
            best[i] = INFINITY;
            bidx[i] = 0;
        }
        __builtin_amdgcn_wave_barrier();
        for (int64_t k0 = 0; k0 < a.K; k0 += 64) {
            const int64_t k = k0 + lane;
            const bool kok = k < a.K;
            const float *row = C + (kok ? k : a.K - 1) * D;
            float dot[GI];
#pragma unroll
            for (int i = 0; i < GI; i++) dot[i] = 0.0f;
            for (int d4 = 0; d4 < D; d4 += 4) {
                const float4 c4 = *reinterpret_cast<const float4 *>(row + d4);
#pragma unroll
                for (int i = 0; i < GI; i++) {
                    const float4 r4 = *reinterpret_cast<const float4 *>(&rbuf[wave][i][d4]);  // (a broadcast read: every lane the same address)
                    dot[i] = fmaf(r4.x, c4.x, dot[i]);
                    dot[i] = fmaf(r4.y, c4.y, dot[i]);
                    dot[i] = fmaf(r4.z, c4.z, dot[i]);
                    dot[i] = fmaf(r4.w, c4.w, dot[i]);
                }
            }
            const float cck = kok ? cc[k] : INFINITY;
            const float cn = sqrtf(cck);
#pragma unroll
            for (int i = 0; i < GI; i++) {
                const float dist = !a.cosine ? fmaf(-2.0f, dot[i], xx[i] + cck) : (kok ? -(dot[i] / cn) : INFINITY);
                if (dist < best[i]) { best[i] = dist; bidx[i] = (int)k; }  // ascending k inside the lane: strict < keeps the first minimum
            }
        }
#pragma unroll
        for (int i = 0; i < GI; i++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {  // across the lanes: equal distances resolve to the lowest code index
                const float ob = __shfl_xor(best[i], o);
                const int oi = __shfl_xor(bidx[i], o);
                if (ob < best[i] || (ob == best[i] && oi < bidx[i])) { best[i] = ob; bidx[i] = oi; }
            }
            const float e = lane < D ? C[(int64_t)bidx[i] * D + lane] : 0.0f;
            const float cce = cc[bidx[i]];
            const float o = gen_output<MODE, TRAIN>(r[i], e, xx[i], cce);
            const float df = r[i] - e;
            const float l1 = hv_wave_sum(df * df);  // loss.py:41-44: both terms are numerically |r-e|^2
            loss[i] = loss[i] + (l1 + a.beta * l1);
            if (live[i]) {
                if (lane == 0) a.ids[(item0 + i) * a.L + lvl] = (int64_t)bidx[i];
                if (a.emb_cat != nullptr && lane < D) a.emb_cat[(item0 + i) * a.ld_cat + lvl * D + lane] = o;
            }
            esum[i] = lvl == 0 ? o : esum[i] + o;
            r[i] = r[i] - o;
        }
        __builtin_amdgcn_wave_barrier();  // rbuf is rewritten by the next level
    }
#pragma unroll
    for (int i = 0; i < GI; i++)
        if (live[i]) {
            if (a.emb_sum != nullptr && lane < D) a.emb_sum[(item0 + i) * D + lane] = esum[i];
            if (a.qloss != nullptr && lane == 0) a.qloss[item0 + i] = loss[i];
        }
}

struct GenTables {  // per-level pointers and flags travel by value in the kernel arguments (the entry points receive HOST arrays)
    const float *E[HIDVAE_MAX_LEVELS];
    float *gE[HIDVAE_MAX_LEVELS];
    int normalize[HIDVAE_MAX_LEVELS];
};

// effective codebook: optional row normalisation + |c|^2, one wave per row
__global__ __launch_bounds__(256) void codebook_prepare_generic_kernel(GenTables t, int L, int64_t K, int D, float *cb_eff, float *cc) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)L * K) return;
    const int lvl = (int)(row / K);
    const int64_t k = row - (int64_t)lvl * K;
    float v = lane < D ? t.E[lvl][k * D + lane] : 0.0f;
    if (t.normalize[lvl]) v = v / fmaxf(sqrtf(hv_wave_sum(v * v)), 1e-12f);
    const float c2 = hv_wave_sum(v * v);
    if (lane < D) cb_eff[row * D + lane] = v;
    if (lane == 0) cc[row] = c2;
}

struct GenBwd {
    const float *y, *z; int64_t B; int normalize_input;
    const float *cb_eff, *cc; int L; int64_t K; int D; float beta;
    const int64_t *ids;
    const float *g_cat; int64_t ld_gcat; const float *g_sum, *g_z_in; int64_t g_z_rows;
    float gq; const float *gq_items; int64_t gq_stride;
    float *g_y, *dE_rows;
};

// the backward of rq_backward_kernel (rq.hip) with a lane per component: one wave per item
template <int MODE>
__global__ __launch_bounds__(256) void rq_generic_bwd_kernel(GenBwd a) {
    const int lane = threadIdx.x & 63;
    const int D = a.D, L = a.L;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= a.B) return;
    const bool on = lane < D;
    const float zr = on ? a.z[item * D + lane] : 0.0f;
    float rs[HIDVAE_MAX_LEVELS];
    {
        float r = zr;
#pragma unroll
        for (int i = 0; i < HIDVAE_MAX_LEVELS; i++) {  // (constant bounds + a guard keep rs[] in registers)
            if (i >= L) break;
            rs[i] = r;
            if (i + 1 < L) {
                const int64_t code = a.ids[item * L + i];
                const float e = on ? a.cb_eff[((int64_t)i * a.K + code) * D + lane] : 0.0f;
                const float o = gen_output<MODE, true>(r, e, hv_wave_sum(r * r), a.cc[(int64_t)i * a.K + code]);
                r = r - o;
            }
        }
    }
    const float gs = (a.g_sum != nullptr && on) ? a.g_sum[item * D + lane] : 0.0f;
    float R = 0.0f;
#pragma unroll
    for (int i = HIDVAE_MAX_LEVELS - 1; i >= 0; i--) {
        if (i >= L) continue;
        const int64_t code = a.ids[item * L + i];
        const float e = on ? a.cb_eff[((int64_t)i * a.K + code) * D + lane] : 0.0f;
        float go = (a.g_cat != nullptr && on) ? a.g_cat[item * a.ld_gcat + i * D + lane] : 0.0f;
        go = (go + gs) - R;  // o_i feeds sum, concat and -r_{i+1}
        float jt;
        if (MODE == HIDVAE_MODE_STE) jt = go;
        else {  // d o / d r = I - 2 w w^T + 2 q u^T with (u, q, w) constants  =>  J^T g = g - 2 w (w.g) + 2 u (q.g)
            const float r = rs[i];
            const float xx = hv_wave_sum(r * r), cce = a.cc[(int64_t)i * a.K + code];
            const float inr = 1.0f / (sqrtf(xx) + 1e-8f), ine = 1.0f / (sqrtf(cce) + 1e-8f);
            const float u = r * inr, qv = e * ine, s = u + qv;
            const float inw = 1.0f / fmaxf(sqrtf(hv_wave_sum(s * s)), 1e-6f);
            const float w = s * inw;
            const float wg = hv_wave_sum(w * go), qg = hv_wave_sum(qv * go);
            jt = (go - 2.0f * (wg * w)) + 2.0f * (qg * u);
        }
        const float gqb = a.gq_items != nullptr ? a.gq * a.gq_items[item * a.gq_stride] : a.gq;
        const float cq = 2.0f * gqb, cr = 2.0f * a.beta * gqb;
        const float df = rs[i] - e;
        R = (R + jt) + cr * df;       // commitment term: beta |r - sg(e)|^2
        const float de = -(cq * df);  // codebook term:   |sg(r) - e|^2
        if (a.dE_rows != nullptr && on) a.dE_rows[item * ((int64_t)L * D) + i * D + lane] = de;
    }
    if (a.g_z_in != nullptr && item < a.g_z_rows && on) R = R + a.g_z_in[item * D + lane];
    if (a.normalize_input) {  // z = y / max(|y|, eps)  =>  g_y = (g_z - z (z.g_z)) / max(|y|, eps)
        const float yv = on ? a.y[item * D + lane] : 0.0f;
        const float den = fmaxf(sqrtf(hv_wave_sum(yv * yv)), 1e-12f);
        const float zg = hv_wave_sum(zr * R);
        R = (R - zr * zg) / den;
    }
    if (on) a.g_y[item * D + lane] = R;
}

// codebook gradient: one wave per (level, code), lane = component; the items are scanned in ascending order (64 ids per step, the
// matching rows added one after the other in item order) => bit-reproducible sums
__global__ __launch_bounds__(256) void codebook_grad_generic_kernel(const int64_t *ids, const float *dE_rows, int64_t B, int L, int64_t K, int D,
                                                                    GenTables t, const float *cb_eff, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)L * K) return;
    const int lvl = (int)(row / K);
    const int64_t k = row - (int64_t)lvl * K;
    const bool on = lane < D;
    float acc = 0.0f;
    for (int64_t b0 = 0; b0 < B; b0 += 64) {
        const int64_t b = b0 + lane;
        unsigned long long m = __ballot(b < B && ids[b * L + lvl] == k);
        while (m != 0ull) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            if (on) acc += dE_rows[(b0 + j) * ((int64_t)L * D) + lvl * D + lane];
        }
    }
    if (t.normalize[lvl]) {  // c = E / max(|E|, eps)  =>  gE = (g - c (c.g)) / max(|E|, eps)
        const float ev = on ? t.E[lvl][k * D + lane] : 0.0f;
        const float cv = on ? cb_eff[row * D + lane] : 0.0f;
        const float n2 = hv_wave_sum(ev * ev), cg = hv_wave_sum(cv * acc);
        acc = (acc - cv * cg) / fmaxf(sqrtf(n2), 1e-12f);
    }
    if (on) {
        float *dst = t.gE[lvl] + k * D + lane;
        *dst = accumulate ? *dst + acc : acc;
    }
}

}  // namespace

// ---- called from the extern "C" entry points of rq.hip when embed_dim != 32 (declared in common.h) -----------------------------
bool hv_rqg_dim_ok(int D) { return D >= 4 && D <= 64 && D % 4 == 0; }

int hv_rqg_prepare(const float *const *E_host, const int32_t *normalize_host, int L, int64_t K, int D, float *cb_eff, float *cc, hipStream_t s) {
    GenTables t{};
    for (int i = 0; i < L; i++) { t.E[i] = E_host[i]; t.normalize[i] = normalize_host ? normalize_host[i] : 0; }
    hipLaunchKernelGGL(codebook_prepare_generic_kernel, dim3((unsigned)hv_cdiv((int64_t)L * K, 4)), dim3(256), 0, s, t, L, K, D, cb_eff, cc);
    HV_LAUNCH_CHECK("codebook_prepare (generic width)");
    return HIDVAE_OK;
}

int hv_rqg_forward(const float *y, int64_t B, int normalize_input, const float *cb_eff, const float *cc, int L, int64_t K, int D, int mode,
                   int training, float beta, float *z, int64_t *ids, float *emb_cat, int64_t ld_cat, float *emb_sum, float *res_cat,
                   float *qloss, int cosine, hipStream_t s) {
    GenFwd a{y, B, normalize_input, cb_eff, cc, L, K, D, beta, z, ids, emb_cat, ld_cat, emb_sum, res_cat, qloss, cosine};
    const dim3 grid((unsigned)hv_cdiv(B, 4 * GI));
    if (!training) hipLaunchKernelGGL((rq_generic_fwd_kernel<HIDVAE_MODE_STE, false>), grid, dim3(256), 0, s, a);
    else if (mode == HIDVAE_MODE_STE) hipLaunchKernelGGL((rq_generic_fwd_kernel<HIDVAE_MODE_STE, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((rq_generic_fwd_kernel<HIDVAE_MODE_ROTATION, true>), grid, dim3(256), 0, s, a);
    HV_LAUNCH_CHECK("rq_forward (generic width)");
    return HIDVAE_OK;
}

int hv_rqg_backward(const float *y, const float *z, int64_t B, int normalize_input, const float *cb_eff, const float *cc, int L, int64_t K, int D,
                    int mode, float beta, const int64_t *ids, const float *g_cat, int64_t ld_gcat, const float *g_sum, const float *g_z_in,
                    int64_t g_z_rows, float gq, const float *gq_items, int64_t gq_stride, float *g_y, float *dE_rows, hipStream_t s) {
    GenBwd a{y, z, B, normalize_input, cb_eff, cc, L, K, D, beta, ids, g_cat, ld_gcat, g_sum, g_z_in, g_z_rows, gq, gq_items, gq_stride, g_y, dE_rows};
    const dim3 grid((unsigned)hv_cdiv(B, 4));
    if (mode == HIDVAE_MODE_STE) hipLaunchKernelGGL(rq_generic_bwd_kernel<HIDVAE_MODE_STE>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(rq_generic_bwd_kernel<HIDVAE_MODE_ROTATION>, grid, dim3(256), 0, s, a);
    HV_LAUNCH_CHECK("rq_backward (generic width)");
    return HIDVAE_OK;
}

int hv_rqg_codebook_grad(const int64_t *ids, const float *dE_rows, int64_t B, int L, int64_t K, int D, const float *const *E_host,
                         const float *cb_eff, const int32_t *normalize_host, float *const *gE_host, int accumulate, hipStream_t s) {
    GenTables t{};
    for (int i = 0; i < L; i++) { t.E[i] = E_host[i]; t.gE[i] = gE_host[i]; t.normalize[i] = normalize_host ? normalize_host[i] : 0; }
    hipLaunchKernelGGL(codebook_grad_generic_kernel, dim3((unsigned)hv_cdiv((int64_t)L * K, 4)), dim3(256), 0, s, ids, dE_rows, B, L, K, D, t,
                       cb_eff, accumulate);
    HV_LAUNCH_CHECK("codebook_grad (generic width)");
    return HIDVAE_OK;
}
