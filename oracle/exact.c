/* TEST INFRASTRUCTURE ONLY (oracle/): plain-C restatement of the ID-determining part of the
 * HiD-VAE tokenizer (encoder MLP -> L-level residual quantisation) in a FIXED fp32 operation
 * order, so that semantic ids -- and every float on this path -- can be compared BIT-FOR-BIT with
 * the HIP kernels.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it.
 *
 * Algorithm (paths relative to /root/reference):
 *   encoder   modules/encoder.py:23-36            Linear(no bias) -> SiLU ... -> Linear -> l2norm
 *   level i   modules/quantize.py:100-153         dist = |r|^2 + |c|^2 - 2 r.c ; first argmin
 *   rotation  modules/quantize.py:34-45,134-140   o = r - 2 (r.w) w + 2 (r.u) q
 *   STE       modules/quantize.py:131-133         o = r + (e - r)
 *   loss      modules/loss.py:41-44               |r-e|^2 + beta |r-e|^2
 *   chain     modules/h_rqvae.py:515-552          r <- r - o ; level-0 codebook row-normalised (:295)
 * Parity: pinned against the reference's outputs in tests/golden (tests/test_exact_oracle.py);
 * the association order inside each sum is this file's own (MKL's is unknowable) and is what the
 * HIP kernels implement:
 *   ORDER-G  GEMM: one fmaf chain per output from +0 over 16-wide k-blocks ascending; inside a block the
 *            chain visits k = 0,4,8,12, 1,5,9,13, 2,6,10,14, 3,7,11,15 ("ORDER-G16": f32 MFMA 16x16x4 with lane
 *            quarter q holding k = 4q+s at step s; the 32x32x2 kernels feed their operands in the same order)
 *   ORDER-Q  sums over D=32: four chains over the contiguous quarters d in [8q,8q+8), then
 *            (p0+p1)+(p2+p3)
 *   ORDER-P  r.c dot: one fmaf chain visiting d = 0,8,16,24, 1,9,17,25, ... 7,15,23,31
 *            (MFMA 16x16x4 with lane-quarter q holding d = 8q+j, step j)
 *   ORDER-GEN (embedding widths other than 32, csrc/rq_generic.hip): sums over the D <= 64 components are a 64-lane
 *            xor butterfly (lane d holds term d, lanes >= D hold 0; steps 32, 16, 8, 4, 2, 1); the r.c dot is ONE fmaf
 *            chain over d = 0, 1, ..., D-1
 *   expE     own exp: Cody-Waite reduction + degree-5 Cephes polynomial, all fmaf (no libm).
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -mfma -mavx2 -fPIC -shared
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define D 32

static inline float expE(float x) {
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

float orc_exp(float x) { return expE(x); }
float orc_silu(float a) { return a / (1.0f + expE(-a)); }

/* ORDER-G.  y[m,n] = sum_k x[m,k] w[n,k]  (F.linear without bias, encoder.py:27).  Skipped k >= K slots are
 * exact no-ops on the GPU too (the kernel feeds 0*0 there). */
void orc_linear(const float *x, int64_t M, int64_t K, const float *w, int64_t N, float *y, int silu) {
    float *wt = (float *)malloc(sizeof(float) * (size_t)K * (size_t)N); /* [K][N] so n vectorises */
    for (int64_t n = 0; n < N; n++)
        for (int64_t k = 0; k < K; k++) wt[k * N + n] = w[n * K + k];
    for (int64_t m = 0; m < M; m++) {
        float *ym = y + m * N;
        for (int64_t n = 0; n < N; n++) ym[n] = 0.0f;
        for (int64_t k0 = 0; k0 < K; k0 += 16)
            for (int s = 0; s < 4; s++)
                for (int h = 0; h < 4; h++) {
                    const int64_t k = k0 + 4 * h + s;
                    if (k >= K) continue;
                    const float xv = x[m * K + k];
                    const float *wk = wt + k * N;
                    for (int64_t n = 0; n < N; n++) ym[n] = fmaf(xv, wk[n], ym[n]);
                }
        if (silu)
            for (int64_t n = 0; n < N; n++) ym[n] = orc_silu(ym[n]);
    }
    free(wt);
}

/* ORDER-Q sum of a[d]*b[d] */
static inline float dotQ(const float *a, const float *b) {
    float p[4];
    for (int q = 0; q < 4; q++) {
        float s = 0.0f;
        for (int j = 0; j < 8; j++) s = fmaf(a[8 * q + j], b[8 * q + j], s);
        p[q] = s;
    }
    return (p[0] + p[1]) + (p[2] + p[3]);
}

/* F.normalize(v, eps) on one 32-vector (modules/normalize.py:7-8): v / max(|v|, eps) */
static inline void normalize32(const float *v, float eps, float *out) {
    float den = fmaxf(sqrtf(dotQ(v, v)), eps);
    for (int d = 0; d < D; d++) out[d] = v[d] / den;
}

void orc_normalize_rows32(const float *v, int64_t M, float eps, float *out) {
    for (int64_t m = 0; m < M; m++) normalize32(v + m * D, eps, out + m * D);
}

/* effective codebook + |c|^2 (quantize.py:106,111): rows normalised iff normalize */
void orc_codebook_prepare(const float *E, int64_t K, int normalize, float *cb, float *cc) {
    for (int64_t k = 0; k < K; k++) {
        if (normalize) normalize32(E + k * D, 1e-12f, cb + k * D);
        else memcpy(cb + k * D, E + k * D, sizeof(float) * D);
        cc[k] = dotQ(cb + k * D, cb + k * D);
    }
}

static const int PERM_J[8] = {0, 1, 2, 3, 4, 5, 6, 7};

/* One item through L levels.  mode: 2 STE, 3 ROTATION; training 0 => eval (o = e).
 * cbs: L effective codebooks [K][32]; ccs: L vectors [K].  Outputs (any may be NULL):
 * ids[L] (int64), emb[L][32], res[L][32] (level inputs), loss (sum over levels). */
static void rq_item(const float *z, int L, int64_t K, const float *const *cbs, const float *const *ccs,
                    int mode, int training, float beta, int64_t *ids, float *emb, float *res, float *loss) {
    float r[D], o[D];
    memcpy(r, z, sizeof(r));
    float lsum = 0.0f;
    for (int i = 0; i < L; i++) {
        const float *cb = cbs[i], *cc = ccs[i];
        if (res) memcpy(res + i * D, r, sizeof(r));
        const float xx = dotQ(r, r);
        float best = INFINITY;
        int64_t bi = 0;
        for (int64_t k = 0; k < K; k++) {
            const float *c = cb + k * D;
            float acc = 0.0f; /* ORDER-P */
            for (int j = 0; j < 8; j++)
                for (int q = 0; q < 4; q++) acc = fmaf(c[8 * q + PERM_J[j]], r[8 * q + PERM_J[j]], acc);
            const float t = xx + cc[k];
            const float dist = fmaf(-2.0f, acc, t);
            if (dist < best || k == 0) { best = dist; bi = k; } /* strict <: first minimum wins */
        }
        const float *e = cb + bi * D;
        if (!training) {
            memcpy(o, e, sizeof(o));
        } else if (mode == 2) {
            for (int d = 0; d < D; d++) o[d] = r[d] + (e[d] - r[d]);
        } else {
            float u[D], q[D], s[D], w[D];
            /* x / (|x| + eps) evaluated as x * (1 / (|x| + eps)): one division per vector (<= 1 ulp from the quotient) */
            const float inr = 1.0f / (sqrtf(xx) + 1e-8f);
            const float ine = 1.0f / (sqrtf(cc[bi]) + 1e-8f);
            for (int d = 0; d < D; d++) { u[d] = r[d] * inr; q[d] = e[d] * ine; s[d] = u[d] + q[d]; }
            const float inw = 1.0f / fmaxf(sqrtf(dotQ(s, s)), 1e-6f);
            for (int d = 0; d < D; d++) w[d] = s[d] * inw;
            const float rw = dotQ(r, w), ru = dotQ(r, u);
            for (int d = 0; d < D; d++) o[d] = (r[d] - 2.0f * (rw * w[d])) + 2.0f * (ru * q[d]);
        }
        float df[D];
        for (int d = 0; d < D; d++) df[d] = r[d] - e[d];
        const float l1 = dotQ(df, df);
        lsum = lsum + (l1 + beta * l1);
        if (ids) ids[i] = bi;
        if (emb) memcpy(emb + i * D, o, sizeof(o));
        for (int d = 0; d < D; d++) r[d] = r[d] - o[d];
    }
    if (loss) *loss = lsum;
}

/* Batch driver.  y: [B][32] encoder output BEFORE normalisation; z_out gets the level-0 input.
 * emb_cat: [B][L*32]; emb_sum: [B][32] = ((o0+o1)+o2)+...; res_cat: [B][L*32]. */
void orc_rq_forward(const float *y, int64_t B, int normalize_input, int L, int64_t K,
                    const float *const *cbs, const float *const *ccs, int mode, int training, float beta,
                    float *z_out, int64_t *ids, float *emb_cat, float *emb_sum, float *res_cat, float *loss) {
    float *emb = (float *)malloc(sizeof(float) * (size_t)L * D);
    for (int64_t b = 0; b < B; b++) {
        float z[D];
        if (normalize_input) normalize32(y + b * D, 1e-12f, z);
        else memcpy(z, y + b * D, sizeof(z));
        if (z_out) memcpy(z_out + b * D, z, sizeof(z));
        rq_item(z, L, K, cbs, ccs, mode, training, beta, ids ? ids + b * L : NULL, emb,
                res_cat ? res_cat + b * L * D : NULL, loss ? loss + b : NULL);
        if (emb_cat) memcpy(emb_cat + b * L * D, emb, sizeof(float) * (size_t)L * D);
        if (emb_sum) {
            for (int d = 0; d < D; d++) {
                float s = emb[d];
                for (int i = 1; i < L; i++) s = s + emb[i * D + d];
                emb_sum[b * D + d] = s;
            }
        }
    }
    free(emb);
}

/* encoder: weights w[j] of shape [dims[j+1]][dims[j]], SiLU between layers (encoder.py:26-31) */
void orc_mlp(const float *x, int64_t B, int n_layers, const int64_t *dims, const float *const *w, float *out) {
    const float *cur = x;
    float *buf = NULL;
    for (int j = 0; j < n_layers; j++) {
        float *nxt = (j == n_layers - 1) ? out : (float *)malloc(sizeof(float) * (size_t)B * (size_t)dims[j + 1]);
        orc_linear(cur, B, dims[j], w[j], dims[j + 1], nxt, j != n_layers - 1);
        free(buf);
        buf = (j == n_layers - 1) ? NULL : nxt;
        cur = nxt;
    }
}

/* ---- ORDER-GEN: the same level loop at any embedding width D <= 64 (a multiple of 4) -- modules/rqvae.py:37-88 with
 * configs/rqvae_ml32m.gin:11 (embed_dim 64), modules/h_rqvae.py:231-256.  Mirrors csrc/rq_generic.hip operation for operation. */
static inline float wave_sum64(const float *term, int Dg) { /* xor butterfly over 64 lanes, lanes >= Dg hold 0 */
    float v[64];
    for (int l = 0; l < 64; l++) v[l] = l < Dg ? term[l] : 0.0f;
    for (int o = 32; o > 0; o >>= 1) {
        float t[64];
        for (int l = 0; l < 64; l++) t[l] = v[l] + v[l ^ o];
        memcpy(v, t, sizeof(v));
    }
    return v[0];
}
static inline float dotG(const float *a, const float *b, int Dg) {
    float t[64];
    for (int d = 0; d < Dg; d++) t[d] = a[d] * b[d];
    return wave_sum64(t, Dg);
}

void orc_codebook_prepare_gen(const float *E, int64_t K, int Dg, int normalize, float *cb, float *cc) {
    for (int64_t k = 0; k < K; k++) {
        const float *e = E + k * Dg;
        float *c = cb + k * Dg;
        if (normalize) {
            const float den = fmaxf(sqrtf(dotG(e, e, Dg)), 1e-12f);
            for (int d = 0; d < Dg; d++) c[d] = e[d] / den;
        } else memcpy(c, e, sizeof(float) * (size_t)Dg);
        cc[k] = dotG(c, c, Dg);
    }
}

void orc_rq_forward_gen(const float *y, int64_t B, int Dg, int normalize_input, int L, int64_t K, const float *const *cbs,
                        const float *const *ccs, int mode, int training, float beta, float *z_out, int64_t *ids, float *emb_cat,
                        float *emb_sum, float *res_cat, float *loss, int cosine) {
    /* cosine != 0: QuantizeDistance.COSINE (modules/quantize.py:115-119): codes ranked by -((r/|r|) . c) / |c|, no epsilons */
    for (int64_t b = 0; b < B; b++) {
        float r[64], o[64], esum[64], rn[64];
        memcpy(r, y + b * Dg, sizeof(float) * (size_t)Dg);
        if (normalize_input) {
            const float den = fmaxf(sqrtf(dotG(r, r, Dg)), 1e-12f);
            for (int d = 0; d < Dg; d++) r[d] = r[d] / den;
        }
        if (z_out) memcpy(z_out + b * Dg, r, sizeof(float) * (size_t)Dg);
        float lsum = 0.0f;
        for (int i = 0; i < L; i++) {
            const float *cb = cbs[i], *cc = ccs[i];
            if (res_cat) memcpy(res_cat + (b * L + i) * Dg, r, sizeof(float) * (size_t)Dg);
            const float xx = dotG(r, r, Dg);
            for (int d = 0; d < Dg; d++) rn[d] = cosine ? r[d] / sqrtf(xx) : r[d];
            float best = INFINITY;
            int64_t bi = 0;
            for (int64_t k = 0; k < K; k++) { /* (lane order on the GPU: ascending k inside a lane, lowest index on ties across lanes) */
                const float *c = cb + k * Dg;
                float acc = 0.0f;
                for (int d = 0; d < Dg; d++) acc = fmaf(rn[d], c[d], acc);
                const float dist = cosine ? -(acc / sqrtf(cc[k])) : fmaf(-2.0f, acc, xx + cc[k]);
                if (dist < best) { best = dist; bi = k; }
            }
            const float *e = cb + bi * Dg;
            if (!training) memcpy(o, e, sizeof(float) * (size_t)Dg);
            else if (mode == 2) { for (int d = 0; d < Dg; d++) o[d] = r[d] + (e[d] - r[d]); }
            else {
                float u[64], q[64], s[64], w[64];
                const float inr = 1.0f / (sqrtf(xx) + 1e-8f), ine = 1.0f / (sqrtf(cc[bi]) + 1e-8f);
                for (int d = 0; d < Dg; d++) { u[d] = r[d] * inr; q[d] = e[d] * ine; s[d] = u[d] + q[d]; }
                const float inw = 1.0f / fmaxf(sqrtf(dotG(s, s, Dg)), 1e-6f);
                for (int d = 0; d < Dg; d++) w[d] = s[d] * inw;
                const float rw = dotG(r, w, Dg), ru = dotG(r, u, Dg);
                for (int d = 0; d < Dg; d++) o[d] = (r[d] - 2.0f * (rw * w[d])) + 2.0f * (ru * q[d]);
            }
            float df[64];
            for (int d = 0; d < Dg; d++) df[d] = r[d] - e[d];
            const float l1 = dotG(df, df, Dg);
            lsum = lsum + (l1 + beta * l1);
            if (ids) ids[b * L + i] = bi;
            if (emb_cat) memcpy(emb_cat + (b * L + i) * Dg, o, sizeof(float) * (size_t)Dg);
            for (int d = 0; d < Dg; d++) { esum[d] = i == 0 ? o[d] : esum[d] + o[d]; r[d] = r[d] - o[d]; }
        }
        if (emb_sum) memcpy(emb_sum + b * Dg, esum, sizeof(float) * (size_t)Dg);
        if (loss) loss[b] = lsum;
    }
}
