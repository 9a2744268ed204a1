// TagPredictor forward behind its attention gate (reference modules/h_rqvae.py:132-188: feature_extractor, two residual blocks,
// classifier) as ONE row-local launch for levels whose layers are at most PRED_WMAX wide.
//
// Why: at B = 1024 a narrow level's predictor is 14 launches of 0.03-0.24 GFLOP each (Linear, LayerNorm, Linear, ...), every one at
// the latency floor, on the lane of the step that is the longest -- and nothing in it crosses rows: Linear, LayerNorm,
// ReLU, Dropout and the residual adds are all per item.  A workgroup takes 16 items through all units: activations stay in LDS as MFMA
// B-fragment images ([k / 16][64 lanes] float4, lane = (item, k-quad): bottleneck_fwd_kernel's layout, rq.hip), weights stream from
// L2 as A fragments, eight waves share a unit's output tiles.  Every tensor the separate launches save for the backward (a Linear's
// output, a LayerNorm's input / mean / rstd / output) is written exactly as they would write it, so the backward is unchanged
// (tagpath.py hands these tensors to the same autograd Functions instead of letting them launch).
//
// Arithmetic: Linear = fp32 MFMA 16x16x4 over ascending k + bias; LayerNorm = two-pass (mean, then sum of squared deviations) with
// the row's partial sums added lane-quad -> wave -> workgroup in a fixed order; dropout = the counter-based generator of common.h on
// element index row * N + col of the unit's site (the same decision the separate launches take).  Not bit-identical to the separate
// LayerNorm kernels (another summation order): compared with them to 1e-5 in tests/test_modules_gpu.py.
#include "common.h"

namespace {

constexpr int PRED_WMAX = 256;   // widest layer (features per item kept in LDS)
constexpr int PRED_WAVES = 8;
constexpr int PRED_MAX_UNITS = 10;
constexpr int PRED_IMG = (PRED_WMAX / 16) * 64;  // float4 per activation image

struct PredUnit {
    const float *W, *bias;        // [N, K] row-major, [N]
    const float *gamma, *beta;    // LayerNorm affine or nullptr (no LayerNorm in this unit)
    int N, K;
    int act1;                     // ReLU -> Dropout(site1) right after the Linear (before a LayerNorm, or as the unit's end)
    int act2;                     // ReLU -> Dropout(site2) after the LayerNorm
    int residual;                 // + the carried residual after the LayerNorm; the result becomes the new carried residual
    int carry;                    // this unit's output becomes the carried residual (feature_extractor)
    unsigned site1, thr1, site2, thr2;
    float scale1, scale2, eps;
    float *lin;                   // [B, N] Linear output (+ bias), after act1 if act1: what LinearFn / LayerNormFn save as their x
    float *y;                     // [B, N] output of the unit (== lin when there is no LayerNorm: then nullptr)
    float *mean, *rstd;           // [B]
    float *g_lin;                 // backward: [B, N] gradient at the Linear's output (what its weight / bias gradients are formed from)
    float *partials;              // backward: [ceil(B / 4)][2][N] LayerNorm affine-gradient partials (layout of hidvae_layernorm_bwd_partial)
};

struct PredArgs {
    const float *h;               // [B, K0] input (the gate's output), row stride ldh
    int64_t B, ldh;
    int n;
    const unsigned long long *rng;  // generator state or nullptr (no dropout anywhere)
    PredUnit u[PRED_MAX_UNITS];
};

__device__ __forceinline__ float4 ld4_guard(const float *row, int k, int K, bool row_ok) {
    // four consecutive k of a weight row; elements past K (or a row past N) read as zero.  Rows are only dword-aligned (K = 230, 115)
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!row_ok) return v;
    if (k + 3 < K) {
        v.x = row[k]; v.y = row[k + 1]; v.z = row[k + 2]; v.w = row[k + 3];
    } else {
        if (k < K) v.x = row[k];
        if (k + 1 < K) v.y = row[k + 1];
        if (k + 2 < K) v.z = row[k + 2];
    }
    return v;
}

// four consecutive features of one item: one sixteen-byte buffer store where all four exist (rows are only dword-aligned: N = 230;
// the buffer form takes that), element-wise at the ragged end of a row
__device__ __forceinline__ void pred_store4(float *base, int64_t item, int N, int col, const float (&v)[4], bool in_range, int64_t B) {
    if (!in_range || col >= N) return;
    if (col + 3 < N) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(4 * B * N), 0x00020000);
        const f32x4 w = {v[0], v[1], v[2], v[3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, w), r, (int)(4 * (item * N + col)), 0, 0);
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (col + e < N) base[item * N + col + e] = v[e];
    }
}

__global__ __launch_bounds__(64 * PRED_WAVES) void predictor_fwd_kernel(PredArgs a) {
    __shared__ float4 img[3][PRED_IMG];                 // activation images: input / output of the running unit, the carried residual
    __shared__ float red[2][PRED_WAVES][16];            // per-wave row partials (sum, then squared deviations)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (a scalar: what depends on it branches on the scalar unit)
    const int it = lane & 15, q = lane >> 4;
    const int64_t item = (int64_t)blockIdx.x * 16 + it;
    const bool in_range = item < a.B;
    const int64_t src = in_range ? item : a.B - 1;
    // the input rows as k-block images
    {
        const int K0 = a.u[0].K, nkb = (K0 + 15) / 16;
        for (int idx = threadIdx.x; idx < nkb * 64; idx += 64 * PRED_WAVES) {
            const int kb = idx >> 6, l = idx & 63;
            const int64_t row = (int64_t)blockIdx.x * 16 + (l & 15);
            const float *p = a.h + (row < a.B ? row : a.B - 1) * a.ldh;
            img[0][idx] = ld4_guard(p, 16 * kb + 4 * (l >> 4), K0, true);
        }
    }
    __syncthreads();
    int cur = 0, res = 2;  // img[cur]: the unit's input; img[cur ^ 1]: its output; img[res]: the carried residual
    for (int ui = 0; ui < a.n; ui++) {
        const PredUnit u = a.u[ui];  // (by value: one batch of scalar loads per unit, not one per use of a run-time-indexed argument field)
        const int N = u.N, K = u.K, ntile = (N + 15) / 16, nkb = (K + 15) / 16;
        const float4 *Hin = img[cur];
        float4 *Hout = img[cur ^ 1];
        // ---- Linear: output tiles dealt to the waves round-robin (at most 2 per wave at N <= 256).  A wave's weight rows for BOTH its
        // tiles are fetched up front (<= 32 sixteen-byte buffer loads in flight: one round trip to L2 per unit instead of one per
        // k-block -- the first version of this kernel, 134 us); rows are only dword-aligned (K = 230, 115), the tail of a row's last
        // load reads into the next row (or, past the end of W, zeros): those k meet activation entries that are exactly zero
        float4 val[2];
        float psum = 0.0f;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(u.W), 0, (int)(4 * (int64_t)N * K), 0x00020000);
        f32x4 wv[2][PRED_WMAX / 16];
        const int t0 = wave, t1 = wave + PRED_WAVES;
        const bool has0 = t0 < ntile, has1 = t1 < ntile;
        {
            const int r0 = 16 * t0 + it, r1 = 16 * t1 + it;
            const int o0 = 4 * ((r0 < N ? r0 : 0) * K + 4 * q), o1 = 4 * ((r1 < N ? r1 : 0) * K + 4 * q);
            // (every load is issued, unconditionally and in order -- one that is not needed is aimed out of the buffer and returns zeros:
            //  with the loads behind branches the compiler waited for ALL of them before the first MFMA)
#pragma unroll
            for (int kb = 0; kb < PRED_WMAX / 16; kb++) {
                wv[0][kb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (kb < nkb && has0) ? o0 + 64 * kb : 0x7FFFFFF0, 0, 0));
                wv[1][kb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (kb < nkb && has1) ? o1 + 64 * kb : 0x7FFFFFF0, 0, 0));
            }
        }
        // the unit's per-feature vectors for this lane's columns, fetched NOW (behind the weights, in front of the MFMAs): read at their
        // point of use they were three more dependent round trips per unit
        f32x4 bia[2], gam[2], bet[2];
        {
            const int nb = 4 * N;
            const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(u.bias != nullptr ? u.bias : u.W), 0, u.bias != nullptr ? nb : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rg_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(u.gamma != nullptr ? u.gamma : u.W), 0, u.gamma != nullptr ? nb : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t re_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(u.gamma != nullptr ? u.beta : u.W), 0, u.gamma != nullptr ? nb : 0, 0x00020000);
#pragma unroll
            for (int s_ = 0; s_ < 2; s_++) {
                const int off = 4 * (16 * (wave + PRED_WAVES * s_) + 4 * q);  // (columns past N, or a vector that is absent: zeros)
                bia[s_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb_, off, 0, 0));
                gam[s_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg_, off, 0, 0));
                bet[s_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(re_, off, 0, 0));
            }
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < PRED_WMAX / 16; kb++) {
            if (kb < nkb) {
                const float4 b = Hin[kb * 64 + lane];
                if (has0) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0][kb][0], b.x, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0][kb][1], b.y, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0][kb][2], b.z, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0][kb][3], b.w, acc0, 0, 0, 0);
                }
                if (has1) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1][kb][0], b.x, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1][kb][1], b.y, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1][kb][2], b.z, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1][kb][3], b.w, acc1, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int t = wave + PRED_WAVES * s;
            val[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t >= ntile) continue;
            const f32x4 acc = s ? acc1 : acc0;
            // lane (it, q) holds features 16 t + 4 q + e of item `it`
            float v[4] = {acc[0], acc[1], acc[2], acc[3]};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int col = 16 * t + 4 * q + e;
                float x = col < N ? v[e] + bia[s][e] : 0.0f;
                if (u.act1 && col < N) {
                    x = fmaxf(x, 0.0f);
                    if (u.thr1 != 0u) {
                        const HvDrop d{a.rng, u.site1, u.thr1};
                        x = hv_drop_keep(d, (unsigned long long)(item * N + col)) ? x * u.scale1 : 0.0f;
                    }
                }
                v[e] = x;
                psum += x;
            }
            val[s] = make_float4(v[0], v[1], v[2], v[3]);
            pred_store4(u.lin, item, N, 16 * t + 4 * q, v, in_range, (int64_t)a.B);
        }
        if (u.gamma != nullptr) {
            // ---- LayerNorm over the item's N features: quad lanes -> wave -> workgroup, fixed order
            psum += __shfl_xor(psum, 16);
            psum += __shfl_xor(psum, 32);
            if (q == 0) red[0][wave][it] = psum;
            __syncthreads();
            float tot = 0.0f;
#pragma unroll
            for (int w = 0; w < PRED_WAVES; w++) tot += red[0][w][it];
            const float mu = tot / (float)N;
            float pdev = 0.0f;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int t = wave + PRED_WAVES * s;
                if (t >= ntile) continue;
                const float v[4] = {val[s].x, val[s].y, val[s].z, val[s].w};
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (16 * t + 4 * q + e < N) { const float d = v[e] - mu; pdev += d * d; }
            }
            pdev += __shfl_xor(pdev, 16);
            pdev += __shfl_xor(pdev, 32);
            if (q == 0) red[1][wave][it] = pdev;
            __syncthreads();
            float dev = 0.0f;
#pragma unroll
            for (int w = 0; w < PRED_WAVES; w++) dev += red[1][w][it];
            const float rs = 1.0f / sqrtf(dev / (float)N + u.eps);
            if (in_range && wave == 0 && q == 0) { u.mean[item] = mu; u.rstd[item] = rs; }
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int t = wave + PRED_WAVES * s;
                if (t >= ntile) continue;
                float v[4] = {val[s].x, val[s].y, val[s].z, val[s].w};
                const float4 r4 = u.residual ? img[res][t * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
                const float rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int col = 16 * t + 4 * q + e;
                    float o = 0.0f;
                    if (col < N) {
                        o = (v[e] - mu) * rs * gam[s][e] + bet[s][e];
                        if (u.act2) {
                            o = fmaxf(o, 0.0f);
                            if (u.thr2 != 0u) {
                                const HvDrop d{a.rng, u.site2, u.thr2};
                                o = hv_drop_keep(d, (unsigned long long)(item * N + col)) ? o * u.scale2 : 0.0f;
                            }
                        }
                        if (u.residual) o = o + rr[e];
                    }
                    v[e] = o;
                }
                val[s] = make_float4(v[0], v[1], v[2], v[3]);
                pred_store4(u.y, item, N, 16 * t + 4 * q, v, in_range, (int64_t)a.B);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int t = wave + PRED_WAVES * s;
            if (t >= ntile) continue;
            Hout[t * 64 + lane] = val[s];
            if (u.residual || u.carry) img[res][t * 64 + lane] = val[s];
        }
        // (tiles past ntile of the NEXT unit's k range must read as zero: a narrower output leaves stale blocks behind)
        const int nkb_next = ui + 1 < a.n ? (a.u[ui + 1].K + 15) / 16 : 0;
        for (int t = ntile + wave; t < nkb_next; t += PRED_WAVES) Hout[t * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        cur ^= 1;
    }
}

// ---- backward: the input-gradient chain of the same units in reverse, one launch.  Per unit, with G the gradient of its output (from
// the next unit's dX) and GR the gradient carried along the residual path:  g = G (+ GR at a residual / carry unit; a residual unit
// hands g on as the new GR);  through ReLU -> Dropout(2) read off the saved output;  through the LayerNorm (two row sums; the affine
// gradients' per-4-row partials go out in hidvae_layernorm_bwd_partial's layout);  through ReLU -> Dropout(1) read off the saved
// Linear output;  g_lin = that, stored for the weight / bias gradients (one grouped launch behind this one);  dX = g_lin W on MFMA with
// W read down its columns (four dword loads where the forward has one sixteen-byte load).
__global__ __launch_bounds__(64 * PRED_WAVES) void predictor_bwd_kernel(PredArgs a, const float *g_out, int64_t ldg, float *g_h, int64_t ldgh) {
    __shared__ float4 img[3][PRED_IMG];       // gradient images: of the running unit's output / input, and the carried residual's
    __shared__ float red[2][PRED_WAVES][16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int it = lane & 15, q = lane >> 4;
    const int64_t item = (int64_t)blockIdx.x * 16 + it;
    const bool in_range = item < a.B;
    const int64_t chunks = (a.B + 3) / 4;
    {
        const int NL = a.u[a.n - 1].N, nb = (NL + 15) / 16;
        for (int idx = threadIdx.x; idx < PRED_IMG; idx += 64 * PRED_WAVES) {
            const int kb = idx >> 6, l = idx & 63;
            const int64_t row = (int64_t)blockIdx.x * 16 + (l & 15);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kb < nb && row < a.B) v = ld4_guard(g_out + row * ldg, 16 * kb + 4 * (l >> 4), NL, true);
            img[0][idx] = v;
            img[2][idx] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    int cur = 0;
    for (int ui = a.n - 1; ui >= 0; ui--) {
        const PredUnit u = a.u[ui];
        const int N = u.N, K = u.K, ntn = (N + 15) / 16, ntk = (K + 15) / 16;
        float4 *G = img[cur];
        float4 *Gp = img[cur ^ 1];
        const bool has_ln = u.gamma != nullptr;
        const int nbytes = 4 * (int)a.B * N;
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(u.lin, 0, nbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(has_ln ? u.y : u.lin, 0, nbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(has_ln ? u.gamma : u.W), 0, has_ln ? 4 * N : 0, 0x00020000);
        // ---- phase A: everything elementwise / per row on the unit's N features
        float g[2][4], xh[2][4], gy[2][4];
        unsigned posmask[2] = {0u, 0u};  // bit e: the saved Linear output of column col0 + e is > 0 (the ReLU -> Dropout(1) gate)
        float s1 = 0.0f, s2 = 0.0f;
        float mu = 0.0f, rs = 0.0f;
        if (has_ln) { mu = u.mean[in_range ? item : a.B - 1]; rs = u.rstd[in_range ? item : a.B - 1]; }
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int t = wave + PRED_WAVES * s;
#pragma unroll
            for (int e = 0; e < 4; e++) { g[s][e] = 0.0f; xh[s][e] = 0.0f; gy[s][e] = 0.0f; }
            if (t >= ntn) continue;
            const int col0 = 16 * t + 4 * q;
            const int off = in_range ? 4 * ((int)item * N + col0) : 0x7FFFFFF0;
            float4 gv = G[t * 64 + lane];
            if (u.residual || u.carry) {
                const float4 r4 = img[2][t * 64 + lane];
                gv.x += r4.x; gv.y += r4.y; gv.z += r4.z; gv.w += r4.w;
            }
            if (u.residual) img[2][t * 64 + lane] = gv;  // (own slot: read above by this lane only)
            const f32x4 l4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rl, off, 0, 0));
            const f32x4 y4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, off, 0, 0));
            const f32x4 ga = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, 4 * col0, 0, 0));
            const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const bool ok = col0 + e < N;
                float v = ok ? gg[e] : 0.0f;
                if (u.act2) v = (ok && y4[e] > 0.0f) ? v * u.scale2 : 0.0f;
                gy[s][e] = v;
                if (ok && l4[e] > 0.0f) posmask[s] |= 1u << e;
                if (has_ln) {
                    const float x = ok ? (l4[e] - mu) * rs : 0.0f;
                    const float dy = v * ga[e];
                    xh[s][e] = x;
                    g[s][e] = dy;
                    s1 += dy;
                    s2 += dy * x;
                } else {
                    g[s][e] = (u.act1 && !((posmask[s] >> e) & 1u)) ? 0.0f : (u.act1 ? v * u.scale1 : v);
                }
            }
        }
        if (has_ln) {
            s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
            if (q == 0) { red[0][wave][it] = s1; red[1][wave][it] = s2; }
            __syncthreads();
            float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
            for (int w = 0; w < PRED_WAVES; w++) { t1 += red[0][w][it]; t2 += red[1][w][it]; }
            t1 = t1 / (float)N;
            t2 = t2 / (float)N;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int t = wave + PRED_WAVES * s;
                if (t >= ntn) continue;
                const int col0 = 16 * t + 4 * q;
                float pg[4], pb[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    pg[e] = gy[s][e] * xh[s][e];  // d gamma, d beta terms of this row
                    pb[e] = gy[s][e];
                    float gx = col0 + e < N ? rs * ((g[s][e] - t1) - xh[s][e] * t2) : 0.0f;
                    if (u.act1) gx = ((posmask[s] >> e) & 1u) ? gx * u.scale1 : 0.0f;
                    g[s][e] = in_range ? gx : 0.0f;
                    // the four rows of a chunk: lanes it ^ 1, it ^ 2
                    pg[e] += __shfl_xor(pg[e], 1); pg[e] += __shfl_xor(pg[e], 2);
                    pb[e] += __shfl_xor(pb[e], 1); pb[e] += __shfl_xor(pb[e], 2);
                }
                const int64_t chunk = (int64_t)blockIdx.x * 4 + (it >> 2);
                if ((it & 3) == 0 && chunk < chunks && u.partials != nullptr) {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (col0 + e < N) {
                            u.partials[(chunk * 2 + 0) * N + col0 + e] = pg[e];
                            u.partials[(chunk * 2 + 1) * N + col0 + e] = pb[e];
                        }
                }
            }
        }
        // g is now the gradient at the Linear's output: out to memory (weight / bias gradients) and into the image for the dX product
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int t = wave + PRED_WAVES * s;
            if (t >= ntn) continue;
            const float v[4] = {g[s][0], g[s][1], g[s][2], g[s][3]};
            pred_store4(u.g_lin, item, N, 16 * t + 4 * q, v, in_range, (int64_t)a.B);
            G[t * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
        // ---- phase B: dX [16 items, K] = g_lin [16, N] W [N, K]
        {
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(u.W), 0, 4 * N * K, 0x00020000);
            const int t0 = wave, t1 = wave + PRED_WAVES;
            const bool has0 = t0 < ntk, has1 = t1 < ntk;
            f32x4 wv[2][PRED_WMAX / 16];
#pragma unroll
            for (int nb = 0; nb < PRED_WMAX / 16; nb++) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int n = 16 * nb + 4 * q + e;
                    const int o0 = (nb < ntn && has0 && n < N) ? 4 * (n * K + 16 * t0 + it) : 0x7FFFFFF0;
                    const int o1 = (nb < ntn && has1 && n < N) ? 4 * (n * K + 16 * t1 + it) : 0x7FFFFFF0;
                    wv[0][nb][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, o0, 0, 0));
                    wv[1][nb][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, o1, 0, 0));
                }
            }
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nb = 0; nb < PRED_WMAX / 16; nb++) {
                if (nb < ntn) {
                    const float4 b = G[nb * 64 + lane];
                    if (has0) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0][nb][0], b.x, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0][nb][1], b.y, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0][nb][2], b.z, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0][nb][3], b.w, acc0, 0, 0, 0);
                    }
                    if (has1) {
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1][nb][0], b.x, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1][nb][1], b.y, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1][nb][2], b.z, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1][nb][3], b.w, acc1, 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int t = wave + PRED_WAVES * s;
                if (t >= ntk) continue;
                const f32x4 acc = s ? acc1 : acc0;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = 16 * t + 4 * q + e < K ? acc[e] : 0.0f;
                if (ui > 0) Gp[t * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
                else if (in_range) {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (16 * t + 4 * q + e < K) g_h[item * ldgh + 16 * t + 4 * q + e] = v[e];
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

}  // namespace

extern "C" int hidvae_predictor_fwd(const float *h, int64_t ldh, int64_t B, const hidvae_pred_unit *units, int n_units,
                                    const unsigned long long *rng_state, void *stream) {
    HV_REQUIRE(h && units && B >= 1 && n_units >= 1 && n_units <= PRED_MAX_UNITS, "predictor_fwd: bad arguments");
    HV_REQUIRE(B <= (int64_t)1 << 20, "predictor_fwd: B=%lld (byte offsets are 32-bit)", (long long)B);
    PredArgs a{};
    a.h = h; a.B = B; a.ldh = ldh; a.n = n_units; a.rng = rng_state;
    for (int i = 0; i < n_units; i++) {
        const hidvae_pred_unit &s = units[i];
        HV_REQUIRE(s.W && s.lin && s.N >= 1 && s.N <= PRED_WMAX && s.K >= 1 && s.K <= PRED_WMAX, "predictor_fwd: unit %d: widths %d x %d (at most %d)",
                   i, s.N, s.K, PRED_WMAX);
        HV_REQUIRE(i == 0 ? ldh >= s.K : s.K == units[i - 1].N, "predictor_fwd: unit %d does not continue unit %d", i, i - 1);
        HV_REQUIRE(s.gamma == nullptr || (s.beta && s.y && s.mean && s.rstd), "predictor_fwd: unit %d: LayerNorm outputs missing", i);
        HV_REQUIRE(s.gamma != nullptr || (!s.act2 && !s.residual), "predictor_fwd: unit %d: act2 / residual need the LayerNorm", i);
        HV_REQUIRE((s.drop_threshold1 == 0u && s.drop_threshold2 == 0u) || rng_state != nullptr, "predictor_fwd: unit %d: dropout needs the generator state", i);
        PredUnit &u = a.u[i];
        u.W = s.W; u.bias = s.bias; u.gamma = s.gamma; u.beta = s.beta; u.N = s.N; u.K = s.K;
        u.act1 = s.act1; u.act2 = s.act2; u.residual = s.residual; u.carry = s.carry;
        u.site1 = s.drop_site1; u.thr1 = s.drop_threshold1; u.scale1 = s.drop_scale1;
        u.site2 = s.drop_site2; u.thr2 = s.drop_threshold2; u.scale2 = s.drop_scale2;
        u.eps = s.eps; u.lin = s.lin; u.y = s.y; u.mean = s.mean; u.rstd = s.rstd;
    }
    hipLaunchKernelGGL(predictor_fwd_kernel, dim3((unsigned)hv_cdiv(B, 16)), dim3(64 * PRED_WAVES), 0, (hipStream_t)stream, a);
    HV_LAUNCH_CHECK("predictor_fwd");
    return HIDVAE_OK;
}

extern "C" int hidvae_predictor_bwd(const float *g_out, int64_t ldg, int64_t B, const hidvae_pred_unit *units, int n_units, float *g_h,
                                    int64_t ldgh, void *stream) {
    HV_REQUIRE(g_out && units && g_h && B >= 1 && n_units >= 1 && n_units <= PRED_MAX_UNITS, "predictor_bwd: bad arguments");
    HV_REQUIRE(B <= (int64_t)1 << 20, "predictor_bwd: B=%lld (byte offsets are 32-bit)", (long long)B);
    PredArgs a{};
    a.B = B; a.n = n_units;
    for (int i = 0; i < n_units; i++) {
        const hidvae_pred_unit &s = units[i];
        HV_REQUIRE(s.W && s.lin && s.g_lin && s.N >= 1 && s.N <= PRED_WMAX && s.K >= 1 && s.K <= PRED_WMAX, "predictor_bwd: unit %d is malformed", i);
        HV_REQUIRE(i == 0 || s.K == units[i - 1].N, "predictor_bwd: unit %d does not continue unit %d", i, i - 1);
        HV_REQUIRE(s.gamma == nullptr || (s.y && s.mean && s.rstd && s.partials), "predictor_bwd: unit %d: LayerNorm tensors missing", i);
        HV_REQUIRE(s.gamma != nullptr || (!s.act2 && !s.residual), "predictor_bwd: unit %d: act2 / residual need the LayerNorm", i);
        PredUnit &u = a.u[i];
        u.W = s.W; u.gamma = s.gamma; u.N = s.N; u.K = s.K;
        u.act1 = s.act1; u.act2 = s.act2; u.residual = s.residual; u.carry = s.carry;
        u.scale1 = s.drop_scale1; u.scale2 = s.drop_scale2;
        u.lin = s.lin; u.y = s.y; u.mean = s.mean; u.rstd = s.rstd; u.g_lin = s.g_lin; u.partials = s.partials;
    }
    HV_REQUIRE(ldg >= units[n_units - 1].N && ldgh >= units[0].K, "predictor_bwd: leading dimension too small");
    hipLaunchKernelGGL(predictor_bwd_kernel, dim3((unsigned)hv_cdiv(B, 16)), dim3(64 * PRED_WAVES), 0, (hipStream_t)stream, a, g_out, ldg, g_h, ldgh);
    HV_LAUNCH_CHECK("predictor_bwd");
    return HIDVAE_OK;
}

