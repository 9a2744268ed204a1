// Fused L-level residual quantisation for gfx950 (CDNA4), embed_dim = 32.
//
// Replaces the per-level op sequence of the reference (modules/quantize.py:100-153 called from
// modules/h_rqvae.py:515-552): distance GEMM -> argmin -> gather -> rotation trick / STE -> QuantizeLoss
// -> residual subtract, for all L levels in ONE kernel, residual kept in registers, codebooks staged in LDS.
//
// Geometry (wave = 64 lanes): a wave owns 16 items; lane = (it = lane & 15, q = lane >> 4) holds the 8
// contiguous components d in [8q, 8q+8) of item `it`.  That one layout is
//   * the B operand of v_mfma_f32_16x16x4_f32 (B[k = lane>>4][col = lane&15]) with the codes as A rows, so the
//     16x16 accumulator puts item `it` on the lane and 4 codes in its 4 registers: the argmin is in-lane over
//     codes and needs only two cross-lane steps (xor 16, xor 32);
//   * two float4 global loads/stores per lane per row (coalesced 128-B rows);
//   * the partial-sum layout of every 32-wide reduction (four chains combined as (p0+p1)+(p2+p3)).
// fp32 MFMA is a k-ordered fmaf chain, so distances are reproducible bit-for-bit by oracle/exact.c, which is
// how the semantic ids are proven bit-exact.  Exact fp32 makes this kernel MFMA-bound, not HBM-bound
// (2*K*32 FLOP per item-level at the vector rate) -- see DESIGN.md for the roofline.
#include <math.h>
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int D = 32;  // the width these kernels are built around (4 quarter-lanes x 8 components); other widths: rq_generic.hip
constexpr int WG_THREADS = 256;
constexpr int ITEMS_PER_WAVE = 16;
constexpr int ITEMS_PER_WG = 64;
constexpr int MAX_KC = 1024;  // codes staged in LDS at a time (32*(1024+2)+1024 floats = 135.3 KB)

__device__ __forceinline__ float sumQ(float p) {
    // (p0+p1)+(p2+p3) over the item's four quarter-lanes.  v_permlane16_swap / v_permlane32_swap (gfx950) hand both partners' values to
    // both lanes, so each step is one vector-ALU exchange + one add instead of a ds_bpermute round trip through the LDS crossbar
    // (~120 cycles each, two per reduction, six reductions per level): same sums bit for bit (a+b on one lane, b+a on the other).
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(p), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// the value of lane ^ 16 / lane ^ 32 (one vector-ALU exchange + a select; see sumQ)
__device__ __forceinline__ unsigned xchg16(unsigned v) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (threadIdx.x & 16) ? r[0] : r[1];
}
__device__ __forceinline__ unsigned xchg32(unsigned v) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (threadIdx.x & 32) ? r[0] : r[1];
}
__device__ __forceinline__ float dotQ(const float (&a)[8], const float (&b)[8]) {
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) s = fmaf(a[j], b[j], s);
    return sumQ(s);
}
__device__ __forceinline__ void load8(const float *p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4 *>(p);
    const float4 b = *reinterpret_cast<const float4 *>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(float *p, const float (&v)[8]) {
    *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4 *>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// o_i for one level (quantize.py:131-140,146-148).  r: level input, e: selected code, xx=|r|^2, cce=|e|^2.
// Also returns the rotation frame (u, qv, w) for the backward.
template <int MODE, bool TRAIN>
__device__ __forceinline__ void level_output(const float (&r)[8], const float (&e)[8], float xx, float cce,
                                             float (&o)[8], float (&u)[8], float (&qv)[8], float (&w)[8]) {
    if (!TRAIN) {
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = e[j];
    } else if (MODE == HIDVAE_MODE_STE) {
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = r[j] + (e[j] - r[j]);
    } else {
        // x / (|x| + eps) as x * (1 / (|x| + eps)): one IEEE division per vector instead of one per component (the
        // component-wise divisions were ~2/3 of this kernel's VALU work); <= 1 ulp from the reference's quotient, and
        // oracle/exact.c does the same, so the GPU/oracle comparison stays bit for bit
        const float inr = 1.0f / (sqrtf(xx) + 1e-8f);
        const float ine = 1.0f / (sqrtf(cce) + 1e-8f);
        float s[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u[j] = r[j] * inr;
            qv[j] = e[j] * ine;
            s[j] = u[j] + qv[j];
        }
        const float inw = 1.0f / fmaxf(sqrtf(dotQ(s, s)), 1e-6f);
#pragma unroll
        for (int j = 0; j < 8; j++) w[j] = s[j] * inw;
        const float rw = dotQ(r, w), ru = dotQ(r, u);
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (r[j] - 2.0f * (rw * w[j])) + 2.0f * (ru * qv[j]);
    }
}

// ------------------------------------------------------------------------------------------------
// effective codebook: optional row normalisation + |c|^2, same lane geometry (16 rows per wave)
// ------------------------------------------------------------------------------------------------
struct PrepArgs {
    const float *E[HIDVAE_MAX_LEVELS];
    int normalize[HIDVAE_MAX_LEVELS];
    int L;
    int64_t K;
    float *cb_eff;
    float *cc;
    int carry;             // one extra workgroup (the last) computes the optimizer's per-step scalars: both are tiny start-of-step
    HvAdamPrepare adam;    // launches that depend on nothing, and a launch costs more than either of them
};

__global__ __launch_bounds__(WG_THREADS) void codebook_prepare_kernel(PrepArgs a) {
    if (a.carry && blockIdx.x == gridDim.x - 1) {
        hv_adamw_prepare(a.adam);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int it = lane & 15, q = lane >> 4;
    const int64_t rows = (int64_t)a.L * a.K;
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * ITEMS_PER_WAVE + it;
    const int64_t src = row < rows ? row : rows - 1;
    const int lvl = (int)(src / a.K);
    const int64_t k = src - (int64_t)lvl * a.K;
    float v[8];
    load8(a.E[lvl] + k * D + 8 * q, v);
    if (a.normalize[lvl]) {
        const float den = fmaxf(sqrtf(dotQ(v, v)), 1e-12f);
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = v[j] / den;
    }
    const float cc = dotQ(v, v);
    if (row < rows) {
        store8(a.cb_eff + row * D + 8 * q, v);
        if (q == 0) a.cc[row] = cc;
    }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
struct FwdArgs {
    const float *y;
    int64_t B;
    int normalize_input;
    const float *cb_eff;  // [L][K][32]
    const float *cc;      // [L][K]
    int L;
    int64_t K;
    int KC;       // codes per LDS chunk (multiple of 32)
    int nchunks;  // ceil(K / KC)
    float beta;
    float *z;
    int64_t *ids;
    float *emb_cat;
    int64_t ld_cat;
    float *emb_sum;
    float *res_cat;
    float *qloss;
    // optional, fused middle launch only (hidvae_bottleneck_fwd): the debug statistics of h_rqvae.py:643-648
    float *embs_norm;             // [B, L]: |emb_out_i| per item and level, summed in id_stats_kernel's order
    unsigned long long *census;   // id-tuple census table (census_size slots + 3 control words), see bottleneck_fwd_kernel
    int64_t census_size;
    float *p_unique;
    // prefilter kernels: items whose approximate search is too close to call are appended here (and get a tentative result); the exact
    // kernel then runs over exactly those items and overwrites every output of theirs (hidvae_rq_forward, "deferred confirmation")
    int *fix_list;                // [>= B] item numbers
    unsigned *fix_count;          // number of entries (zeroed before the prefilter launch)
    // exact kernels: when `list` is set, the launch processes items list[0 .. *list_count) instead of 0 .. B
    const int *list;
    const unsigned *list_count;
};

// stage codes [c0, c0+KC) of level `lvl` into LDS, d-major: Cs[d][KC+2], then |c|^2 (padding: 0 / +inf)
__device__ __forceinline__ void stage_codes(float *Cs, const FwdArgs &a, int lvl, int c0) {
    const int LDK = a.KC + 2;
    float *ccs = Cs + 32 * LDK;
    const float *src = a.cb_eff + (int64_t)lvl * a.K * D;
    for (int idx0 = threadIdx.x; idx0 < a.KC * 8; idx0 += 4 * blockDim.x) {  // four loads in flight per thread, then their LDS writes
        float4 v[4];                                                           // (one load per loop trip paid the L2 latency each time)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int idx = idx0 + j * blockDim.x;
            const int64_t k = (int64_t)c0 + (idx >> 3);
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < a.KC * 8 && k < a.K) v[j] = *reinterpret_cast<const float4 *>(src + k * D + 4 * (idx & 7));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int idx = idx0 + j * blockDim.x;
            if (idx >= a.KC * 8) break;
            const int kl = idx >> 3, d4 = idx & 7;
            Cs[(4 * d4 + 0) * LDK + kl] = v[j].x;
            Cs[(4 * d4 + 1) * LDK + kl] = v[j].y;
            Cs[(4 * d4 + 2) * LDK + kl] = v[j].z;
            Cs[(4 * d4 + 3) * LDK + kl] = v[j].w;
        }
    }
    for (int kl = threadIdx.x; kl < a.KC; kl += blockDim.x) {
        const int64_t k = (int64_t)c0 + kl;
        ccs[kl] = k < a.K ? a.cc[(int64_t)lvl * a.K + k] : INFINITY;
    }
}

// CSPLIT (small batches): the 4 waves of a workgroup share ONE 16-item tile and each scans a quarter of the codes;
// the four (distance, index) candidates meet in LDS and are merged in ascending-quarter order (so equal distances still
// resolve to the lowest index), then every wave carries on with the same residual.  4x more workgroups, 4x shorter
// search per wave; results are bit-identical to the unsplit kernel.
// The level loop for one 16-item tile held as r[8] per lane (see the geometry note at the top): argmin per level, output,
// loss, residual update, stores of ids / emb_cat / res_cat.  r is consumed; esum and loss are returned in registers.
// STREAM (code-split, codebooks too large to stay in LDS -- 4 x 1024 x 32 fp32 is 512 KB): no LDS copy of the codes at all.  A code
// tile is the MFMA A operand, and lane (it, q) needs floats [8q, 8q+8) of code row t+it: two 16-byte loads from the row-major table,
// 2 KB per 16 codes and wave, fully coalesced, so every wave streams ITS share of the level's codes from L2 straight into MFMA
// registers (next 32 codes in flight behind the current 16 MFMAs).  16 items per workgroup, i.e. B/16 workgroups instead of B/64
// staging 131 KB per level each: at B = 4096 the launch fills 256 CUs instead of 64.  Same arithmetic, same merge order: bit-identical.
template <int MODE, bool TRAIN, bool RESIDENT, bool CSPLIT, int SW = 4, bool STREAM = false>  // SW: waves that share the codes of one 16-item tile
__device__ __forceinline__ void rq_level_loop(const FwdArgs &a, float *lds, float (*cand_d)[SW][16], int (*cand_i)[SW][16], int &phase,
                                              int wave, int it, int q, int64_t item, bool valid, float (&r)[8], float (&esum)[8],
                                              float &loss, unsigned long long &tuple) {
    tuple = 0ull;  // the item's ids, 10 bits per level (meaningful for L <= 4, K <= 1024: what the fused census needs)
    const int LDK = a.KC + 2;
    const int lvl_floats = 32 * LDK + a.KC;
    loss = 0.0f;
    for (int i = 0; i < a.L; i++) {
        if (a.res_cat != nullptr && valid) store8(a.res_cat + item * (a.L * D) + i * D + 8 * q, r);
        const float xx = dotQ(r, r);
        float best = INFINITY;
        int bidx = 0;
        if (STREAM) {
            const float *cbl = a.cb_eff + (int64_t)i * a.K * D + 8 * q;
            const float *ccl = a.cc + (int64_t)i * a.K;
            const int Kp = (int)((a.K + 31) / 32 * 32);
            const int share = ((Kp / 32 + SW - 1) / SW) * 32;  // codes per wave, a multiple of 32
            const int t_lo = wave * share, t_hi = (t_lo + share < Kp) ? t_lo + share : Kp;
            // 128 codes (4 x 32) per step, the NEXT step's 16 row loads in flight behind this step's 64 MFMAs: with one workgroup per
            // CU nothing else hides the L2 round trip (one 32-code step ahead: 46 us at 4096 x 4 x 1024; this form: see DESIGN.md)
            float cur[4][2][8], nxt[4][2][8];
            auto fetch = [&](int t, float (&x)[4][2][8]) {
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int hh = 0; hh < 2; hh++) {
                        const int k = t + 32 * u + 16 * hh + it;
                        if (t + 32 * u < t_hi && k < a.K) load8(cbl + (int64_t)k * D, x[u][hh]);
                        else {
#pragma unroll
                            for (int j = 0; j < 8; j++) x[u][hh][j] = 0.0f;
                        }
                    }
            };
            fetch(t_lo, cur);
            for (int t0 = t_lo; t0 < t_hi; t0 += 128) {
                fetch(t0 + 128, nxt);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int t = t0 + 32 * u;
                    if (t < t_hi) {
                        float cc0[4], cc1[4];
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            const int k0 = t + 4 * q + g, k1 = t + 16 + 4 * q + g;
                            cc0[g] = k0 < a.K ? ccl[k0] : INFINITY;
                            cc1[g] = k1 < a.K ? ccl[k1] : INFINITY;
                        }
                        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[u][0][j], r[j], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[u][1][j], r[j], acc1, 0, 0, 0);
                        }
#pragma unroll
                        for (int g = 0; g < 4; g++) {  // ascending code index inside the lane: strict < keeps the first min
                            const float d0 = fmaf(-2.0f, acc0[g], xx + cc0[g]);
                            if (d0 < best) { best = d0; bidx = t + 4 * q + g; }
                        }
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            const float d1 = fmaf(-2.0f, acc1[g], xx + cc1[g]);
                            if (d1 < best) { best = d1; bidx = t + 16 + 4 * q + g; }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int hh = 0; hh < 2; hh++)
#pragma unroll
                        for (int j = 0; j < 8; j++) cur[u][hh][j] = nxt[u][hh][j];
            }
        } else
        for (int c = 0; c < a.nchunks; c++) {
            const float *Cs;
            if (RESIDENT) {
                Cs = lds + i * lvl_floats;
            } else {
                __syncthreads();  // previous chunk fully consumed
                stage_codes(lds, a, i, c * a.KC);
                __syncthreads();
                Cs = lds;
            }
            const float *ccs = Cs + 32 * LDK;
            const float *arow = Cs + (8 * q) * LDK + it;
            const int cbase = c * a.KC + 4 * q;
            // two 16-code tiles per iteration: two independent accumulator chains keep the MFMA pipe full
            const int t_lo = CSPLIT ? wave * (a.KC / SW) : 0, t_hi = CSPLIT ? (wave + 1) * (a.KC / SW) : a.KC;
            for (int t = t_lo; t < t_hi; t += 32) {
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float a0 = arow[j * LDK + t];
                    const float a1 = arow[j * LDK + t + 16];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, r[j], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, r[j], acc1, 0, 0, 0);
                }
                const float4 c0 = *reinterpret_cast<const float4 *>(ccs + t + 4 * q);
                const float4 c1 = *reinterpret_cast<const float4 *>(ccs + t + 16 + 4 * q);
                const float cc0[4] = {c0.x, c0.y, c0.z, c0.w};
                const float cc1[4] = {c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                for (int g = 0; g < 4; g++) {  // ascending code index inside the lane: strict < keeps the first min
                    const float d0 = fmaf(-2.0f, acc0[g], xx + cc0[g]);
                    if (d0 < best) { best = d0; bidx = cbase + t + g; }
                }
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float d1 = fmaf(-2.0f, acc1[g], xx + cc1[g]);
                    if (d1 < best) { best = d1; bidx = cbase + t + 16 + g; }
                }
            }
        }
        // combine the 4 quarter-lanes of the item; equal distances resolve to the lowest code index
        {
            float ob = __uint_as_float(xchg16(__float_as_uint(best)));
            int oi = (int)xchg16((unsigned)bidx);
            if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
            ob = __uint_as_float(xchg32(__float_as_uint(best)));
            oi = (int)xchg32((unsigned)bidx);
            if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
        }
        if (CSPLIT) {  // merge the four code quarters (double-buffered by level parity: one barrier per level)
            const int pb = phase & 1;
            phase++;
            if (q == 0) { cand_d[pb][wave][it] = best; cand_i[pb][wave][it] = bidx; }
            __syncthreads();
            best = cand_d[pb][0][it];
            bidx = cand_i[pb][0][it];
#pragma unroll
            for (int ww = 1; ww < SW; ww++) {
                const float ob = cand_d[pb][ww][it];
                const int oi = cand_i[pb][ww][it];
                if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
            }
        }
        float e[8];
        float cce;
        if (RESIDENT) {  // the winning row is already in LDS (same floats as cb_eff): no dependent trip to L2
            const float *Cl = lds + i * lvl_floats;
#pragma unroll
            for (int j = 0; j < 8; j++) e[j] = Cl[(8 * q + j) * LDK + bidx];
            cce = Cl[32 * LDK + bidx];
        } else {
            load8(a.cb_eff + ((int64_t)i * a.K + bidx) * D + 8 * q, e);
            cce = a.cc[(int64_t)i * a.K + bidx];
        }
        float o[8], u[8], qv[8], w[8];
        level_output<MODE, TRAIN>(r, e, xx, cce, o, u, qv, w);
        float df[8];
#pragma unroll
        for (int j = 0; j < 8; j++) df[j] = r[j] - e[j];
        const float l1 = dotQ(df, df);  // loss.py:41-44: both terms are numerically |r-e|^2
        loss = loss + (l1 + a.beta * l1);
        if (valid) {
            if (q == 0) a.ids[item * a.L + i] = (int64_t)bidx;
            if (a.emb_cat != nullptr) store8(a.emb_cat + item * a.ld_cat + i * D + 8 * q, o);
        }
        tuple |= (unsigned long long)(unsigned)bidx << (10 * (i & 3));
        if (a.embs_norm != nullptr) {
            // |o|: id_stats_kernel adds ((x^2+y^2)+(z^2+w^2)) of the row's eight float4 one after the other; the lane quarter q
            // holds float4 number 2q and 2q+1, so the running sum walks q = 0, 1, 2, 3
            const float ta = (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
            const float tb = (o[4] * o[4] + o[5] * o[5]) + (o[6] * o[6] + o[7] * o[7]);
            float sn = ta + tb;                                                   // q = 0: t0 + t1
            float prev = __uint_as_float(xchg16(__float_as_uint(sn)));            // q = 1 <- q = 0
            if (q == 1) sn = (prev + ta) + tb;
            prev = __uint_as_float(xchg16(xchg32(__float_as_uint(sn))));          // q = 2 <- q = 3 <- q = 1
            if (q == 2) sn = (prev + ta) + tb;
            prev = __uint_as_float(xchg16(__float_as_uint(sn)));                  // q = 3 <- q = 2
            if (q == 3) sn = (prev + ta) + tb;
            if (valid && q == 3) a.embs_norm[item * a.L + i] = sqrtf(sn);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            esum[j] = (i == 0) ? o[j] : esum[j] + o[j];
            r[j] = r[j] - o[j];
        }
    }
}

template <int MODE, bool TRAIN, bool RESIDENT, bool CSPLIT, int NW, bool STREAM = false>
__global__ __launch_bounds__(64 * NW) void rq_forward_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float cand_d[2][4][16];
    __shared__ int cand_i[2][4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int it = lane & 15, q = lane >> 4;
    const int LDK = a.KC + 2;
    const int lvl_floats = 32 * LDK + a.KC;

    if (RESIDENT) {
        for (int i = 0; i < a.L; i++) stage_codes(lds + i * lvl_floats, a, i, 0);
        __syncthreads();
    }
    static_assert(!CSPLIT || NW == 4, "the code-split variant merges four quarter searches");
    constexpr int TILE_ITEMS = CSPLIT ? ITEMS_PER_WAVE : ITEMS_PER_WAVE * NW;
    int phase = 0;  // running level count across tiles: candidate buffers alternate, one barrier per level suffices
    const int64_t nB = a.list != nullptr ? (int64_t)*a.list_count : a.B;  // (a listed launch: the items the prefilter could not decide)
    const int64_t ntiles = (nB + TILE_ITEMS - 1) / TILE_ITEMS;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t pos = tile * TILE_ITEMS + (CSPLIT ? 0 : wave * ITEMS_PER_WAVE) + it;
        const bool in_range = pos < nB;
        const bool valid = in_range && (!CSPLIT || wave == 0);  // stores: every item once
        const int64_t spos = in_range ? pos : nB - 1;
        const int64_t item = a.list != nullptr ? (int64_t)a.list[spos] : pos;
        const int64_t src = a.list != nullptr ? item : spos;
        float r[8];
        load8(a.y + src * D + 8 * q, r);
        if (a.normalize_input) {  // F.normalize(eps=1e-12), modules/encoder.py:32
            const float den = fmaxf(sqrtf(dotQ(r, r)), 1e-12f);
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = r[j] / den;
        }
        if (a.z != nullptr && valid) store8(a.z + item * D + 8 * q, r);
        float loss;
        float esum[8];
        unsigned long long tuple;
        rq_level_loop<MODE, TRAIN, RESIDENT, CSPLIT, 4, STREAM>(a, lds, cand_d, cand_i, phase, wave, it, q, item, valid, r, esum, loss, tuple);
        if (valid) {
            if (a.emb_sum != nullptr) store8(a.emb_sum + item * D + 8 * q, esum);
            if (a.qloss != nullptr && q == 0) a.qloss[item] = loss;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Large batches: split-bf16 PREFILTER + deferred exact confirmation.  The exact search above spends 8 fp32 MFMAs (256 cycles) per
// 16 codes x 16 items; here every code row and every residual is split into two bf16 numbers (x = hi + lo + O(2^-18 |x|)) and the
// three products hi.hi + hi.lo + lo.hi run on bf16 MFMA.  The approximate score s_k = |c_k|^2 - 2 dot~ differs from the exact fp32
// distance (minus |x|^2) by less than
//     e = 2 (3 * 2^-18 + 128 * 2^-24) |x||c_k| + 2^-21 (|x|^2 + |c_k|^2)  <  2^-15 (|x|^2 + max_k |c_k|^2),
// so if the runner-up's score exceeds the best by more than DELTA = 2^-14 (|x|^2 + max|c|^2), the approximate argmin IS the
// exact argmin (the exact winner k* has s_k* <= d_k* + e <= d_j + e <= s_j + 2e for every j).  Each lane keeps the smallest and the
// second smallest score it has seen (one v_med3 on top of the running minimum).  An item with a level whose two best scores are
// closer than DELTA is UNDECIDED: the prefilter launch carries on with the approximate winner, appends the item to a list, and the
// exact kernel (rq_forward_kernel over that list, ~1 % of the items on codebook-like data) recomputes and overwrites every output of
// the listed items afterwards.  Everything after the argmin (winner row, rotation, loss, residual) is the exact fp32 code, so ids
// and every float are bit-identical to rq_forward_kernel.  (Rounds 1-2 confirmed inside the prefilter launch: a wave with one
// undecided item re-scanned the level -- 9 % of the 32-item waves, 20 % of the launch's time and 40 registers per lane.)
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned bf16_rne(float x) {  // round-to-nearest-even: v_cvt_pk_bf16_f32 on gfx950
    const __bf16 h = (__bf16)x;
    return (unsigned)__builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ void split_bf16(float x, unsigned &hi, unsigned &lo) {
    hi = bf16_rne(x);
    lo = bf16_rne(x - __uint_as_float(hi << 16));
}

// ------------------------------------------------------------------------------------------------
// The prefilter kernel (B >= 65536): v_mfma_f32_32x32x16_bf16, 32 items per wave.
//   * lane = (n = lane & 31, h = lane >> 5) holds dims [16h, 16h+16) of item n: two lanes per item, one cross-lane step per reduction
//     (p0+p1 | p2+p3 are in-lane, then one v_permlane32_swap: the same (p0+p1)+(p2+p3) as everywhere else, bit for bit);
//   * the LDS images hold bf16 hi/lo of -2c and the accumulators start from |c|^2, so a score IS the accumulator (no fma);
//   * the lane's running best carries its accumulator number j in the 4 low mantissa bits (one v_and_or per score, |error| < 2^-19 |s|,
//     inside the slack of DELTA), the 32-code tile is tracked by one compare per 16 scores: per score and_or + med3 + min;
//   * one 32x32x16 MFMA yields 1024 partial scores per 8 issue cycles.
// 64-byte code rows are stored with their four 16-byte chunks XOR-swizzled by (code >> 2) & 3: ds_read_b128 fragment reads are
// conflict-free without padding (SQ_LDS_BANK_CONFLICT 0; 33 KB per level at K = 256).  Everything after the argmin is the exact fp32
// code in this lane geometry.
// WHAT BOUNDS IT (rocprofv3 counters of the ids-only form at 1,048,576 items, 3 x 256, profiles/r03_*): the vector ALU.  A wave64
// vector instruction occupies its SIMD for 4 cycles (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.1), the launch issues 790 of them
// per wave and level -- 3 per score (index pack, med3, min: the minimum for an exact top-2 with its index) = 371, the rest the
// residual split, the merge and the tail -- i.e. 71 % of the launch's cycles; the matrix pipe is 36 % busy, LDS 25 %, and doubling
// the waves per SIMD (8 -> 16 per workgroup) buys 7 %.  HBM traffic is 152-796 B/item against a vector-ALU floor of ~85 us per
// million items at the clock the chip holds under this load (1.6 GHz): an HBM roofline fraction above ~0.2 is out of reach for an
// EXACT argmin in this formulation.
// What the launch is NOT bound by (each measured): the winner-row gather, in-order vmcnt waits behind the stores, LDS bandwidth.
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float PF_PAD_CC = 1.0e30f;  // |c|^2 of padding codes: finite (the packed index must not turn +inf into a NaN)

__device__ __forceinline__ float swap32_sum(float p) {  // p(lane) + p(lane ^ 32), the same bits on both lanes
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(p), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ unsigned swap32_other(unsigned v, int h) {  // v of lane ^ 32
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return h ? r[0] : r[1];
}
__device__ __forceinline__ float dotH(const float (&a)[16], const float (&b)[16]) {
    float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) p0 = fmaf(a[j], b[j], p0);
#pragma unroll
    for (int j = 0; j < 8; j++) p1 = fmaf(a[8 + j], b[8 + j], p1);
    return swap32_sum(p0 + p1);
}
__device__ __forceinline__ void load16(const float *p, float (&v)[16]) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float4 a = *reinterpret_cast<const float4 *>(p + 4 * k);
        v[4 * k] = a.x; v[4 * k + 1] = a.y; v[4 * k + 2] = a.z; v[4 * k + 3] = a.w;
    }
}
__device__ __forceinline__ void store16(float *p, const float (&v)[16]) {
#pragma unroll
    for (int k = 0; k < 4; k++) *reinterpret_cast<float4 *>(p + 4 * k) = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
}

template <int MODE, bool TRAIN>
__device__ __forceinline__ void level_output16(const float (&r)[16], const float (&e)[16], float xx, float cce, float (&o)[16]) {
    if (!TRAIN) {
#pragma unroll
        for (int j = 0; j < 16; j++) o[j] = e[j];
    } else if (MODE == HIDVAE_MODE_STE) {
#pragma unroll
        for (int j = 0; j < 16; j++) o[j] = r[j] + (e[j] - r[j]);
    } else {  // (the arithmetic of level_output, 16 dims per lane)
        const float inr = 1.0f / (sqrtf(xx) + 1e-8f);
        const float ine = 1.0f / (sqrtf(cce) + 1e-8f);
        float u[16], qv[16], s[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            u[j] = r[j] * inr;
            qv[j] = e[j] * ine;
            s[j] = u[j] + qv[j];
        }
        const float inw = 1.0f / fmaxf(sqrtf(dotH(s, s)), 1e-6f);
        float w[16];
#pragma unroll
        for (int j = 0; j < 16; j++) w[j] = s[j] * inw;
        const float rw = dotH(r, w), ru = dotH(r, u);
#pragma unroll
        for (int j = 0; j < 16; j++) o[j] = (r[j] - 2.0f * (rw * w[j])) + 2.0f * (ru * qv[j]);
    }
}

__host__ __device__ __forceinline__ size_t pf32_level_bytes(int KC) { return (size_t)KC * (64 + 64 + 4); }

// stage level `lvl`: bf16 hi / lo images of -2c (64-byte rows, chunks swizzled), |c|^2 (padding: PF_PAD_CC), max |c|^2
__device__ __forceinline__ void pf32_stage(char *base, const FwdArgs &a, int lvl, unsigned *ccmax_bits) {
    char *Ch = base;
    char *Cl = base + (size_t)a.KC * 64;
    float *ccs = reinterpret_cast<float *>(base + (size_t)a.KC * 128);
    const float *src = a.cb_eff + (int64_t)lvl * a.K * D;
    for (int idx = threadIdx.x; idx < a.KC * 8; idx += blockDim.x) {
        const int kl = idx >> 3, d4 = idx & 7;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kl < a.K) v = *reinterpret_cast<const float4 *>(src + (int64_t)kl * D + 4 * d4);
        unsigned h0, h1, h2, h3, l0, l1, l2, l3;
        split_bf16(-2.0f * v.x, h0, l0); split_bf16(-2.0f * v.y, h1, l1); split_bf16(-2.0f * v.z, h2, l2); split_bf16(-2.0f * v.w, h3, l3);
        const int off = kl * 64 + ((((d4 >> 1) ^ (kl >> 2)) & 3) << 4) + ((d4 & 1) << 3);
        *reinterpret_cast<uint2 *>(Ch + off) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
        *reinterpret_cast<uint2 *>(Cl + off) = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));
    }
    float mx = 0.0f;
    for (int kl = threadIdx.x; kl < a.KC; kl += blockDim.x) {
        const float c = kl < a.K ? a.cc[(int64_t)lvl * a.K + kl] : PF_PAD_CC;
        ccs[kl] = c;
        if (kl < a.K) mx = fmaxf(mx, c);
    }
    atomicMax(ccmax_bits, __float_as_uint(mx));
}

__device__ __forceinline__ float vmin_f32(float x, float y) {  // one v_min_f32 (fminf would canonicalise its operands first)
    float m;
    asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(x), "v"(y));
    return m;
}

// one 32-code tile against the wave's 32 items: scores in the accumulators, running (smallest, second smallest, tile of the smallest)
__device__ __forceinline__ void pf32_tile(const char *Ch, const char *Cl, const float *ccs, int t, int off_p0, int off_p1, int h,
                                          const bf16x8_t (&bh)[2], const bf16x8_t (&bl)[2], float &best1, float &best2, int &tbest) {
    const int rowb = t * 64;
    const bf16x8_t ah0 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(Ch + rowb + off_p0));
    const bf16x8_t ah1 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(Ch + rowb + off_p1));
    const bf16x8_t al0 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(Cl + rowb + off_p0));
    const bf16x8_t al1 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(Cl + rowb + off_p1));
    f32x16 acc;  // accumulator j <-> code t + (j & 3) + 8 (j >> 2) + 4 h: starts from |c|^2
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const float4 c4 = *reinterpret_cast<const float4 *>(ccs + t + 8 * g + 4 * h);
        acc[4 * g] = c4.x; acc[4 * g + 1] = c4.y; acc[4 * g + 2] = c4.z; acc[4 * g + 3] = c4.w;
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh[1], acc, 0, 0, 0);
    const float before = best1;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const float s = __uint_as_float((__float_as_uint(acc[j]) & 0xfffffff0u) | (unsigned)j);
        best2 = __builtin_amdgcn_fmed3f(best1, best2, s);
        best1 = vmin_f32(best1, s);
    }
    if (best1 < before) tbest = t;
}

// PIPE (needs NT > 0): two accumulator sets in ping-pong -- the MFMAs of tile t + 1 are issued one at a time between the scan
// operations of tile t (a wave issues in order: behind a chain of dependent MFMAs it could not issue anything), so the matrix pipe
// and the vector ALU work at the same time INSIDE a wave, not only across the waves of a SIMD.
// IDS (the corpus tokenisation of HSemanticIdTokenizer.precompute_corpus_ids, reference h_semids.py:109-195: only the ids leave the
// launch): eval-mode search with NO output but ids -- no z / emb_cat / emb_sum / res_cat / loss stores and none of their arithmetic,
// no winner-row fetch after the last level (nothing consumes that residual).  Same scores, same decisions: the same ids.
// SCAN2 (with PIPE): the two-stage scan.  The one-stage scan spends three vector instructions per score (index pack, med3, min) on every
// one of the level's K scores -- the measured bound of this kernel (71 % of its cycles are vector-ALU issue).  But the index and the
// runner-up are only needed where the minimum is: per 32-code tile the lane takes the tile's minimum with v_min3_f32 (two scores per
// instruction), keeps the two smallest tile minima, and copies the tile's sixteen accumulators aside when the tile is the best so far
// (one v_cndmask each): 28 instructions per tile instead of 50.  The kept tile then gets the exact index-carrying scan once per level,
// and the level's second smallest score is the smaller of the kept tile's second smallest and the second smallest TILE minimum
// (every other tile's minimum bounds its scores from below).  Same scores, same decision rule: the same ids and the same undecided list.
template <int MODE, bool TRAIN, int NW, int NT, bool PIPE = false, bool IDS = false, bool SCAN2 = false>  // NT: 32-code tiles per level when known at compile time (full unroll), else 0
__global__ __launch_bounds__(64 * NW) void rq_forward_pf32_kernel(FwdArgs a) {
    static_assert(!SCAN2 || PIPE, "the two-stage scan lives in the ping-pong form");
    static_assert(!IDS || !TRAIN, "the ids-only form is an eval-mode search");
    extern __shared__ __attribute__((aligned(16))) char pf_lds[];
    __shared__ unsigned ccmax_bits[HIDVAE_MAX_LEVELS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, h = lane >> 5;
    const size_t lvl_bytes = pf32_level_bytes(a.KC);
    if (threadIdx.x < HIDVAE_MAX_LEVELS) ccmax_bits[threadIdx.x] = 0u;
    __syncthreads();
    for (int i = 0; i < a.L; i++) pf32_stage(pf_lds + i * lvl_bytes, a, i, &ccmax_bits[i]);
    __syncthreads();
    constexpr int TILE_ITEMS = 32 * NW;
    const int sw = (n >> 2) & 3;
    const int off_p0 = n * 64 + (((2 * h) ^ sw) << 4), off_p1 = n * 64 + (((2 * h + 1) ^ sw) << 4);  // this lane's A-fragment chunks
    const int64_t ntiles = (a.B + TILE_ITEMS - 1) / TILE_ITEMS;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t item = tile * TILE_ITEMS + wave * 32 + n;
        const bool valid = item < a.B;
        const int64_t src = valid ? item : a.B - 1;
        float r[16];
        load16(a.y + src * D + 16 * h, r);
        if (a.normalize_input) {
            const float den = fmaxf(sqrtf(dotH(r, r)), 1e-12f);
#pragma unroll
            for (int j = 0; j < 16; j++) r[j] = r[j] / den;
        }
        if (!IDS && a.z != nullptr && valid) store16(a.z + item * D + 16 * h, r);
        float loss = 0.0f;
        float esum[16];
        bool undecided_any = false;  // some level of this item was too close to call: the exact kernel redoes the item
        for (int i = 0; i < a.L; i++) {
            if (!IDS && a.res_cat != nullptr && valid) store16(a.res_cat + item * (a.L * D) + i * D + 16 * h, r);
            const float xx = dotH(r, r);
            const char *Ch = pf_lds + i * lvl_bytes;
            const char *Cl = Ch + (size_t)a.KC * 64;
            const float *ccs = reinterpret_cast<const float *>(Ch + (size_t)a.KC * 128);
            // the residual as bf16 hi / lo B fragments: pass p contracts dims [16h + 8p, 16h + 8p + 8) = r[8p .. 8p+7]
            unsigned hw[16], lw[16];
#pragma unroll
            for (int j = 0; j < 16; j++) split_bf16(r[j], hw[j], lw[j]);
            bf16x8_t bh[2], bl[2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const uint4 hu = make_uint4(hw[8 * p] | (hw[8 * p + 1] << 16), hw[8 * p + 2] | (hw[8 * p + 3] << 16),
                                            hw[8 * p + 4] | (hw[8 * p + 5] << 16), hw[8 * p + 6] | (hw[8 * p + 7] << 16));
                const uint4 lu = make_uint4(lw[8 * p] | (lw[8 * p + 1] << 16), lw[8 * p + 2] | (lw[8 * p + 3] << 16),
                                            lw[8 * p + 4] | (lw[8 * p + 5] << 16), lw[8 * p + 6] | (lw[8 * p + 7] << 16));
                bh[p] = __builtin_bit_cast(bf16x8_t, hu);
                bl[p] = __builtin_bit_cast(bf16x8_t, lu);
            }
            float best1 = INFINITY, best2 = INFINITY;
            int tbest = 0;
            if (NT > 0 && PIPE) {
                f32x16 acc2[2];
                auto issue = [&](int tt, int sl) {
                    const int rowb = 32 * tt * 64;
                    const bf16x8_t ah0 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(Ch + rowb + off_p0));
                    const bf16x8_t ah1 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(Ch + rowb + off_p1));
                    const bf16x8_t al0 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(Cl + rowb + off_p0));
                    const bf16x8_t al1 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(Cl + rowb + off_p1));
                    f32x16 acc;
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const float4 c4 = *reinterpret_cast<const float4 *>(ccs + 32 * tt + 8 * g + 4 * h);
                        acc[4 * g] = c4.x; acc[4 * g + 1] = c4.y; acc[4 * g + 2] = c4.z; acc[4 * g + 3] = c4.w;
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh[0], acc, 0, 0, 0);  // (the MFMA order of pf32_tile: the same scores)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh[1], acc, 0, 0, 0);
                    acc2[sl] = acc;
                };
                float g1 = INFINITY, g2 = INFINITY;  // SCAN2: the two smallest tile minima
                f32x16 keep;                           //        the accumulators of the tile that holds the smallest
                if (SCAN2) {
#pragma unroll
                    for (int j = 0; j < 16; j++) keep[j] = INFINITY;
                }
                auto scan = [&](int tt, int sl) {
                    if (SCAN2) {
                        const f32x16 &ac = acc2[sl];
                        // (plain fminf chains: the compiler emits v_min3_f32 on accumulator values AND pads the MFMA -> reader wait states;
                        //  an asm v_min3 reading the accumulators straight after the MFMAs got stale values: 13 wrong ids in 210,000)
                        float tm = __builtin_fminf(__builtin_fminf(ac[0], ac[1]), ac[2]);
                        tm = __builtin_fminf(__builtin_fminf(tm, ac[3]), ac[4]);
                        tm = __builtin_fminf(__builtin_fminf(tm, ac[5]), ac[6]);
                        tm = __builtin_fminf(__builtin_fminf(tm, ac[7]), ac[8]);
                        tm = __builtin_fminf(__builtin_fminf(tm, ac[9]), ac[10]);
                        tm = __builtin_fminf(__builtin_fminf(tm, ac[11]), ac[12]);
                        tm = __builtin_fminf(__builtin_fminf(tm, ac[13]), ac[14]);
                        tm = __builtin_fminf(tm, ac[15]);
                        const bool better = tm < g1;  // (strict: the first of equal tiles stays -- and equal minima make the item undecided anyway)
                        g2 = __builtin_amdgcn_fmed3f(g1, g2, tm);
                        g1 = __builtin_fminf(g1, tm);
                        tbest = better ? 32 * tt : tbest;
#pragma unroll
                        for (int j = 0; j < 16; j++) keep[j] = better ? ac[j] : keep[j];
                        return;
                    }
                    const float before = best1;
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const float sc = __uint_as_float((__float_as_uint(acc2[sl][j]) & 0xfffffff0u) | (unsigned)j);
                        best2 = __builtin_amdgcn_fmed3f(best1, best2, sc);
                        best1 = vmin_f32(best1, sc);
                    }
                    if (best1 < before) tbest = 32 * tt;
                };
                issue(0, 0);
#pragma unroll
                for (int tt = 0; tt < NT; tt++) {
                    if (tt + 1 < NT) issue(tt + 1, (tt + 1) & 1);
                    scan(tt, tt & 1);
                    if (tt + 1 < NT) {
#pragma unroll
                        for (int i = 0; i < 6; i++) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA of the next tile ...
                            __builtin_amdgcn_sched_group_barrier(0x002, SCAN2 ? 5 : 8, 0);  // ... a share of this one's scan operations under it
                        }
                    }
                }
                if (SCAN2) {  // the kept tile: the index-carrying scan, once
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const float sc = __uint_as_float((__float_as_uint(keep[j]) & 0xfffffff0u) | (unsigned)j);
                        best2 = __builtin_amdgcn_fmed3f(best1, best2, sc);
                        best1 = vmin_f32(best1, sc);
                    }
                    best2 = vmin_f32(best2, g2);
                }
            } else if (NT > 0) {
#pragma unroll
                for (int tt = 0; tt < NT; tt++) pf32_tile(Ch, Cl, ccs, 32 * tt, off_p0, off_p1, h, bh, bl, best1, best2, tbest);
            } else {
                for (int t = 0; t < a.KC; t += 32) pf32_tile(Ch, Cl, ccs, t, off_p0, off_p1, h, bh, bl, best1, best2, tbest);
            }
            const int jb = (int)(__float_as_uint(best1) & 15u);
            int idx1 = tbest + (jb & 3) + 8 * (jb >> 2) + 4 * h;
            {   // merge the item's two lanes: smallest (lowest index on ties) and second smallest of the union
                const float o1 = __uint_as_float(swap32_other(__float_as_uint(best1), h));
                const float o2 = __uint_as_float(swap32_other(__float_as_uint(best2), h));
                const int oi = (int)swap32_other((unsigned)idx1, h);
                best2 = fminf(fmaxf(best1, o1), fminf(best2, o2));
                if (o1 < best1 || (o1 == best1 && oi < idx1)) { best1 = o1; idx1 = oi; }
            }
            const float delta = 6.103515625e-05f * (xx + __uint_as_float(ccmax_bits[i]));  // 2^-14 (|x|^2 + max |c|^2)
            undecided_any |= !(best2 - best1 > delta);  // (NaN-safe: anything not provably apart is undecided)
            const int bidx = idx1;                      // decided items: the exact argmin; undecided ones: tentative, redone by the exact kernel
            if (IDS) {
                if (valid && h == 0) a.ids[item * a.L + i] = (int64_t)bidx;
                if (i + 1 < a.L) {  // eval: o = e, the next level searches r - e (quantize.py:146-148, h_rqvae.py:552)
                    float e[16];
                    load16(a.cb_eff + ((int64_t)i * a.K + bidx) * D + 16 * h, e);
#pragma unroll
                    for (int j = 0; j < 16; j++) r[j] = r[j] - e[j];
                }
                continue;
            }
            float e[16];
            load16(a.cb_eff + ((int64_t)i * a.K + bidx) * D + 16 * h, e);
            const float cce = a.cc[(int64_t)i * a.K + bidx];
            float o[16];
            level_output16<MODE, TRAIN>(r, e, xx, cce, o);
            float df[16];
#pragma unroll
            for (int j = 0; j < 16; j++) df[j] = r[j] - e[j];
            const float l1 = dotH(df, df);
            loss = loss + (l1 + a.beta * l1);
            if (valid && h == 0) a.ids[item * a.L + i] = (int64_t)bidx;
            if (a.emb_cat != nullptr && valid) store16(a.emb_cat + item * a.ld_cat + i * D + 16 * h, o);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                esum[j] = (i == 0) ? o[j] : esum[j] + o[j];
                r[j] = r[j] - o[j];
            }
        }
        {   // the wave's undecided items join the list the exact kernel works through: one atomic per wave that has any
            const unsigned long long m = __ballot(undecided_any && valid && h == 0);
            if (m != 0ull) {
                unsigned base = 0u;
                if (lane == 0) base = atomicAdd(a.fix_count, (unsigned)__popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                if (undecided_any && valid && h == 0) a.fix_list[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (int)item;
            }
        }
        if (IDS) continue;
        if (a.emb_sum != nullptr && valid) store16(a.emb_sum + item * D + 16 * h, esum);
        if (valid && a.qloss != nullptr && h == 0) a.qloss[item] = loss;
    }
}

// ------------------------------------------------------------------------------------------------
// The narrow middle of the step in ONE launch (small batches): the encoder's last two layers, the L-level quantisation and the
// decoder's first two layers are all row-local, 16 items wide here, and each is a ~5 us launch on its own:
//     h2 = silu(h1 W2^T) -> y = h2 W3^T -> [z = normalize(y)] -> L levels -> d0 = silu(emb_sum Wd0^T) -> d1 = silu(d0 Wd1^T)
// One workgroup (8 waves) per 16 items.  Every Linear runs on v_mfma_f32_16x16x4_f32 with the WEIGHT rows as the A operand
// (straight from global memory, 16 bytes per lane per 16-wide k-block) and the 16 ITEMS as the B operand, which makes the
// accumulator layout of one layer (lane = item + 16 q, register r  <->  feature 16 t + 4 q + r of output tile t) exactly the B
// operand layout of k-block t of the next layer: activations pass from layer to layer as float4-per-lane images in LDS with no
// re-layout, and the same image with q <-> 8-dims-per-lane bridges to the quantiser's geometry.  Each output is the same
// ORDER-G16 fmaf chain as in gemm_direct16_kernel, so every tensor written here is bit-identical to the unfused launches.
// Measured (B = 1024, 256->128->32 | 3x256 | 32->128->256): 28 us against 31 us for the five launches back to back in a
// microbenchmark and ~37 us for them inside the step; a workgroup pulls 288 KB of weights + 96 KB of codebooks through one CU's
// L2 port (~70 GB/s) and runs ~300 dependent-ish MFMAs per wave, so the launch is bound by that, not by launch overhead.  A
// variant that fetched every weight fragment into ~300 VGPRs up front was 2 us slower.
// ------------------------------------------------------------------------------------------------
struct BneckArgs {
    FwdArgs rq;           // rq.y is not read: y is produced here (and written to y_out)
    const float *h1;      // [B, K2] input activations (silu already applied)
    int K2, N2;           // enc layer a: [N2, K2]; enc layer b: [32, N2]
    const float *W2, *W3;
    float *pre2, *h2, *y_out;
    int Nd0, Nd1;         // dec layer a: [Nd0, 32]; dec layer b: [Nd1, Nd0]
    const float *Wd0, *Wd1;
    float *pre_d0, *d0, *pre_d1, *d1;
};

constexpr int BN_HMAX = 256;  // widest activation kept in LDS (features per item)

// One Linear layer for the workgroup's 16 items.  Hin: K/16 float4-per-lane k-block images; outputs: pre (before the activation,
// optional), act (after it) to global [B, N] and act to Hout (optional).  Output tiles are dealt to the workgroup's waves round-robin,
// two tiles in flight per wave.
template <int NWAVES>
__device__ __forceinline__ void bneck_layer(const float *W, int N, int K, const float4 *Hin, bool silu, float *pre, float *act,
                                            float4 *Hout, int64_t item, bool valid, int wave, int lane) {
    const int i16 = lane & 15, q = lane >> 4;
    const int ntile = N / 16, nkb = K / 16;
    for (int t0 = wave; t0 < ntile; t0 += 2 * NWAVES) {
        const int t1 = t0 + NWAVES;
        const bool two = t1 < ntile;
        const float *w0 = W + (int64_t)(16 * t0 + i16) * K + 4 * q;
        const float *w1 = W + (int64_t)(16 * (two ? t1 : t0) + i16) * K + 4 * q;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (two) {
#pragma unroll 4
            for (int kb = 0; kb < nkb; kb++) {
                const float4 a0 = *reinterpret_cast<const float4 *>(w0 + 16 * kb);
                const float4 a1 = *reinterpret_cast<const float4 *>(w1 + 16 * kb);
                const float4 b = Hin[kb * 64 + lane];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, acc1, 0, 0, 0);
            }
        } else {  // one tile for this wave (layers of <= NWAVES tiles): no second chain to keep the matrix pipe busy for nothing
#pragma unroll 8
            for (int kb = 0; kb < nkb; kb++) {
                const float4 a0 = *reinterpret_cast<const float4 *>(w0 + 16 * kb);
                const float4 b = Hin[kb * 64 + lane];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, acc0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int which = 0; which < 2; which++) {
            if (which == 1 && !two) break;
            const int t = which ? t1 : t0;
            const f32x4 acc = which ? acc1 : acc0;
            float4 v = make_float4(acc[0] + 0.0f, acc[1] + 0.0f, acc[2] + 0.0f, acc[3] + 0.0f);  // (+ bias 0: as the GEMM epilogue)
            float4 o = v;
            if (silu) o = make_float4(hv_silu(v.x), hv_silu(v.y), hv_silu(v.z), hv_silu(v.w));
            if (valid) {
                const int64_t off = item * N + 16 * t + 4 * q;
                if (pre != nullptr) *reinterpret_cast<float4 *>(pre + off) = v;
                if (act != nullptr) *reinterpret_cast<float4 *>(act + off) = o;
            }
            if (Hout != nullptr) Hout[t * 64 + lane] = o;
        }
    }
}

constexpr int BN_WAVES = 8;  // waves per workgroup: they share the 16 items; output tiles and code ranges are dealt among them

template <int MODE>
__global__ __launch_bounds__(64 * BN_WAVES) void bottleneck_fwd_kernel(BneckArgs b) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float cand_d[2][BN_WAVES][16];
    __shared__ int cand_i[2][BN_WAVES][16];
    const FwdArgs &a = b.rq;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int it = lane & 15, q = lane >> 4;
    const int LDK = a.KC + 2;
    const int lvl_floats = 32 * LDK + a.KC;
    float4 *HA = reinterpret_cast<float4 *>(lds + (size_t)a.L * lvl_floats);  // [BN_HMAX/16][64] float4 = 16 items x 256 features
    float4 *HB = HA + (BN_HMAX / 16) * 64;
    float4 *HY = HB + (BN_HMAX / 16) * 64;                                     // [2][64]: 16 items x 32 dims
    const int64_t item = (int64_t)blockIdx.x * ITEMS_PER_WAVE + it;
    const bool in_range = item < a.B;
    const int64_t src = in_range ? item : a.B - 1;
    // the items' input rows as k-block images (all four waves need all of them)
    for (int idx0 = threadIdx.x; idx0 < (b.K2 / 16) * 64; idx0 += 2 * 64 * BN_WAVES) {  // (two loads in flight per thread)
        float4 v[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int idx = idx0 + j * 64 * BN_WAVES;
            const int kb = idx >> 6, l = idx & 63;
            const int64_t row = (int64_t)blockIdx.x * ITEMS_PER_WAVE + (l & 15);
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < (b.K2 / 16) * 64) v[j] = *reinterpret_cast<const float4 *>(b.h1 + (row < a.B ? row : a.B - 1) * b.K2 + 16 * kb + 4 * (l >> 4));
        }
#pragma unroll
        for (int j = 0; j < 2; j++)
            if (idx0 + j * 64 * BN_WAVES < (b.K2 / 16) * 64) HA[idx0 + j * 64 * BN_WAVES] = v[j];
    }
    for (int i = 0; i < a.L; i++) stage_codes(lds + i * lvl_floats, a, i, 0);
    __syncthreads();
    bneck_layer<BN_WAVES>(b.W2, b.N2, b.K2, HA, true, b.pre2, b.h2, HB, item, in_range, wave, lane);
    __syncthreads();
    bneck_layer<BN_WAVES>(b.W3, D, b.N2, HB, false, nullptr, b.y_out, HY, item, in_range, wave, lane);
    __syncthreads();
    // quantiser geometry: lane (it, q) holds dims 8q .. 8q+7 = registers of (tile q>>1, quarter 2(q&1)) and (.., 2(q&1)+1)
    float r[8];
    {
        const float4 lo = HY[(q >> 1) * 64 + (2 * (q & 1)) * 16 + it];
        const float4 hi = HY[(q >> 1) * 64 + (2 * (q & 1) + 1) * 16 + it];
        r[0] = lo.x; r[1] = lo.y; r[2] = lo.z; r[3] = lo.w; r[4] = hi.x; r[5] = hi.y; r[6] = hi.z; r[7] = hi.w;
    }
    const bool valid = in_range && wave == 0;
    if (a.normalize_input) {
        const float den = fmaxf(sqrtf(dotQ(r, r)), 1e-12f);
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = r[j] / den;
    }
    if (a.z != nullptr && valid) store8(a.z + item * D + 8 * q, r);
    float loss;
    float esum[8];
    int phase = 0;
    unsigned long long tuple;
    rq_level_loop<MODE, true, true, true, BN_WAVES>(a, lds, cand_d, cand_i, phase, wave, it, q, item, valid, r, esum, loss, tuple);
    if (valid) {
        if (a.emb_sum != nullptr) store8(a.emb_sum + item * D + 8 * q, esum);
        if (a.qloss != nullptr && q == 0) a.qloss[item] = loss;
    }
    // id census, first half: the first table probe is issued here and read after the decoder layers (its latency hides under them)
    const bool census_lane = a.census != nullptr && valid && q == 0;
    unsigned long long census_calls = 0ull, census_cur = 0ull;
    int64_t census_slot = 0;
    if (a.census != nullptr && wave == 0) census_calls = a.census[a.census_size + 2];
    if (census_lane) {
        unsigned long long h = 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < a.L; i++) {
            h ^= ((tuple >> (10 * i)) & 1023ull) + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
            h *= 0xBF58476D1CE4E5B9ull;
            h ^= h >> 29;
        }
        census_slot = (int64_t)(h % (unsigned long long)a.census_size);
        census_cur = __atomic_load_n(a.census + census_slot, __ATOMIC_RELAXED);
    }
    __syncthreads();  // HY is free again (every wave has read its r)
    if (wave == 0) {  // every wave carries the same esum: one of them writes the k-block images of the decoder's input
        HY[(q >> 1) * 64 + (2 * (q & 1)) * 16 + it] = make_float4(esum[0], esum[1], esum[2], esum[3]);
        HY[(q >> 1) * 64 + (2 * (q & 1) + 1) * 16 + it] = make_float4(esum[4], esum[5], esum[6], esum[7]);
    }
    __syncthreads();
    bneck_layer<BN_WAVES>(b.Wd0, b.Nd0, D, HY, true, b.pre_d0, b.d0, HA, item, in_range, wave, lane);
    __syncthreads();
    // id census, middle: the probe has landed; a slot of an older generation is claimed now, the answer is read after the last layer
    constexpr unsigned long long CENSUS_GEN_MOD = 0xFFFFFEull;
    const unsigned long long census_gen = census_calls % CENSUS_GEN_MOD + 1ull;
    bool census_claimed = false;
    unsigned long long census_prev = 0ull;
    if (census_lane && (census_cur >> 40) != census_gen) {
        census_prev = atomicCAS(a.census + census_slot, census_cur, (census_gen << 40) | tuple);
        census_claimed = true;
    }
    bneck_layer<BN_WAVES>(b.Wd1, b.Nd1, b.Nd0, HA, true, b.pre_d1, b.d1, nullptr, item, in_range, wave, lane);
    // id census, second half (p_unique_ids of h_rqvae.py:645-648 = distinct id tuples / B): the scheme of id_stats_kernel
    // (core_ops.hip) -- a never-cleared open-addressing table with generation-tagged slots and ONE packed ticket/count atomic per
    // workgroup -- except that a slot holds the tuple itself, (generation << 40) | 10 bits per level, so no workgroup reads ids
    // another one wrote in this launch.  Saves the 11 us launch of its own; costs one atomic round trip at this kernel's tail.
    if (a.census != nullptr && wave == 0) {
        constexpr unsigned long long GEN_MOD = CENSUS_GEN_MOD, LOW40 = (1ull << 40) - 1ull;
        unsigned long long *ctrl = a.census + a.census_size;
        const unsigned long long gen = census_gen;
        bool is_new = census_claimed && census_prev == census_cur;  // first item with this tuple, settled by the early claim
        if (census_lane && !is_new) {
            const unsigned long long mine = (gen << 40) | tuple;
            unsigned long long cur = census_claimed ? census_prev : census_cur;  // (a lost claim returns the slot's real content)
            int64_t slot = census_slot;
            for (int64_t probe = 0; probe < a.census_size;) {
                if ((cur >> 40) != gen) {
                    const unsigned long long prev = atomicCAS(a.census + slot, cur, mine);
                    if (prev == cur) { is_new = true; break; }  // first item with this tuple
                    cur = prev;
                    if ((cur >> 40) != gen) continue;  // lost a race against a stale view of the slot: look again
                }
                if ((cur & LOW40) == tuple) break;  // duplicate of an already counted tuple
                slot = slot + 1 == a.census_size ? 0 : slot + 1;
                cur = __atomic_load_n(a.census + slot, __ATOMIC_RELAXED);
                probe++;
            }
        }
        const int fresh = __popcll(__ballot(is_new));
        int last = 0;
        unsigned long long total = 0ull;
        if (lane == 0) {
            const unsigned long long old = atomicAdd(ctrl, ((unsigned long long)fresh << 32) + 1ull);
            last = (old & 0xFFFFFFFFull) == (unsigned long long)gridDim.x - 1ull;
            total = (unsigned long long)fresh + (old >> 32);
        }
        last = __shfl(last, 0);
        if (last) {  // every other workgroup has drawn its ticket, i.e. finished its probes
            if (lane == 0) {
                *a.p_unique = (float)total / (float)a.B;
                ctrl[0] = 0ull;
                ctrl[1] = 0ull;
                ctrl[2] = census_calls + 1ull;
            }
            if ((census_calls + 1ull) % GEN_MOD == 0ull)
                for (int64_t i = lane; i < a.census_size; i += 64) a.census[i] = 0ull;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward: recompute the residual chain from z and ids, then walk the levels in reverse
// ------------------------------------------------------------------------------------------------
struct BwdArgs {
    const float *y;
    const float *z;
    int64_t B;
    int normalize_input;
    const float *cb_eff;
    const float *cc;
    int64_t K;
    float beta;
    const int64_t *ids;
    const float *g_cat;
    int64_t ld_gcat;
    const float *g_sum;
    const float *g_z_in;
    int64_t g_z_rows;  // rows of g_z_in that exist (B for a dense gradient, n_layers for the uniqueness term)
    float gq;
    const float *gq_items;  // per-item d(loss)/d(qloss[b]) at stride gq_stride (0 = one broadcast scalar), times gq
    int64_t gq_stride;
    float *g_y;
    float *dE_rows;
    // optional, instead of g_cat: the gradient of emb_cat as prefix slices still to be added up (hidvae_rq_backward_slices) -- slice s is a
    // contiguous [B, width[s]] tensor holding the gradient of the first width[s] columns; summed here in slice order, from zero
    int n_slices;
    const float *slice[2 * HIDVAE_MAX_LEVELS];
    int slice_w[2 * HIDVAE_MAX_LEVELS];
};

template <int MODE, int L>
__global__ __launch_bounds__(WG_THREADS) void rq_backward_kernel(BwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int it = lane & 15, q = lane >> 4;
    const int64_t item = ((int64_t)blockIdx.x * 4 + wave) * ITEMS_PER_WAVE + it;
    const bool valid = item < a.B;
    const int64_t src = valid ? item : a.B - 1;
    float rs[L][8];
    int64_t code[L];
    float zr[8];
    load8(a.z + src * D + 8 * q, zr);
    {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = zr[j];
#pragma unroll
        for (int i = 0; i < L; i++) {
#pragma unroll
            for (int j = 0; j < 8; j++) rs[i][j] = r[j];
            code[i] = a.ids[src * L + i];
            if (i + 1 < L) {
                float e[8], o[8], u[8], qv[8], w[8];
                load8(a.cb_eff + ((int64_t)i * a.K + code[i]) * D + 8 * q, e);
                const float xx = dotQ(r, r);
                level_output<MODE, true>(r, e, xx, a.cc[(int64_t)i * a.K + code[i]], o, u, qv, w);
#pragma unroll
                for (int j = 0; j < 8; j++) r[j] = r[j] - o[j];
            }
        }
    }
    float gs[8];
    if (a.g_sum != nullptr) load8(a.g_sum + src * D + 8 * q, gs);
    else {
#pragma unroll
        for (int j = 0; j < 8; j++) gs[j] = 0.0f;
    }
    float R[8];
#pragma unroll
    for (int j = 0; j < 8; j++) R[j] = 0.0f;
#pragma unroll
    for (int i = L - 1; i >= 0; i--) {
        float e[8];
        load8(a.cb_eff + ((int64_t)i * a.K + code[i]) * D + 8 * q, e);
        float go[8];
        if (a.g_cat != nullptr) load8(a.g_cat + src * a.ld_gcat + i * D + 8 * q, go);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) go[j] = 0.0f;
            if (a.n_slices > 0) {  // (the order and the starting zero of hidvae_sum_prefix_slices: the same bits)
#pragma unroll
                for (int s = 0; s < 2 * HIDVAE_MAX_LEVELS; s++)
                    if (s < a.n_slices && i * D < a.slice_w[s]) {
                        float t[8];
                        load8(a.slice[s] + src * a.slice_w[s] + i * D + 8 * q, t);
#pragma unroll
                        for (int j = 0; j < 8; j++) go[j] += t[j];
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) go[j] = (go[j] + gs[j]) - R[j];  // o_i feeds sum, concat and -r_{i+1}
        float jt[8];
        if (MODE == HIDVAE_MODE_STE) {
#pragma unroll
            for (int j = 0; j < 8; j++) jt[j] = go[j];
        } else {  // d o / d r = I - 2 w w^T + 2 q u^T with (u,q,w) constants  =>  J^T g = g - 2 w (w.g) + 2 u (q.g)
            float o[8], u[8], qv[8], w[8];
            const float xx = dotQ(rs[i], rs[i]);
            level_output<MODE, true>(rs[i], e, xx, a.cc[(int64_t)i * a.K + code[i]], o, u, qv, w);
            const float wg = dotQ(w, go), qg = dotQ(qv, go);
#pragma unroll
            for (int j = 0; j < 8; j++) jt[j] = (go[j] - 2.0f * (wg * w[j])) + 2.0f * (qg * u[j]);
        }
        float de[8];
        const float gqb = a.gq_items != nullptr ? a.gq * a.gq_items[src * a.gq_stride] : a.gq;
        const float cq = 2.0f * gqb, cr = 2.0f * a.beta * gqb;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float df = rs[i][j] - e[j];
            R[j] = (R[j] + jt[j]) + cr * df;  // commitment term: beta |r - sg(e)|^2
            de[j] = -(cq * df);               // codebook term:   |sg(r) - e|^2
        }
        if (valid && a.dE_rows != nullptr) store8(a.dE_rows + item * (L * D) + i * D + 8 * q, de);
    }
    if (a.g_z_in != nullptr && src < a.g_z_rows) {
        float gz[8];
        load8(a.g_z_in + src * D + 8 * q, gz);
#pragma unroll
        for (int j = 0; j < 8; j++) R[j] = R[j] + gz[j];
    }
    if (a.normalize_input) {  // z = y / max(|y|, eps)  =>  g_y = (g_z - z (z.g_z)) / max(|y|, eps)
        float yv[8];
        load8(a.y + src * D + 8 * q, yv);
        const float den = fmaxf(sqrtf(dotQ(yv, yv)), 1e-12f);
        const float zg = dotQ(zr, R);
#pragma unroll
        for (int j = 0; j < 8; j++) R[j] = (R[j] - zr[j] * zg) / den;
    }
    if (valid) store8(a.g_y + item * D + 8 * q, R);
}

// ------------------------------------------------------------------------------------------------
// codebook gradient: one wave per (level, code); items scanned in ascending order => bit-reproducible
// ------------------------------------------------------------------------------------------------
struct CbGradArgs {
    const int64_t *ids;
    const float *dE_rows;
    int64_t B;
    int L;
    int64_t K;
    const float *E[HIDVAE_MAX_LEVELS];
    const float *cb_eff;
    int normalize[HIDVAE_MAX_LEVELS];
    float *gE[HIDVAE_MAX_LEVELS];
    int accumulate;
    // large batches: blockIdx.y = slab of `slab_items` items; every (code, slab) wave leaves its sum in partial[row][slab][32] and
    // codebook_grad_final_kernel adds the slabs in ascending order (slabs == 1: the one-pass form, partial unused)
    int slabs;
    int64_t slab_items;
    float *partial;
};

template <bool STAGED>  // STAGED (K % 4 == 0: the workgroup's four codes share a level): the chunk's ids go through LDS once
__global__ __launch_bounds__(WG_THREADS) void codebook_grad_kernel(CbGradArgs a) {
    // Two phases per 1024-item chunk so that no load depends on another: (A) scan the ids of the chunk (independent,
    // coalesced loads) and append the matching item numbers, in ascending order, to a per-wave LDS list;
    // (B) walk the list and add the rows (lane & 31 = d; lanes 0..31 add the even entries, lanes 32..63 the odd ones, each in
    // ascending item order, and the two chains are added at the end) => a fixed summation order, bit-reproducible sums.
    // Every wave needs the ids of ALL items of its level; read straight from global memory that is L*K waves x B strided 8-byte
    // loads (150 MB of L2 traffic at B = 8192, 69 us); staged, a workgroup reads each chunk once for its four waves.
    __shared__ int hits[4][2048];
    __shared__ int chunk[STAGED ? 4096 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    const bool live = row < (int64_t)a.L * a.K;
    if (!STAGED && !live) return;
    const int64_t rowc = live ? row : (int64_t)a.L * a.K - 1;
    const int lvl = (int)(rowc / a.K);
    const int64_t k = rowc - (int64_t)lvl * a.K;
    const int d = lane & 31;
    int *list = hits[wave];
    float acc = 0.0f;
    // operands of the epilogue are fetched first so that their latency hides behind the scan
    const bool nrm = a.normalize[lvl] != 0;
    const float ev = nrm ? a.E[lvl][k * D + d] : 0.0f;
    const float cv = nrm ? a.cb_eff[rowc * D + d] : 0.0f;
    int n = 0;
    // (B) the list so far, rows fetched eight at a time and added in ascending item order.  It runs when the list could overflow
    // on the next chunk and at the end: normally ONCE, so the scan of all chunks is not interleaved with dependent row loads.
    const int half = lane >> 5;  // the two half-waves take the even / the odd entries of the list: two fixed-order chains
    auto drain = [&]() {
        __builtin_amdgcn_wave_barrier();
        for (int e0 = 0; e0 < n; e0 += 16) {
            float r[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int e = e0 + 2 * j + half;
                r[j] = e < n ? a.dE_rows[(int64_t)list[e] * ((int64_t)a.L * D) + lvl * D + d] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (e0 + 2 * j + half < n) acc += r[j];
        }
        __builtin_amdgcn_wave_barrier();
        n = 0;
    };
    constexpr int CH = STAGED ? 4096 : 1024;  // items per staging round (16 independent id loads per thread in flight)
    const int64_t lo = (int64_t)blockIdx.y * a.slab_items;
    const int64_t hi = lo + a.slab_items < a.B ? lo + a.slab_items : a.B;
    for (int64_t c0 = lo; c0 < hi; c0 += CH) {
        if (STAGED) {
            __syncthreads();  // the previous chunk has been consumed by every wave
#pragma unroll
            for (int j = 0; j < CH / 256; j++) {
                const int64_t b = c0 + j * 256 + threadIdx.x;
                chunk[j * 256 + threadIdx.x] = b < hi ? (int)a.ids[b * a.L + lvl] : -1;
            }
            __syncthreads();
        }
        for (int s0 = 0; s0 < CH && c0 + s0 < hi; s0 += 1024) {
            if (n > 1024) drain();  // (wave-uniform)
            int64_t idv[16];
            if (STAGED) {
#pragma unroll
                for (int j = 0; j < 16; j++) idv[j] = chunk[s0 + j * 64 + lane];
            } else {
#pragma unroll
                for (int j = 0; j < 16; j++) {  // 16 independent loads in flight
                    const int64_t b = c0 + s0 + j * 64 + lane;
                    idv[j] = b < hi ? a.ids[b * a.L + lvl] : -1;
                }
            }
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const bool hit = idv[j] == k;
                const unsigned long long m = __ballot(hit);
                if (hit) list[n + __popcll(m & ((1ull << lane) - 1ull))] = (int)(c0 + s0 + j * 64 + lane);
                n += __popcll(m);
            }
        }
    }
    drain();
    acc = acc + __shfl_xor(acc, 32);  // even-entry chain + odd-entry chain (the same value on both halves)
    if (!live) return;
    if (a.slabs > 1) {  // this slab's share; the epilogue runs in codebook_grad_final_kernel
        if (lane < 32) a.partial[(rowc * a.slabs + blockIdx.y) * D + d] = acc;
        return;
    }
    if (nrm) {  // c = E / max(|E|, eps)  =>  gE = (g - c (c.g)) / max(|E|, eps)
        float n2 = ev * ev, cg = cv * acc;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            n2 += __shfl_xor(n2, o);
            cg += __shfl_xor(cg, o);
        }
        acc = (acc - cv * cg) / fmaxf(sqrtf(n2), 1e-12f);
    }
    if (lane < 32) {
        float *dst = a.gE[lvl] + k * D + d;
        *dst = a.accumulate ? *dst + acc : acc;
    }
}

// slabs of a large batch, added in ascending order, then the same epilogue: one half-wave per (level, code)
__global__ __launch_bounds__(WG_THREADS) void codebook_grad_final_kernel(CbGradArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= (int64_t)a.L * a.K) return;
    const int lvl = (int)(row / a.K);
    const int64_t k = row - (int64_t)lvl * a.K;
    const int d = lane & 31;
    const bool nrm = a.normalize[lvl] != 0;
    const float ev = nrm ? a.E[lvl][k * D + d] : 0.0f;
    const float cv = nrm ? a.cb_eff[row * D + d] : 0.0f;
    const float *p = a.partial + row * a.slabs * D + d;
    float acc = p[0];
    for (int s = 1; s < a.slabs; s++) acc += p[(int64_t)s * D];
    if (nrm) {
        float n2 = ev * ev, cg = cv * acc;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            n2 += __shfl_xor(n2, o);
            cg += __shfl_xor(cg, o);
        }
        acc = (acc - cv * cg) / fmaxf(sqrtf(n2), 1e-12f);
    }
    if (lane < 32) {
        float *dst = a.gE[lvl] + k * D + d;
        *dst = a.accumulate ? *dst + acc : acc;
    }
}

// rows of 32 L2-normalised with the same partial-sum order as the RQ prologue, so HRqVae.encode() and the
// fused forward produce the same z bit for bit (modules/normalize.py:7-8)
__global__ __launch_bounds__(WG_THREADS) void l2norm32_kernel(const float *x, int64_t M, int64_t ldx, float eps, float *out,
                                                              int64_t ldo, float *norms) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int it = lane & 15, q = lane >> 4;
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * ITEMS_PER_WAVE + it;
    const int64_t src = row < M ? row : M - 1;
    float v[8];
    load8(x + src * ldx + 8 * q, v);
    const float nrm = sqrtf(dotQ(v, v));
    const float den = fmaxf(nrm, eps);
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = v[j] / den;
    if (row < M) {
        store8(out + row * ldo + 8 * q, v);
        if (norms != nullptr && q == 0) norms[row] = nrm;
    }
}

size_t level_lds_bytes(int KC) { return (size_t)(32 * (KC + 2) + KC) * sizeof(float); }

template <int MODE, bool TRAIN>
int launch_fwd(const FwdArgs &a, bool resident, bool csplit, int nw, int grid, size_t lds, hipStream_t s) {
#define HV_GO(R, C, W)                                                                                                     \
    {                                                                                                                      \
        auto kern = rq_forward_kernel<MODE, TRAIN, R, C, W>;                                                               \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * W), lds, s, a);                                                     \
    }
    if (!resident && csplit) {  // streamed code-split variant (no LDS copy of the codes)
        auto kern = rq_forward_kernel<MODE, TRAIN, false, true, 4, true>;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, a);
    } else if (resident && csplit) HV_GO(true, true, 4)
    else if (resident && nw == 16) HV_GO(true, false, 16)
    else if (resident && nw == 8) HV_GO(true, false, 8)
    else if (resident) HV_GO(true, false, 4)
    else if (nw == 8) HV_GO(false, false, 8)
    else HV_GO(false, false, 4)
#undef HV_GO
    HV_LAUNCH_CHECK("rq_forward");
    return HIDVAE_OK;
}

template <int MODE>
int launch_bwd(const BwdArgs &a, int L, int grid, hipStream_t s) {
    switch (L) {
#define HV_CASE(n) \
    case n: hipLaunchKernelGGL((rq_backward_kernel<MODE, n>), dim3(grid), dim3(WG_THREADS), 0, s, a); break;
        HV_CASE(1) HV_CASE(2) HV_CASE(3) HV_CASE(4) HV_CASE(5) HV_CASE(6) HV_CASE(7) HV_CASE(8)
#undef HV_CASE
        default: return hv_fail(HIDVAE_EINVAL, "rq_backward: L=%d out of range", L);
    }
    HV_LAUNCH_CHECK("rq_backward");
    return HIDVAE_OK;
}

}  // namespace

extern "C" int hidvae_codebook_prepare(const float *const *E_host, const int32_t *normalize_host, int L, int64_t K,
                                       float *cb_eff, float *cc, int embed_dim, void *stream) {
    HV_REQUIRE(L >= 1 && L <= HIDVAE_MAX_LEVELS, "codebook_prepare: n_layers=%d not in [1,%d]", L, HIDVAE_MAX_LEVELS);
    HV_REQUIRE(K >= 1 && E_host && cb_eff && cc, "codebook_prepare: bad arguments");
    if (embed_dim != D) {
        HV_REQUIRE(hv_rqg_dim_ok(embed_dim), "codebook_prepare: embed_dim=%d (a multiple of 4, at most 64)", embed_dim);
        return hv_rqg_prepare(E_host, normalize_host, L, K, embed_dim, cb_eff, cc, (hipStream_t)stream);
    }
    PrepArgs a{};
    for (int i = 0; i < L; i++) {
        a.E[i] = E_host[i];
        a.normalize[i] = normalize_host ? normalize_host[i] : 0;
    }
    a.L = L; a.K = K; a.cb_eff = cb_eff; a.cc = cc;
    const int grid = (int)hv_cdiv((int64_t)L * K, ITEMS_PER_WG);
    hipLaunchKernelGGL(codebook_prepare_kernel, dim3(grid), dim3(WG_THREADS), 0, (hipStream_t)stream, a);
    HV_LAUNCH_CHECK("codebook_prepare");
    return HIDVAE_OK;
}

extern "C" int hidvae_codebook_prepare_adamw(const float *const *E_host, const int32_t *normalize_host, int L, int64_t K, float *cb_eff,
                                             float *cc, int64_t *step_dev, const float *base_lr_dev, const float *wd_dev, int n_tensors,
                                             float beta1, float beta2, float eta_min, int64_t T_max, int64_t step_size, float gamma,
                                             float *hyper_dev, int embed_dim, void *stream) {
    HV_REQUIRE(L >= 1 && L <= HIDVAE_MAX_LEVELS, "codebook_prepare_adamw: n_layers=%d not in [1,%d]", L, HIDVAE_MAX_LEVELS);
    HV_REQUIRE(K >= 1 && E_host && cb_eff && cc, "codebook_prepare_adamw: bad arguments");
    HV_REQUIRE(step_dev && base_lr_dev && wd_dev && hyper_dev && n_tensors >= 1, "codebook_prepare_adamw: bad optimizer arguments");
    if (embed_dim != D) {  // other widths: the two launches one after the other
        HV_REQUIRE(hv_rqg_dim_ok(embed_dim), "codebook_prepare_adamw: embed_dim=%d (a multiple of 4, at most 64)", embed_dim);
        const int rc = hv_rqg_prepare(E_host, normalize_host, L, K, embed_dim, cb_eff, cc, (hipStream_t)stream);
        if (rc != HIDVAE_OK) return rc;
        return hidvae_adamw_prepare(step_dev, base_lr_dev, wd_dev, n_tensors, beta1, beta2, eta_min, T_max, step_size, gamma, hyper_dev, stream);
    }
    PrepArgs a{};
    for (int i = 0; i < L; i++) {
        a.E[i] = E_host[i];
        a.normalize[i] = normalize_host ? normalize_host[i] : 0;
    }
    a.L = L; a.K = K; a.cb_eff = cb_eff; a.cc = cc;
    a.carry = 1;
    a.adam = HvAdamPrepare{step_dev, base_lr_dev, wd_dev, n_tensors, beta1, beta2, eta_min, T_max, step_size, gamma, hyper_dev};
    const int grid = (int)hv_cdiv((int64_t)L * K, ITEMS_PER_WG) + 1;
    hipLaunchKernelGGL(codebook_prepare_kernel, dim3(grid), dim3(WG_THREADS), 0, (hipStream_t)stream, a);
    HV_LAUNCH_CHECK("codebook_prepare_adamw");
    return HIDVAE_OK;
}

extern "C" int hidvae_rq_forward(const float *y, int64_t B, int normalize_input, const float *cb_eff, const float *cc,
                                 int L, int64_t K, int mode, int training, float beta, float *z, int64_t *ids,
                                 float *emb_cat, int64_t ld_cat, float *emb_sum, float *res_cat, float *qloss,
                                 void *workspace, int embed_dim, int distance, void *stream) {
    HV_REQUIRE(L >= 1 && L <= HIDVAE_MAX_LEVELS, "rq_forward: n_layers=%d not in [1,%d]", L, HIDVAE_MAX_LEVELS);
    HV_REQUIRE(distance == HIDVAE_DIST_L2 || distance == HIDVAE_DIST_COSINE, "rq_forward: distance %d", distance);
    HV_REQUIRE(B >= 1 && K >= 1, "rq_forward: empty batch or codebook (B=%lld K=%lld)", (long long)B, (long long)K);
    HV_REQUIRE(y && cb_eff && cc && ids, "rq_forward: null pointer");
    HV_REQUIRE(mode == HIDVAE_MODE_STE || mode == HIDVAE_MODE_ROTATION || !training,
               "rq_forward: mode %d is not fused (GUMBEL_SOFTMAX is composed from GEMM + softmax)", mode);
    if (embed_dim != D || distance != HIDVAE_DIST_L2) {  // (the cosine ranking, which no shipped config selects, lives in the width-independent kernel only)
        HV_REQUIRE(hv_rqg_dim_ok(embed_dim), "rq_forward: embed_dim=%d (a multiple of 4, at most 64)", embed_dim);
        HV_REQUIRE(emb_cat == nullptr || ld_cat >= (int64_t)L * embed_dim, "rq_forward: ld_cat=%lld", (long long)ld_cat);
        return hv_rqg_forward(y, B, normalize_input, cb_eff, cc, L, K, embed_dim, mode, training, beta, z, ids, emb_cat, ld_cat, emb_sum, res_cat,
                              qloss, distance == HIDVAE_DIST_COSINE, (hipStream_t)stream);
    }
    HV_REQUIRE(emb_cat == nullptr || (ld_cat >= (int64_t)L * D && ld_cat % 4 == 0), "rq_forward: ld_cat=%lld", (long long)ld_cat);
    FwdArgs a{};
    a.y = y; a.B = B; a.normalize_input = normalize_input; a.cb_eff = cb_eff; a.cc = cc; a.L = L; a.K = K;
    const int64_t Kp = hv_cdiv(K, 32) * 32;
    a.KC = (int)(Kp < MAX_KC ? Kp : MAX_KC);
    a.nchunks = (int)hv_cdiv(K, a.KC);
    a.beta = beta; a.z = z; a.ids = ids; a.emb_cat = emb_cat; a.ld_cat = ld_cat; a.emb_sum = emb_sum;
    a.res_cat = res_cat; a.qloss = qloss;
    const size_t per_level = level_lds_bytes(a.KC);
    const bool resident = a.nchunks == 1 && per_level * (size_t)L <= 152 * 1024;
    const size_t lds = resident ? per_level * (size_t)L : per_level;
    // small batches: split the codes over the 4 waves of a workgroup (16 items per workgroup) to cut the serial search
    // (codebooks that do not fit LDS -- 4 x 1024: the streamed code-split kernel up to 16384 items, 16 per workgroup; above that the
    //  staged kernel amortises a level's codes over 128 items per workgroup)
    const bool csplit = (resident && a.KC % 128 == 0 && B <= 4096) || (!resident && B <= 16384);
    // large batches: 8 or 16 waves per workgroup (two / four per SIMD) share one LDS copy of the codebooks, so one wave's per-level VALU
    // work (rotation, loss, argmin merge) overlaps the other's MFMAs; LDS allows only one workgroup per CU either way
    // (measured at 1,048,576 items, 3x256: 4 waves 1050 us, 8 waves 787 us, 16 waves 704 us; 85 VGPRs, so 4 waves per SIMD fit)
    int nw = (!csplit && B >= 256 * 128) ? 8 : 4;
    if (!csplit && resident && B >= 256 * 256) nw = 16;
    const int64_t ntiles = hv_cdiv(B, csplit ? ITEMS_PER_WAVE : ITEMS_PER_WAVE * nw);
    const int64_t gcap = (csplit && !resident) ? 2048 : 256;  // one workgroup per CU when LDS-limited; the streamed kernel holds no codes in LDS
    const int grid = (int)(ntiles < gcap ? ntiles : gcap);  // grid-stride over tiles
    hipStream_t s = (hipStream_t)stream;
    // large batches whose codebooks fit as bf16 hi/lo images: split-bf16 prefilter, then the exact kernel over the items it could not
    // decide (bit-identical results; needs the caller's workspace for that list: without one the exact kernels run alone)
    const int KCp = (int)(hv_cdiv(K, 32) * 32);
    const size_t pf32_bytes = pf32_level_bytes(KCp) * (size_t)L;
    if (workspace != nullptr && B >= 256 * 256 && B < (1ll << 31) && pf32_bytes <= 152 * 1024) {
        const bool ids_only = !training && !z && !emb_cat && !emb_sum && !res_cat && !qloss;  // eval, every output but ids NULL
        FwdArgs p = a;
        p.KC = KCp;
        p.nchunks = 1;
        p.fix_count = reinterpret_cast<unsigned *>(workspace);  // 16-byte header, then the list
        p.fix_list = reinterpret_cast<int *>(workspace) + 4;
        if (hipMemsetAsync(workspace, 0, 16, s) != hipSuccess) return hv_fail(HIDVAE_ELAUNCH, "rq_forward: clearing the undecided-item counter failed");
        const bool pipe = KCp == 256;  // ping-pong accumulators need the tile count at compile time
        // waves per workgroup (one workgroup per CU: the LDS images take 101 KB at 3 x 256).  ids-only: 16 = four per SIMD at 124
        // registers per lane from 131,072 items on (1M items: 248 us against 258 at 8; 65,536 items: 52 against 45, the launch is
        // then 128 workgroups); the full forms keep their output rows live, 8 waves at 158-166 registers (12 waves: no faster)
        const int pnw = !pipe ? 12 : ((ids_only && B >= 128 * 1024) ? 16 : 8);
        const int64_t nt = hv_cdiv(B, 32 * pnw);
        const int pgrid = (int)(nt < 256 ? nt : 256);
#define HV_PF32_GO(M, T, W, NTI, PP, II)                                                                                       \
    {                                                                                                                          \
        auto kern = rq_forward_pf32_kernel<M, T, W, NTI, PP, II, PP>;  /* the two-stage scan wherever the tiles ping-pong */    \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pf32_bytes); \
        hipLaunchKernelGGL(kern, dim3(pgrid), dim3(64 * W), pf32_bytes, s, p);                                                 \
    }
        if (ids_only) {
            if (!pipe) HV_PF32_GO(HIDVAE_MODE_STE, false, 12, 0, false, true)
            else if (pnw == 16) HV_PF32_GO(HIDVAE_MODE_STE, false, 16, 8, true, true)
            else HV_PF32_GO(HIDVAE_MODE_STE, false, 8, 8, true, true)
        } else if (!training) {
            if (!pipe) HV_PF32_GO(HIDVAE_MODE_STE, false, 12, 0, false, false)
            else HV_PF32_GO(HIDVAE_MODE_STE, false, 8, 8, true, false)
        } else if (mode == HIDVAE_MODE_STE) {
            if (!pipe) HV_PF32_GO(HIDVAE_MODE_STE, true, 12, 0, false, false)
            else HV_PF32_GO(HIDVAE_MODE_STE, true, 8, 8, true, false)
        } else {
            if (!pipe) HV_PF32_GO(HIDVAE_MODE_ROTATION, true, 12, 0, false, false)
            else HV_PF32_GO(HIDVAE_MODE_ROTATION, true, 8, 8, true, false)
        }
#undef HV_PF32_GO
        HV_LAUNCH_CHECK("rq_forward prefilter");
        // the exact search over the listed items (streamed code-split form: 16 items per workgroup, no LDS staging, so workgroups
        // beyond the list's end cost nothing); it rewrites every output of those items
        FwdArgs f = a;
        f.list = p.fix_list;
        f.list_count = p.fix_count;
        const int64_t ft = hv_cdiv(B, ITEMS_PER_WAVE);
        const int fgrid = (int)(ft < 1024 ? ft : 1024);
        if (!training) return launch_fwd<HIDVAE_MODE_STE, false>(f, false, true, 4, fgrid, 0, s);
        if (mode == HIDVAE_MODE_STE) return launch_fwd<HIDVAE_MODE_STE, true>(f, false, true, 4, fgrid, 0, s);
        return launch_fwd<HIDVAE_MODE_ROTATION, true>(f, false, true, 4, fgrid, 0, s);
    }
    if (!training) return launch_fwd<HIDVAE_MODE_STE, false>(a, resident, csplit, nw, grid, lds, s);
    if (mode == HIDVAE_MODE_STE) return launch_fwd<HIDVAE_MODE_STE, true>(a, resident, csplit, nw, grid, lds, s);
    return launch_fwd<HIDVAE_MODE_ROTATION, true>(a, resident, csplit, nw, grid, lds, s);
}

extern "C" int hidvae_bottleneck_fwd(const float *h1, int64_t B, int K2, int N2, const float *W2, const float *W3, float *pre2, float *h2,
                                     float *y, int normalize_input, const float *cb_eff, const float *cc, int L, int64_t K, int mode,
                                     float beta, float *z, int64_t *ids, float *emb_cat, int64_t ld_cat, float *emb_sum, float *qloss,
                                     int Nd0, int Nd1, const float *Wd0, const float *Wd1, float *pre_d0, float *d0, float *pre_d1,
                                     float *d1, float *embs_norm, float *p_unique, int64_t *census_scratch, void *stream) {
    HV_REQUIRE(L >= 1 && L <= HIDVAE_MAX_LEVELS && B >= 1 && K >= 1, "bottleneck_fwd: bad sizes");
    HV_REQUIRE(h1 && W2 && W3 && h2 && y && cb_eff && cc && ids && emb_sum && Wd0 && Wd1 && d0 && d1 && pre2 && pre_d0 && pre_d1,
               "bottleneck_fwd: null pointer");
    HV_REQUIRE(mode == HIDVAE_MODE_STE || mode == HIDVAE_MODE_ROTATION, "bottleneck_fwd: mode %d is not fused", mode);
    HV_REQUIRE(K2 % 16 == 0 && N2 % 16 == 0 && Nd0 % 16 == 0 && Nd1 % 16 == 0 && K2 >= 16 && N2 >= 16 && Nd0 >= 16 && Nd1 >= 16 &&
                   K2 <= BN_HMAX && N2 <= BN_HMAX && Nd0 <= BN_HMAX,
               "bottleneck_fwd: layer widths must be multiples of 16 and at most %d (K2=%d N2=%d Nd0=%d Nd1=%d)", BN_HMAX, K2, N2, Nd0, Nd1);
    HV_REQUIRE(emb_cat == nullptr || (ld_cat >= (int64_t)L * D && ld_cat % 4 == 0), "bottleneck_fwd: ld_cat=%lld", (long long)ld_cat);
    BneckArgs b{};
    FwdArgs &a = b.rq;
    a.y = nullptr; a.B = B; a.normalize_input = normalize_input; a.cb_eff = cb_eff; a.cc = cc; a.L = L; a.K = K;
    const int64_t Kp = hv_cdiv(K, 32 * BN_WAVES) * (32 * BN_WAVES);  // every wave scans its share, 32 codes at a time (padding: +inf)
    a.KC = (int)Kp;
    a.nchunks = 1;
    a.beta = beta; a.z = z; a.ids = ids; a.emb_cat = emb_cat; a.ld_cat = ld_cat; a.emb_sum = emb_sum; a.res_cat = nullptr; a.qloss = qloss;
    const size_t lds = level_lds_bytes(a.KC) * (size_t)L + (size_t)(2 * (BN_HMAX / 16) * 64 + 2 * 64) * sizeof(float4);
    HV_REQUIRE(a.KC % (32 * BN_WAVES) == 0 && a.KC <= MAX_KC && lds <= 160 * 1024 - 1024,
               "bottleneck_fwd: the codebooks (L=%d, K=%lld) do not fit in LDS beside the activations", L, (long long)K);
    HV_REQUIRE((embs_norm == nullptr) == (census_scratch == nullptr) && (p_unique == nullptr) == (census_scratch == nullptr),
               "bottleneck_fwd: embs_norm, p_unique and census_scratch come together or not at all");
    HV_REQUIRE(census_scratch == nullptr || (L <= 4 && K <= 1024 && emb_cat != nullptr),
               "bottleneck_fwd: the fused id census packs 10 bits per level into 40 (L=%d, K=%lld)", L, (long long)K);
    a.embs_norm = embs_norm; a.p_unique = p_unique;
    a.census = reinterpret_cast<unsigned long long *>(census_scratch);
    a.census_size = 4 * B;
    b.h1 = h1; b.K2 = K2; b.N2 = N2; b.W2 = W2; b.W3 = W3; b.pre2 = pre2; b.h2 = h2; b.y_out = y;
    b.Nd0 = Nd0; b.Nd1 = Nd1; b.Wd0 = Wd0; b.Wd1 = Wd1; b.pre_d0 = pre_d0; b.d0 = d0; b.pre_d1 = pre_d1; b.d1 = d1;
    const int grid = (int)hv_cdiv(B, ITEMS_PER_WAVE);
    hipStream_t s = (hipStream_t)stream;
    if (mode == HIDVAE_MODE_STE) {
        auto kern = bottleneck_fwd_kernel<HIDVAE_MODE_STE>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * BN_WAVES), lds, s, b);
    } else {
        auto kern = bottleneck_fwd_kernel<HIDVAE_MODE_ROTATION>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * BN_WAVES), lds, s, b);
    }
    HV_LAUNCH_CHECK("bottleneck_fwd");
    return HIDVAE_OK;
}

extern "C" int hidvae_rq_backward(const float *y, const float *z, int64_t B, int normalize_input, const float *cb_eff,
                                  const float *cc, int L, int64_t K, int mode, float beta, const int64_t *ids,
                                  const float *g_cat, int64_t ld_gcat, const float *g_sum, const float *g_z_in,
                                  int64_t g_z_rows, float gq, const float *gq_items, int64_t gq_stride, float *g_y, float *dE_rows, int embed_dim, void *stream) {
    HV_REQUIRE(L >= 1 && L <= HIDVAE_MAX_LEVELS, "rq_backward: n_layers=%d not in [1,%d]", L, HIDVAE_MAX_LEVELS);
    HV_REQUIRE(B >= 1 && K >= 1 && z && cb_eff && cc && ids && g_y, "rq_backward: bad arguments");
    HV_REQUIRE(!normalize_input || y, "rq_backward: normalize_input needs y");
    HV_REQUIRE(mode == HIDVAE_MODE_STE || mode == HIDVAE_MODE_ROTATION, "rq_backward: mode %d is not fused", mode);
    if (embed_dim != D) {
        HV_REQUIRE(hv_rqg_dim_ok(embed_dim), "rq_backward: embed_dim=%d (a multiple of 4, at most 64)", embed_dim);
        HV_REQUIRE(g_cat == nullptr || ld_gcat >= (int64_t)L * embed_dim, "rq_backward: ld_gcat=%lld", (long long)ld_gcat);
        return hv_rqg_backward(y, z, B, normalize_input, cb_eff, cc, L, K, embed_dim, mode, beta, ids, g_cat, ld_gcat, g_sum, g_z_in, g_z_rows, gq,
                               gq_items, gq_stride, g_y, dE_rows, (hipStream_t)stream);
    }
    HV_REQUIRE(g_cat == nullptr || (ld_gcat >= (int64_t)L * D && ld_gcat % 4 == 0), "rq_backward: ld_gcat=%lld", (long long)ld_gcat);
    BwdArgs a{y, z, B, normalize_input, cb_eff, cc, K, beta, ids, g_cat, ld_gcat, g_sum, g_z_in, g_z_rows, gq, gq_items, gq_stride, g_y, dE_rows};
    const int grid = (int)hv_cdiv(B, ITEMS_PER_WG);
    if (mode == HIDVAE_MODE_STE) return launch_bwd<HIDVAE_MODE_STE>(a, L, grid, (hipStream_t)stream);
    return launch_bwd<HIDVAE_MODE_ROTATION>(a, L, grid, (hipStream_t)stream);
}

extern "C" int hidvae_rq_backward_slices(const float *y, const float *z, int64_t B, int normalize_input, const float *cb_eff, const float *cc,
                                         int L, int64_t K, int mode, float beta, const int64_t *ids, const float *const *g_slices_host,
                                         const int32_t *slice_width_host, int n_slices, const float *g_sum, const float *g_z_in,
                                         int64_t g_z_rows, float gq, const float *gq_items, int64_t gq_stride, float *g_y, float *dE_rows,
                                         int embed_dim, void *stream) {
    HV_REQUIRE(L >= 1 && L <= HIDVAE_MAX_LEVELS, "rq_backward_slices: n_layers=%d not in [1,%d]", L, HIDVAE_MAX_LEVELS);
    HV_REQUIRE(B >= 1 && K >= 1 && z && cb_eff && cc && ids && g_y, "rq_backward_slices: bad arguments");
    HV_REQUIRE(!normalize_input || y, "rq_backward_slices: normalize_input needs y");
    HV_REQUIRE(mode == HIDVAE_MODE_STE || mode == HIDVAE_MODE_ROTATION, "rq_backward_slices: mode %d is not fused", mode);
    HV_REQUIRE(embed_dim == D, "rq_backward_slices: embed_dim=%d (the width-independent kernels take the summed gradient: hidvae_sum_prefix_slices first)", embed_dim);
    HV_REQUIRE(n_slices >= 1 && n_slices <= 2 * HIDVAE_MAX_LEVELS && g_slices_host && slice_width_host, "rq_backward_slices: %d slices (1..%d)", n_slices,
               2 * HIDVAE_MAX_LEVELS);
    BwdArgs a{y, z, B, normalize_input, cb_eff, cc, K, beta, ids, nullptr, 0, g_sum, g_z_in, g_z_rows, gq, gq_items, gq_stride, g_y, dE_rows};
    a.n_slices = n_slices;
    for (int s = 0; s < n_slices; s++) {
        HV_REQUIRE(g_slices_host[s] != nullptr && slice_width_host[s] >= D && slice_width_host[s] <= L * D && slice_width_host[s] % D == 0,
                   "rq_backward_slices: slice %d has width %d (a multiple of %d up to %d)", s, slice_width_host[s], D, L * D);
        a.slice[s] = g_slices_host[s];
        a.slice_w[s] = slice_width_host[s];
    }
    const int grid = (int)hv_cdiv(B, ITEMS_PER_WG);
    if (mode == HIDVAE_MODE_STE) return launch_bwd<HIDVAE_MODE_STE>(a, L, grid, (hipStream_t)stream);
    return launch_bwd<HIDVAE_MODE_ROTATION>(a, L, grid, (hipStream_t)stream);
}

extern "C" int hidvae_codebook_grad(const int64_t *ids, const float *dE_rows, int64_t B, int L, int64_t K,
                                    const float *const *E_host, const float *cb_eff, const int32_t *normalize_host,
                                    float *const *gE_host, int accumulate, float *workspace, int embed_dim, void *stream) {
    HV_REQUIRE(L >= 1 && L <= HIDVAE_MAX_LEVELS && B >= 1 && K >= 1, "codebook_grad: bad sizes");
    HV_REQUIRE(ids && dE_rows && E_host && cb_eff && gE_host, "codebook_grad: null pointer");
    if (embed_dim != D) {
        HV_REQUIRE(hv_rqg_dim_ok(embed_dim), "codebook_grad: embed_dim=%d (a multiple of 4, at most 64)", embed_dim);
        return hv_rqg_codebook_grad(ids, dE_rows, B, L, K, embed_dim, E_host, cb_eff, normalize_host, gE_host, accumulate, (hipStream_t)stream);
    }
    CbGradArgs a{};
    a.ids = ids; a.dE_rows = dE_rows; a.B = B; a.L = L; a.K = K; a.cb_eff = cb_eff; a.accumulate = accumulate;
    for (int i = 0; i < L; i++) {
        a.E[i] = E_host[i];
        a.normalize[i] = normalize_host ? normalize_host[i] : 0;
        a.gE[i] = gE_host[i];
    }
    // one wave per (level, code) walks the items in ascending order: fine for a training batch, but the longest code's chain sets
    // the time (47 us at B = 8192 with the skewed ids of an untrained model).  With a workspace, batches above 2048 items are
    // summed in slabs of 2048 by (level, code, slab) waves and the slabs added in ascending order by a second launch.
    a.slab_items = 2048;
    a.slabs = (workspace != nullptr && B > a.slab_items) ? (int)hv_cdiv(B, a.slab_items) : 1;
    if (a.slabs == 1) a.slab_items = B;
    a.partial = workspace;
    const dim3 grid((unsigned)hv_cdiv((int64_t)L * K, 4), (unsigned)a.slabs);
    if (K % 4 == 0) hipLaunchKernelGGL(codebook_grad_kernel<true>, grid, dim3(WG_THREADS), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(codebook_grad_kernel<false>, grid, dim3(WG_THREADS), 0, (hipStream_t)stream, a);
    HV_LAUNCH_CHECK("codebook_grad");
    if (a.slabs > 1) {
        hipLaunchKernelGGL(codebook_grad_final_kernel, dim3(grid.x), dim3(WG_THREADS), 0, (hipStream_t)stream, a);
        HV_LAUNCH_CHECK("codebook_grad_final");
    }
    return HIDVAE_OK;
}

extern "C" int hidvae_l2norm32_fwd(const float *x, int64_t M, int64_t ldx, float eps, float *out, int64_t ldo, float *norms,
                                   void *stream) {
    HV_REQUIRE(x && out && M >= 1 && ldx >= 32 && ldo >= 32 && ldx % 4 == 0 && ldo % 4 == 0, "l2norm32: bad arguments");
    hipLaunchKernelGGL(l2norm32_kernel, dim3((unsigned)hv_cdiv(M, ITEMS_PER_WG)), dim3(WG_THREADS), 0, (hipStream_t)stream, x, M,
                       ldx, eps, out, ldo, norms);
    HV_LAUNCH_CHECK("l2norm32");
    return HIDVAE_OK;
}
