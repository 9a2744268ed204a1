// The attention gate of TagPredictor (reference modules/h_rqvae.py:128-139 built, :196-206 applied), one row-local launch each way:
//     a1 = relu(x W0^T + b0)   [E -> E/4]       a2 = gelu(a1 W2^T + b2)   [E/4 -> E/2]       a3 = sigmoid(a2 W4^T + b4)   [E/2 -> E]
//     h  = x * a3              and, for levels > 0,   h <- h / max(|h|, 1e-12)              (h_rqvae.py:203-206)
// E = 32 (i+1) <= 128: the three Linears together are 0.24 kFLOP ... 16 kFLOP per item -- as five launches forward (three GEMMs,
// the product, the normalisation) and nine backward they were launch latency and nothing else (round 2: 4-6 us each, 14 launches
// per level and step).  Here one wave owns a row: the weights sit transposed in LDS, a lane owns an output column and walks k.
// The backward returns the gradient with respect to x through BOTH of its uses (gate input and attention input: the two are views
// of the same columns of emb_cat) and the three pre-activation gradients, from which one grouped launch forms dW / db.
#include "common.h"

namespace {

constexpr int GATE_ROWS = 4;  // rows per workgroup (one per wave: 256 workgroups at 1024 rows; two per wave left half the CUs without one)

struct GateArgs {
    const float *x; int64_t ldx;
    int64_t B;
    int E, H1, H2;
    const float *W0, *b0, *W2, *b2, *W4, *b4;  // [H1,E], [H2,H1], [E,H2]
    int normalize;
    float eps;
    // forward outputs / backward inputs
    float *a1, *pre2, *a2, *a3, *h, *nrm;
    // backward
    const float *gh; int64_t ldgh;
    float *gx, *g3, *g2, *g1;
};

__device__ __forceinline__ float gate_dsigmoid(float s) { return s * (1.0f - s); }

__global__ __launch_bounds__(256) void gate_fwd_kernel(GateArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int E = a.E, H1 = a.H1, H2 = a.H2;
    float *Wt0 = sm;                 // [E][H1]   Wt0[k * H1 + j] = W0[j][k]
    float *Wt2 = Wt0 + E * H1;       // [H1][H2]
    float *Wt4 = Wt2 + H1 * H2;      // [H2][E]
    float *bb = Wt4 + H2 * E;        // b0 | b2 | b4
    float *rowbuf = bb + (H1 + H2 + E);  // per wave: x[E] | a1[H1] | a2[H2]
    for (int i = threadIdx.x; i < H1 * E; i += 256) { const int j = i / E, k = i - j * E; Wt0[k * H1 + j] = a.W0[i]; }
    for (int i = threadIdx.x; i < H2 * H1; i += 256) { const int j = i / H1, k = i - j * H1; Wt2[k * H2 + j] = a.W2[i]; }
    for (int i = threadIdx.x; i < E * H2; i += 256) { const int j = i / H2, k = i - j * H2; Wt4[k * E + j] = a.W4[i]; }
    for (int i = threadIdx.x; i < H1; i += 256) bb[i] = a.b0[i];
    for (int i = threadIdx.x; i < H2; i += 256) bb[H1 + i] = a.b2[i];
    for (int i = threadIdx.x; i < E; i += 256) bb[H1 + H2 + i] = a.b4[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *xr = rowbuf + wave * (E + H1 + H2), *a1r = xr + E, *a2r = a1r + H1;
    __syncthreads();
    for (int rr = 0; rr < GATE_ROWS / 4; rr++) {
        const int64_t row = (int64_t)blockIdx.x * GATE_ROWS + rr * 4 + wave;
        const bool live = row < a.B;
        const int64_t src = live ? row : a.B - 1;
        float xv[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int c = lane + 64 * t;
            xv[t] = c < E ? a.x[src * a.ldx + c] : 0.0f;
            if (c < E) xr[c] = xv[t];
        }
        __syncthreads();
        if (lane < H1) {
            float acc = bb[lane];
            for (int k = 0; k < E; k++) acc = fmaf(xr[k], Wt0[k * H1 + lane], acc);
            const float v = fmaxf(acc, 0.0f);
            a1r[lane] = v;
            if (live) a.a1[row * H1 + lane] = v;
        }
        __syncthreads();
        if (lane < H2) {
            float acc = bb[H1 + lane];
            for (int k = 0; k < H1; k++) acc = fmaf(a1r[k], Wt2[k * H2 + lane], acc);
            const float v = hv_gelu(acc);
            a2r[lane] = v;
            if (live) { a.pre2[row * H2 + lane] = acc; a.a2[row * H2 + lane] = v; }
        }
        __syncthreads();
        float u[2], s3[2];
        float ss = 0.0f;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int c = lane + 64 * t;
            u[t] = 0.0f; s3[t] = 0.0f;
            if (c < E) {
                float acc = bb[H1 + H2 + c];
                for (int k = 0; k < H2; k++) acc = fmaf(a2r[k], Wt4[k * E + c], acc);
                s3[t] = hv_sigmoid(acc);
                u[t] = xv[t] * s3[t];
                ss += u[t] * u[t];
            }
        }
        float den = 1.0f;
        if (a.normalize) {
            ss = hv_wave_sum(ss);
            const float nrm = sqrtf(ss);
            den = fmaxf(nrm, a.eps);
            if (live && lane == 0) a.nrm[row] = nrm;
        }
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int c = lane + 64 * t;
            if (c < E && live) {
                a.a3[row * E + c] = s3[t];
                a.h[row * E + c] = a.normalize ? u[t] / den : u[t];
            }
        }
        __syncthreads();  // the row buffers are rewritten by the next row
    }
}

__global__ __launch_bounds__(256) void gate_bwd_kernel(GateArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int E = a.E, H1 = a.H1, H2 = a.H2;
    float *W0s = sm;                 // [H1][E]  (as stored: the backward products walk the OUTPUT index of the forward layer)
    float *W2s = W0s + H1 * E;       // [H2][H1]
    float *W4s = W2s + H2 * H1;      // [E][H2]
    float *rowbuf = W4s + E * H2;    // per wave: g3[E] | g2[H2] | g1[H1]
    for (int i = threadIdx.x; i < H1 * E; i += 256) W0s[i] = a.W0[i];
    for (int i = threadIdx.x; i < H2 * H1; i += 256) W2s[i] = a.W2[i];
    for (int i = threadIdx.x; i < E * H2; i += 256) W4s[i] = a.W4[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *g3r = rowbuf + wave * (E + H1 + H2), *g2r = g3r + E, *g1r = g2r + H2;
    __syncthreads();
    for (int rr = 0; rr < GATE_ROWS / 4; rr++) {
        const int64_t row = (int64_t)blockIdx.x * GATE_ROWS + rr * 4 + wave;
        const bool live = row < a.B;
        const int64_t src = live ? row : a.B - 1;
        float xv[2], s3[2], gv[2], u[2];
        float dot = 0.0f;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int c = lane + 64 * t;
            xv[t] = s3[t] = gv[t] = u[t] = 0.0f;
            if (c < E) {
                xv[t] = a.x[src * a.ldx + c];
                s3[t] = a.a3[src * E + c];
                gv[t] = a.gh[src * a.ldgh + c];
                u[t] = xv[t] * s3[t];
            }
        }
        float gu[2] = {gv[0], gv[1]};
        if (a.normalize) {  // F.normalize backward: (g - hn (hn . g)) / |u|, or g / eps where the norm was clamped (l2norm_bwd_kernel)
            const float nrm = a.nrm[src];
            const float den = fmaxf(nrm, a.eps);
            float hn[2];
#pragma unroll
            for (int t = 0; t < 2; t++) { hn[t] = u[t] / den; dot += hn[t] * gv[t]; }
            dot = hv_wave_sum(dot);
            const float proj = nrm > a.eps ? dot : 0.0f;
#pragma unroll
            for (int t = 0; t < 2; t++) gu[t] = (gv[t] - hn[t] * proj) / den;
        }
        float gxg[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int c = lane + 64 * t;
            gxg[t] = gu[t] * s3[t];                                  // through the gate input
            const float g3 = (gu[t] * xv[t]) * gate_dsigmoid(s3[t]);  // through the attention output, at the sigmoid's pre-activation
            if (c < E) {
                g3r[c] = g3;
                if (live) a.g3[row * E + c] = g3;
            }
        }
        __syncthreads();
        if (lane < H2) {
            float acc = 0.0f;
            for (int i = 0; i < E; i++) acc = fmaf(g3r[i], W4s[i * H2 + lane], acc);
            const float g2 = acc * hv_dgelu(a.pre2[src * H2 + lane]);
            g2r[lane] = g2;
            if (live) a.g2[row * H2 + lane] = g2;
        }
        __syncthreads();
        if (lane < H1) {
            float acc = 0.0f;
            for (int i = 0; i < H2; i++) acc = fmaf(g2r[i], W2s[i * H1 + lane], acc);
            const float g1 = a.a1[src * H1 + lane] > 0.0f ? acc : 0.0f;
            g1r[lane] = g1;
            if (live) a.g1[row * H1 + lane] = g1;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int c = lane + 64 * t;
            if (c < E) {
                float acc = 0.0f;
                for (int i = 0; i < H1; i++) acc = fmaf(g1r[i], W0s[i * E + c], acc);
                if (live) a.gx[row * E + c] = gxg[t] + acc;
            }
        }
        __syncthreads();
    }
}

size_t gate_lds_bytes(int E, int H1, int H2, bool fwd) {
    const size_t w = (size_t)E * H1 + (size_t)H1 * H2 + (size_t)H2 * E;
    return (w + (fwd ? (size_t)(E + H1 + H2) : 0) + 4 * (size_t)(E + H1 + H2)) * sizeof(float);
}

}  // namespace

extern "C" int hidvae_gate_fwd(const float *x, int64_t ldx, int64_t B, int E, const float *W0, const float *b0, const float *W2,
                               const float *b2, const float *W4, const float *b4, int normalize, float eps, float *a1, float *pre2,
                               float *a2, float *a3, float *h, float *nrm, void *stream) {
    HV_REQUIRE(B >= 1 && E >= 4 && E <= 128 && E % 4 == 0, "gate_fwd: B=%lld E=%d (E must be a multiple of 4, at most 128)", (long long)B, E);
    HV_REQUIRE(x && W0 && b0 && W2 && b2 && W4 && b4 && a1 && pre2 && a2 && a3 && h && ldx >= E, "gate_fwd: bad arguments");
    HV_REQUIRE(!normalize || nrm != nullptr, "gate_fwd: normalize needs the norms output");
    GateArgs a{};
    a.x = x; a.ldx = ldx; a.B = B; a.E = E; a.H1 = E / 4; a.H2 = E / 2;
    a.W0 = W0; a.b0 = b0; a.W2 = W2; a.b2 = b2; a.W4 = W4; a.b4 = b4;
    a.normalize = normalize; a.eps = eps; a.a1 = a1; a.pre2 = pre2; a.a2 = a2; a.a3 = a3; a.h = h; a.nrm = nrm;
    const size_t lds = gate_lds_bytes(E, a.H1, a.H2, true);
    hipLaunchKernelGGL(gate_fwd_kernel, dim3((unsigned)hv_cdiv(B, GATE_ROWS)), dim3(256), lds, (hipStream_t)stream, a);
    HV_LAUNCH_CHECK("gate_fwd");
    return HIDVAE_OK;
}

extern "C" int hidvae_gate_bwd(const float *gh, int64_t ldgh, const float *x, int64_t ldx, int64_t B, int E, const float *W0,
                               const float *W2, const float *W4, int normalize, float eps, const float *a1, const float *pre2,
                               const float *a3, const float *nrm, float *gx, float *g3, float *g2, float *g1, void *stream) {
    HV_REQUIRE(B >= 1 && E >= 4 && E <= 128 && E % 4 == 0, "gate_bwd: B=%lld E=%d", (long long)B, E);
    HV_REQUIRE(gh && x && W0 && W2 && W4 && a1 && pre2 && a3 && gx && g3 && g2 && g1 && ldx >= E && ldgh >= E, "gate_bwd: bad arguments");
    HV_REQUIRE(!normalize || nrm != nullptr, "gate_bwd: normalize needs the saved norms");
    GateArgs a{};
    a.x = x; a.ldx = ldx; a.B = B; a.E = E; a.H1 = E / 4; a.H2 = E / 2;
    a.W0 = W0; a.W2 = W2; a.W4 = W4; a.normalize = normalize; a.eps = eps;
    a.a1 = const_cast<float *>(a1); a.pre2 = const_cast<float *>(pre2); a.a3 = const_cast<float *>(a3); a.nrm = const_cast<float *>(nrm);
    a.gh = gh; a.ldgh = ldgh; a.gx = gx; a.g3 = g3; a.g2 = g2; a.g1 = g1;
    const size_t lds = gate_lds_bytes(E, a.H1, a.H2, false);
    hipLaunchKernelGGL(gate_bwd_kernel, dim3((unsigned)hv_cdiv(B, GATE_ROWS)), dim3(256), lds, (hipStream_t)stream, a);
    HV_LAUNCH_CHECK("gate_bwd");
    return HIDVAE_OK;
}
