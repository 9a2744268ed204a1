// OPT-IN bf16 throughput mode for the large-batch Linear layers (BASELINE config 2 is quoted "bf16"; the reference plumbs mixed
// precision at train_hidvae.py:77,80,186-189 although every shipped config runs with amp=False).  Operands stay fp32 in HBM -- master
// weights, activations and gradients are what they always were -- and are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32)
// on their way into LDS; products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the epilogue (bias, activation, dropout
// mask, accumulate) is the fp32 code of gemm.hip.  NOT bit-compatible with the fp32 path and never selected by default: parity is
// claimed on fp32.
//
// Why this shape: at batch >= 4096 the fp32 LDS-tiled kernel (64x64 tiles) plateaus at ~92 TFLOP/s because every workgroup pulls
// (64+64) x K floats through L2 for 64x64xK products: 16 FLOP per L2 byte, and L2->CU delivers ~5.7 TB/s on this access pattern
// (DESIGN.md).  A bf16 MFMA is 16x cheaper than the fp32 one, so arithmetic stops mattering and the tile is sized for L2 traffic
// instead: 128x128 per workgroup = 32 FLOP per L2 byte, each of the 4 waves holding a 64x64 block as 2x2 MFMA tiles (64 accumulator
// registers), LDS holding bf16 (half the bytes of the fp32 image: 80-byte rows = conflict-free 128-bit fragment reads).
#include "common.h"

namespace {

constexpr int TB = 128;   // workgroup tile (rows and columns)
constexpr int KB = 32;    // k per LDS tile (two 32x32x16 MFMA steps)
constexpr int ROW = 40;   // bf16 per LDS row: 32 + 8 padding (80 bytes)
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

struct BfArgs {
    int64_t M, N, K;
    const float *A;
    int64_t lda;
    const float *B;
    int64_t ldb;
    const float *bias;
    float *C;
    int64_t ldc;
    int epilogue;
    float *aux;
    int64_t ldaux;
    int accumulate;
    const float *mask;
    int64_t ldmask;
    float mask_scale;
    int64_t k_per_split;  // multiple of KB
    float *partial;       // [splits][M][N] or nullptr
};

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 v;
    v[0] = (__bf16)lo;  // hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) for the pair
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

// one operand tile [TB rows x KB k] global fp32 -> registers (16 floats per thread) -> LDS bf16 [row][ROW].
// Rows past the end of the matrix are CLAMPED to the last row instead of zero-filled: an A row past M only feeds C rows past M (a B row
// past N only C columns past N), which are never stored -- so the full-tile path is branch-free.  Only the ragged last k-tile takes
// the per-element path.
template <bool KCONTIG>
struct BfLoader {
    float v[1][4][4];
    const float *base[4];  // this thread's four slots at k = 0
    int64_t ld;
    __device__ __forceinline__ void init(const float *P, int64_t ld_, int64_t row0, int64_t nrows, int tid) {
        ld = ld_;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int idx = tid + 256 * s;
            const int r = KCONTIG ? idx >> 3 : idx & 127, k4 = KCONTIG ? idx & 7 : idx >> 7;
            int64_t row = row0 + r;
            if (row >= nrows) row = nrows - 1;
            base[s] = KCONTIG ? P + row * ld + 4 * k4 : P + (int64_t)(4 * k4) * ld + row;
        }
    }
    __device__ __forceinline__ void load_full(int st, int64_t k0) {  // k0 + KB <= kend
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (KCONTIG) {
                const f4u t = *reinterpret_cast<const f4u *>(base[s] + k0);
                v[st][s][0] = t[0]; v[st][s][1] = t[1]; v[st][s][2] = t[2]; v[st][s][3] = t[3];
            } else {
                const float *p = base[s] + k0 * ld;
                v[st][s][0] = p[0]; v[st][s][1] = p[ld]; v[st][s][2] = p[2 * ld]; v[st][s][3] = p[3 * ld];
            }
        }
    }
    __device__ __forceinline__ void load_tail(int st, int64_t k0, int64_t kend, int tid) {  // the ragged last k-tile
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int idx = tid + 256 * s;
            const int k4 = KCONTIG ? idx & 7 : idx >> 7;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int64_t k = k0 + 4 * k4 + j;
                v[st][s][j] = k < kend ? (KCONTIG ? base[s][k0 + j] : base[s][(k0 + j) * ld]) : 0.0f;
            }
        }
    }
    __device__ __forceinline__ void store(int st, unsigned short *T, int tid) const {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int idx = tid + 256 * s;
            const int r = KCONTIG ? idx >> 3 : idx & 127, k4 = KCONTIG ? idx & 7 : idx >> 7;
            uint2 w;
            w.x = pack_bf16(v[st][s][0], v[st][s][1]);
            w.y = pack_bf16(v[st][s][2], v[st][s][3]);
            *reinterpret_cast<uint2 *>(T + r * ROW + 4 * k4) = w;
        }
    }
};

__device__ __forceinline__ float bf_epilogue(int epi, float v, const float *aux, int64_t off) {
    switch (epi) {
        case HIDVAE_EPI_SILU: return hv_silu(v);
        case HIDVAE_EPI_RELU: return fmaxf(v, 0.0f);
        case HIDVAE_EPI_GELU: return hv_gelu(v);
        case HIDVAE_EPI_SIGMOID: return hv_sigmoid(v);
        case HIDVAE_EPI_DSILU: return v * hv_dsilu(aux[off]);
        case HIDVAE_EPI_DRELU: return aux[off] > 0.0f ? v : 0.0f;
        case HIDVAE_EPI_DGELU: return v * hv_dgelu(aux[off]);
        case HIDVAE_EPI_DSIGMOID: { const float s = aux[off]; return v * (s * (1.0f - s)); }
        default: return v;
    }
}

template <int LAYOUT>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(BfArgs g) {
    constexpr bool A_KC = (LAYOUT != HIDVAE_GEMM_TN);
    constexpr bool B_KC = (LAYOUT == HIDVAE_GEMM_NT);
    __shared__ __attribute__((aligned(16))) unsigned short As[2][TB * ROW];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][TB * ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;
    // XCD-aware order (see gemm.hip xcd_tile): each XCD's L2 holds one band of the longer output dimension
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int nbx = gridDim.x, nby = gridDim.y, total = nbx * nby;
        if ((total & 7) == 0) {
            const int id = by * nbx + bx;
            const int t = (id & 7) * (total >> 3) + (id >> 3);
            if (g.M >= g.N) { by = t / nbx; bx = t - by * nbx; }
            else { bx = t / nby; by = t - bx * nby; }
        }
    }
    const int64_t m0 = (int64_t)by * TB, n0 = (int64_t)bx * TB;
    const int64_t kbeg = (int64_t)blockIdx.z * g.k_per_split;
    const int64_t kend = (kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K;
    f32x16 c00, c01, c10, c11;  // (named, not an array: an indexed array of accumulators ends up in scratch)
#pragma unroll
    for (int r = 0; r < 16; r++) c00[r] = c01[r] = c10[r] = c11[r] = 0.0f;
    BfLoader<A_KC> la;
    BfLoader<B_KC> lb;
    la.init(g.A, g.lda, m0, g.M, tid);
    lb.init(g.B, g.ldb, n0, g.N, tid);
    auto fetch = [&](int st, int64_t k0) {  // (uniform branches: the whole workgroup takes the same path; past kend: nothing)
        if (k0 >= kend) return;
        if (k0 + KB <= kend) {
            la.load_full(st, k0);
            lb.load_full(st, k0);
        } else {
            la.load_tail(st, k0, kend, tid);
            lb.load_tail(st, k0, kend, tid);
        }
    };
    auto mma = [&](int buf) {
        const unsigned short *ap = As[buf] + (wm * 64 + r32) * ROW + 8 * h;
        const unsigned short *bp = Bs[buf] + (wn * 64 + r32) * ROW + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const bf16x8_t a0 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(ap + ks * 16));
            const bf16x8_t a1 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(ap + 32 * ROW + ks * 16));
            const bf16x8_t b0 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(bp + ks * 16));
            const bf16x8_t b1 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(bp + 32 * ROW + ks * 16));
            c00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c11, 0, 0, 0);
        }
    };
    // (tried: two register stages so that tile t+2 is requested before tile t's MFMAs -- the forward layer went 73 -> 61 us, the
    //  backward products got slower (252-365 VGPRs) and the B = 8192 step as a whole 1.16 -> 1.24 ms; not kept)
    fetch(0, kbeg);
    la.store(0, As[0], tid);
    lb.store(0, Bs[0], tid);
    __syncthreads();
    int buf = 0;
    for (int64_t k0 = kbeg; k0 < kend; k0 += KB) {
        fetch(0, k0 + KB);
        mma(buf);
        if (k0 + KB < kend) {
            la.store(0, As[buf ^ 1], tid);
            lb.store(0, Bs[buf ^ 1], tid);
        }
        __syncthreads();
        buf ^= 1;
    }
    // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    auto emit = [&](const f32x16 &c, int i, int j) {
        const int64_t col = n0 + wn * 64 + j * 32 + r32;
        if (col >= g.N) return;
        const float bias = (g.bias != nullptr && g.partial == nullptr) ? g.bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < g.M) {
                if (g.partial != nullptr) {
                    g.partial[((int64_t)blockIdx.z * g.M + row) * g.N + col] = c[r];
                } else {
                    float v = c[r] + bias;
                    if (g.aux != nullptr && g.epilogue < HIDVAE_EPI_DSILU && g.epilogue != HIDVAE_EPI_NONE) g.aux[row * g.ldaux + col] = v;
                    v = bf_epilogue(g.epilogue, v, g.aux, row * g.ldaux + col);
                    if (g.mask != nullptr) v = v * (g.mask[row * g.ldmask + col] * g.mask_scale);
                    float *dst = g.C + row * g.ldc + col;
                    *dst = g.accumulate ? *dst + v : v;
                }
            }
        }
    };
    emit(c00, 0, 0);
    emit(c01, 0, 1);
    emit(c10, 1, 0);
    emit(c11, 1, 1);
}

// fixed-order reduction of the split-K slabs + the epilogue
__global__ __launch_bounds__(256) void bf16_splitk_reduce_kernel(BfArgs g, int splits) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= g.M * g.N) return;
    const int64_t row = idx / g.N, col = idx - row * g.N;
    float v = g.partial[idx];
    for (int s = 1; s < splits; s++) v += g.partial[(int64_t)s * g.M * g.N + idx];
    v += g.bias != nullptr ? g.bias[col] : 0.0f;
    if (g.aux != nullptr && g.epilogue < HIDVAE_EPI_DSILU && g.epilogue != HIDVAE_EPI_NONE) g.aux[row * g.ldaux + col] = v;
    v = bf_epilogue(g.epilogue, v, g.aux, row * g.ldaux + col);
    if (g.mask != nullptr) v = v * (g.mask[row * g.ldmask + col] * g.mask_scale);
    float *dst = g.C + row * g.ldc + col;
    *dst = g.accumulate ? *dst + v : v;
}

}  // namespace

extern "C" int hidvae_gemm_bf16(int layout, int64_t M, int64_t N, int64_t K, const float *A, int64_t lda, const float *B, int64_t ldb,
                                const float *bias, float *C, int64_t ldc, int epilogue, float *aux, int64_t ldaux, const float *mask,
                                int64_t ldmask, float mask_scale, float *workspace, int64_t workspace_floats, int accumulate, void *stream) {
    HV_REQUIRE(layout >= 0 && layout <= 2, "gemm_bf16: layout %d", layout);
    HV_REQUIRE(M >= 1 && N >= 1 && K >= 1 && A && B && C, "gemm_bf16: empty problem or null operand");
    const int64_t a_min = (layout == HIDVAE_GEMM_TN) ? M : K, b_min = (layout == HIDVAE_GEMM_NT) ? K : N;
    HV_REQUIRE(lda >= a_min && ldb >= b_min && ldc >= N, "gemm_bf16: leading dimension too small");
    HV_REQUIRE(epilogue < HIDVAE_EPI_DSILU || (aux != nullptr && ldaux >= N), "gemm_bf16: backward epilogue %d needs aux", epilogue);
    HV_REQUIRE(mask == nullptr || ldmask >= N, "gemm_bf16: ldmask");
    BfArgs g{};
    g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.bias = bias; g.C = C; g.ldc = ldc; g.epilogue = epilogue;
    g.aux = aux; g.ldaux = aux ? ldaux : 0; g.accumulate = accumulate; g.mask = mask; g.ldmask = ldmask; g.mask_scale = mask_scale;
    const int64_t tiles = hv_cdiv(M, TB) * hv_cdiv(N, TB);
    // deep-K problems with few output tiles (weight gradients: K = batch): K slabs through the workspace, summed in slab order
    int splits = 1;
    if (workspace != nullptr && tiles < 256 && K >= 1024) {
        splits = (int)hv_cdiv(512, tiles);
        if (splits > K / 256) splits = (int)(K / 256);
        if (splits > 32) splits = 32;
        while (splits > 1 && (int64_t)splits * M * N > workspace_floats) splits--;
    }
    int64_t kps = hv_cdiv(hv_cdiv(K, splits), KB) * KB;
    splits = (int)hv_cdiv(K, kps);
    g.k_per_split = kps;
    g.partial = splits > 1 ? workspace : nullptr;
    const dim3 grid((unsigned)hv_cdiv(N, TB), (unsigned)hv_cdiv(M, TB), (unsigned)splits);
    hipStream_t s = (hipStream_t)stream;
    if (layout == HIDVAE_GEMM_NT) hipLaunchKernelGGL(gemm_bf16_kernel<HIDVAE_GEMM_NT>, grid, dim3(256), 0, s, g);
    else if (layout == HIDVAE_GEMM_NN) hipLaunchKernelGGL(gemm_bf16_kernel<HIDVAE_GEMM_NN>, grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL(gemm_bf16_kernel<HIDVAE_GEMM_TN>, grid, dim3(256), 0, s, g);
    HV_LAUNCH_CHECK("gemm_bf16");
    if (splits > 1) {
        hipLaunchKernelGGL(bf16_splitk_reduce_kernel, dim3((unsigned)hv_cdiv(M * N, 256)), dim3(256), 0, s, g, splits);
        HV_LAUNCH_CHECK("gemm_bf16 split-K reduce");
    }
    return HIDVAE_OK;
}
