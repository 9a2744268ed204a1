// fp32 MFMA GEMM for gfx950 with fused epilogues: the Linear layers of the HiD-VAE tokenizer
// (modules/encoder.py:23-36 encoder/decoder MLPs; h_rqvae.py:132-188,322-331 tag heads) and their
// input / weight gradients.
//
// v_mfma_f32_32x32x2_f32: per lane ONE A value (row = lane&31, k = lane>>5) and ONE B value, 16 accumulator
// registers; the result is a k-ordered fmaf chain, so with split_k == 1 an output element is bit-identical to the
// fmaf chain oracle/exact.c runs (ORDER-G16: 16-wide k-blocks ascending, inside a block k = 0,4,8,12,1,5,9,13,...).
//
// Workgroup tile (32*WM) x (32*WN), one 32x32 accumulator per wave, BK = 16.  Both operand tiles live in LDS
// k-major (As[k][m], Bs[k][n]) so a fragment read is 32 consecutive floats per half-wave (conflict-free
// ds_read_b32) whatever the global layout was; the transposition happens on the LDS store:
//   k-contiguous source (A of NT/NN, B of NT): float4 along k -> four ds_write_b32 (LD = tile+4 keeps them 2-way)
//   m-contiguous source (A of TN, B of NN/TN): float4 along m -> one ds_write_b128
// Global loads for tile t+1 are issued before the MFMAs of tile t (register prefetch, two LDS buffers, one
// barrier per k-tile).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "rules.h"
#include "gemm_ring.h"

namespace {

constexpr int BK = 16;

struct GemmArgs {
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
    int64_t k_per_split;  // multiple of BK
    float *partial;       // split-K slabs [splits][M][N] (nullptr when split_k == 1)
    int vecA, vecB;       // 16-byte loads legal for the operand
    const float *mask;    // optional dropout keep-mask multiplied in after the activation (same shape as C)
    int64_t ldmask;
    float mask_scale;
    HvDrop drop;          // ... or the keep decision evaluated here (counter-based generator, common.h): element index row * N + col
};

// the factor Dropout applies to output element (row, col): keep-mask * scale (handed in), or the in-kernel decision
__device__ __forceinline__ bool has_dropout(const GemmArgs &g) { return g.mask != nullptr || g.drop.state != nullptr; }
__device__ __forceinline__ float dropout_factor(const GemmArgs &g, int64_t row, int64_t col) {
    if (g.mask != nullptr) return g.mask[row * g.ldmask + col] * g.mask_scale;
    return hv_drop_keep(g.drop, (unsigned long long)(row * g.N + col)) ? g.mask_scale : 0.0f;
}

// (the epilogue switch lives in common.h: gemm_ring.hip applies the same one)
__device__ __forceinline__ float apply_epilogue(int epi, float v, const float *aux, int64_t off, float scale = 1.0f) {
    return hv_apply_epilogue(epi, v, aux, off, scale);
}

// XCD-aware tile order.  Workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest), and every XCD has its own
// L2: with the identity mapping each XCD touches every row block and every column block, so each L2 pulls its own copy of both
// operands (PMC, round 1: 26.6 MB fetched for 4.7 MB of operands on the encoder's first layer).  Remapping dispatch slot i to
// tile (i % 8) * (T/8) + i / 8 gives XCD j the j-th contiguous eighth of the row-major tile list, i.e. a band of rows: its L2
// then holds one eighth of A and all of B.  A bijection when T % 8 == 0, otherwise the identity is kept.
// The band runs along the LONGER output dimension (rows if M >= N, else columns), so the operand that is split eight ways is
// the larger one and the replicated one the smaller.
__device__ __forceinline__ void xcd_tile(int nbx, int nby, bool row_bands, int &bx, int &by) {
    const int total = nbx * nby;
    if ((total & 7) != 0) return;
    const int id = by * nbx + bx;
    const int t = (id & 7) * (total >> 3) + (id >> 3);
    if (row_bands) {
        by = t / nbx;
        bx = t - by * nbx;
    } else {
        bx = t / nby;
        by = t - bx * nby;
    }
}

// Load a [ROWS x BK] operand tile (ROWS = 32*W along m or n) into registers.  KCONTIG: element (row, k) is at
// P[row*ld + k]; otherwise at P[k*ld + row].  Out-of-range elements read as 0.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // a 16-byte global load that only promises dword alignment

template <int ROWS, int NT, bool KCONTIG>
struct TileLoader {
    static constexpr int NV = (ROWS * BK / 4 + NT - 1) / NT;  // float4 slots per thread
    float4 v[1][NV];
    const float *base[NV];  // this thread's slot at k0 = 0 (nullptr: outside the matrix, reads as 0)
    int64_t kstride;        // elements to advance per unit of k0
    bool fast;              // every slot is either outside or a legal 16-byte load for any FULL k-tile

    // Addresses, bounds and alignment do not change from one k-tile to the next: settle them once, so that the full tiles of
    // the main loop cost one pointer add and one load per slot (PMC on the 65,536-row GEMM: 7.4 VALU instructions per MFMA
    // before, most of them this per-tile bookkeeping).
    __device__ __forceinline__ void init(const float *P, int64_t ld, int64_t row0, int64_t nrows, bool vec, int tid) {
        fast = vec;
        kstride = KCONTIG ? 1 : ld;
#pragma unroll
        for (int s = 0; s < NV; s++) {
            const int idx = tid + s * NT;
            base[s] = nullptr;
            if (idx < ROWS * BK / 4) {
                if (KCONTIG) {
                    const int r = idx / (BK / 4), k4 = idx % (BK / 4);
                    if (row0 + r < nrows) base[s] = P + (row0 + r) * ld + 4 * k4;
                } else {
                    const int k = idx / (ROWS / 4), r4 = idx % (ROWS / 4);
                    const int64_t row = row0 + 4 * r4;
                    if (row + 4 <= nrows) base[s] = P + (int64_t)k * ld + row;
                    else if (row < nrows) fast = false;  // a ragged row group: the generic loader handles it
                }
            }
        }
    }

    // (global 16-byte loads only need dword alignment on gfx950: rows of odd width -- the tag heads' 230 / 460 / 691 -- take them too)
    __device__ __forceinline__ void load_full(int sl, int64_t k0) {  // a k-tile that lies completely inside [0, K)
#pragma unroll
        for (int s = 0; s < NV; s++) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (base[s] != nullptr) {
                const f4u t = *reinterpret_cast<const f4u *>(base[s] + k0 * kstride);
                x = make_float4(t[0], t[1], t[2], t[3]);
            }
            v[sl][s] = x;
        }
    }

    __device__ __forceinline__ void load(int sl, const float *P, int64_t ld, int64_t row0, int64_t nrows, int64_t k0,
                                         int64_t kend, bool vec, int tid) {
#pragma unroll
        for (int s = 0; s < NV; s++) {
            const int idx = tid + s * NT;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < ROWS * BK / 4) {
                if (KCONTIG) {
                    const int r = idx / (BK / 4), k4 = idx % (BK / 4);
                    const int64_t row = row0 + r, k = k0 + 4 * k4;
                    if (row < nrows) {
                        const float *p = P + row * ld + k;
                        if (vec && k + 4 <= kend) {
                            const f4u t = *reinterpret_cast<const f4u *>(p);
                            x = make_float4(t[0], t[1], t[2], t[3]);
                        } else {
                            if (k + 0 < kend) x.x = p[0];
                            if (k + 1 < kend) x.y = p[1];
                            if (k + 2 < kend) x.z = p[2];
                            if (k + 3 < kend) x.w = p[3];
                        }
                    }
                } else {
                    const int k = idx / (ROWS / 4), r4 = idx % (ROWS / 4);
                    const int64_t kk = k0 + k, row = row0 + 4 * r4;
                    if (kk < kend) {
                        const float *p = P + kk * ld + row;
                        if (vec && row + 4 <= nrows) {
                            const f4u t = *reinterpret_cast<const f4u *>(p);
                            x = make_float4(t[0], t[1], t[2], t[3]);
                        } else {
                            if (row + 0 < nrows) x.x = p[0];
                            if (row + 1 < nrows) x.y = p[1];
                            if (row + 2 < nrows) x.z = p[2];
                            if (row + 3 < nrows) x.w = p[3];
                        }
                    }
                }
            }
            v[sl][s] = x;
        }
    }

    // LDS image: T[k][ROWS + 4]
    __device__ __forceinline__ void store(int sl, float *T, int tid) const {
        constexpr int LD = ROWS + 4;
#pragma unroll
        for (int s = 0; s < NV; s++) {
            const int idx = tid + s * NT;
            if (idx < ROWS * BK / 4) {
                if (KCONTIG) {
                    const int r = idx / (BK / 4), k4 = idx % (BK / 4);
                    T[(4 * k4 + 0) * LD + r] = v[sl][s].x;
                    T[(4 * k4 + 1) * LD + r] = v[sl][s].y;
                    T[(4 * k4 + 2) * LD + r] = v[sl][s].z;
                    T[(4 * k4 + 3) * LD + r] = v[sl][s].w;
                } else {
                    const int k = idx / (ROWS / 4), r4 = idx % (ROWS / 4);
                    *reinterpret_cast<float4 *>(T + k * LD + 4 * r4) = v[sl][s];
                }
            }
        }
    }
};

// body of the LDS-tiled kernel for workgroup tile (bx, by) of an nbx x nby grid, K slab bz; As / Bs: the workgroup's two operand
// buffers, 2 * BK * (32*WM + 4) and 2 * BK * (32*WN + 4) floats
template <int WM, int WN, int LAYOUT>
__device__ __forceinline__ void tile_body(const GemmArgs &g, int bx, int by, int nbx, int nby, int bz, float *As_, float *Bs_) {
    constexpr int BM = 32 * WM, BN = 32 * WN, NT = 64 * WM * WN;
    constexpr bool A_KC = (LAYOUT != HIDVAE_GEMM_TN);  // A[M,K] row-major
    constexpr bool B_KC = (LAYOUT == HIDVAE_GEMM_NT);  // B[N,K] row-major
    constexpr int LDA = BM + 4, LDB = BN + 4;
    float (*As)[BK * LDA] = reinterpret_cast<float (*)[BK * LDA]>(As_);
    float (*Bs)[BK * LDB] = reinterpret_cast<float (*)[BK * LDB]>(Bs_);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // (PMC at 65,536 x 768 x 512 before this mapping: 1.6 GB fetched for 203 MB of operands -- the 8 column tiles that share a
    //  row block are neighbours in dispatch order, i.e. on 8 different XCDs, and every one pulled the block from HBM itself)
    xcd_tile(nbx, nby, g.M >= g.N, bx, by);
    const int64_t m0 = (int64_t)by * BM, n0 = (int64_t)bx * BN;
    const int64_t kbeg = (int64_t)bz * g.k_per_split;
    const int64_t kend = (kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K;

    TileLoader<BM, NT, A_KC> la;
    TileLoader<BN, NT, B_KC> lb;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;

    la.init(g.A, g.lda, m0, g.M, true, tid);
    lb.init(g.B, g.ldb, n0, g.N, true, tid);
    const bool fast = la.fast && lb.fast;  // (uniform per workgroup only for la/lb separately; evaluated per thread, both paths legal)
    auto fetch = [&](int sl, int64_t k0) {  // (a k0 at or past kend loads nothing)
        if (k0 >= kend) return;
        if (fast && k0 + BK <= kend) {
            la.load_full(sl, k0);
            lb.load_full(sl, k0);
        } else {
            la.load(sl, g.A, g.lda, m0, g.M, k0, kend, true, tid);
            lb.load(sl, g.B, g.ldb, n0, g.N, k0, kend, true, tid);
        }
    };
    // (tried: a three-deep register ring so that a tile's loads have two tiles of MFMAs to arrive -- no gain at 1-2 workgroups per
    //  CU and 3 % slower at batch 8192: the kernel is not waiting on load latency)
    fetch(0, kbeg);
    la.store(0, As[0], tid);
    lb.store(0, Bs[0], tid);
    __syncthreads();
    int buf = 0;
    const int i32 = lane & 31, h = lane >> 5;
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) fetch(0, k0 + BK);
        const float *ap = As[buf] + wm * 32 + i32, *bp = Bs[buf] + wn * 32 + i32;
#pragma unroll
        for (int s = 0; s < BK / 2; s++) {
            const int kk = (s >> 1) + 8 * (s & 1) + 4 * h;  // ORDER-G16 (see kmap32)
            const float a = ap[kk * LDA];
            const float b = bp[kk * LDB];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (more) {
            la.store(0, As[buf ^ 1], tid);
            lb.store(0, Bs[buf ^ 1], tid);
        }
        __syncthreads();
        buf ^= 1;
    }

    // epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int64_t col = n0 + wn * 32 + i32;
    if (col >= g.N) return;
    const float bias = (g.bias != nullptr && g.partial == nullptr) ? g.bias[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int64_t row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= g.M) continue;
        if (g.partial != nullptr) {
            g.partial[((int64_t)bz * g.M + row) * g.N + col] = acc[r];
        } else {
            float v = acc[r] + bias;
            if (g.aux != nullptr && g.epilogue < HIDVAE_EPI_DSILU && g.epilogue != HIDVAE_EPI_NONE)
                g.aux[row * g.ldaux + col] = v;
            v = apply_epilogue(g.epilogue, v, g.aux, row * g.ldaux + col, g.mask_scale);
            if (has_dropout(g)) v = v * dropout_factor(g, row, col);
            float *dst = g.C + row * g.ldc + col;
            *dst = g.accumulate ? *dst + v : v;
        }
    }
}

template <int WM, int WN, int LAYOUT>
__global__ __launch_bounds__(64 * WM * WN) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2 * BK * (32 * WM + 4)];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BK * (32 * WN + 4)];
    tile_body<WM, WN, LAYOUT>(g, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)gridDim.y, (int)blockIdx.z, As, Bs);
}

// fixed-order reduction of the split-K slabs + the same epilogue
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g, int splits) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= g.M * g.N) return;
    const int64_t row = idx / g.N, col = idx - row * g.N;
    float v = g.partial[idx];
    for (int s = 1; s < splits; s++) v += g.partial[(int64_t)s * g.M * g.N + idx];
    v += g.bias != nullptr ? g.bias[col] : 0.0f;
    if (g.aux != nullptr && g.epilogue < HIDVAE_EPI_DSILU && g.epilogue != HIDVAE_EPI_NONE) g.aux[row * g.ldaux + col] = v;
    v = apply_epilogue(g.epilogue, v, g.aux, row * g.ldaux + col, g.mask_scale);
    if (has_dropout(g)) v = v * dropout_factor(g, row, col);
    float *dst = g.C + row * g.ldc + col;
    *dst = g.accumulate ? *dst + v : v;
}

// column sums (bias gradients) in two fixed-order passes: 64-row chunks, then the chunks in ascending order
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *X, int64_t M, int64_t N, int64_t ldx,
                                                             float *partial) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 64 + c;
    const int64_t r0 = (int64_t)blockIdx.y * 64;
    float s = 0.0f;
    if (n < N) {
#pragma unroll 4
        for (int j = 0; j < 16; j++) {
            const int64_t m = r0 + rl + 4 * j;
            if (m < M) s += X[m * ldx + n];
        }
    }
    red[rl][c] = s;
    __syncthreads();
    if (rl == 0 && n < N) partial[(int64_t)blockIdx.y * N + n] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float *partial, int64_t chunks, int64_t N, float *out,
                                                           int accumulate) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.0f;
    for (int64_t c = 0; c < chunks; c++) s += partial[c * N + n];
    out[n] = accumulate ? out[n] + s : s;
}


// one launch for the batches of a training step (M <= 16384): 32 columns x 32 row groups per workgroup, group r adds rows
// r, r+32, ... in ascending order, the groups are combined in ascending order
__global__ __launch_bounds__(1024) void colsum_one_kernel(const float *X, int64_t M, int64_t N, int64_t ldx, float *out, int accumulate) {
    __shared__ float red[32][33];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int64_t n = (int64_t)blockIdx.x * 32 + c;
    float s = 0.0f;
    if (n < N)
        for (int64_t m0 = rg; m0 < M; m0 += 8 * 32) {  // eight independent loads in flight, added in the same ascending order
            float v[8];                                 // (rows past the end contribute +0.0f: s is unchanged by them)
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = m0 + 32 * j < M ? X[(m0 + 32 * j) * ldx + n] : 0.0f;
#pragma unroll
            for (int j = 0; j < 8; j++) s += v[j];
        }
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0 && n < N) {
        float v = red[0][c];
        for (int j = 1; j < 32; j++) v += red[j][c];
        out[n] = accumulate ? out[n] + v : v;
    }
}

// ------------------------------------------------------------------------------------------------
// "direct" kernel for the small, L2-resident GEMMs of this model (M = batch ~ 1e3, K,N <= 768): one wave per
// 32x32 output tile, operands loaded from global memory STRAIGHT into the MFMA operand registers (no LDS, no
// barriers), a 3-deep register ring of 16-wide k-blocks so two blocks of loads are always in flight, and an
// optional in-workgroup split of K over SPLIT waves reduced through LDS in fixed order (backward GEMMs only).
// k-order inside a 16-block: ORDER-G16 of oracle/exact.c (0,4,8,12,1,5,...), identical in all three kernels of this file,
// so SPLIT == 1 stays bit-exact against the oracle whichever tile form the dispatcher picks.
// ------------------------------------------------------------------------------------------------
// Operands are read with raw BUFFER loads: 128-bit resource (scalar) + 32-bit lane byte offset + scalar k offset, so
// addressing costs one VGPR per operand, rows need only dword alignment for the 16-byte form, and anything past the
// end of the matrix reads as 0.  The pipelined loop below is completely branch-free: blocks past a wave's range are
// "loaded" from an out-of-range offset (zeros) and multiplied in as exact no-ops, so every path issues the same loads
// in the same order and the compiler can keep two blocks in flight behind counted vmcnt waits.
// `voff`: lane byte offset (row*ld + 8h for k-contiguous operands, 8h*ld + row otherwise, times 4).
constexpr int HV_OOB = 0x7FFFFFF0;

// k visited by MFMA step j on lane half h of the 32x32x2 form: the chain order inside a 16-block is ORDER-G16 =
// 0,4,8,12, 1,5,9,13, 2,6,10,14, 3,7,11,15 (what the 16x16x4 form does natively: step s, lane quarter q -> k = 4q + s)
__device__ __forceinline__ int kmap32(int j, int h) { return (j >> 1) + 8 * (j & 1) + 4 * h; }

template <bool KCONTIG>
__device__ __forceinline__ void load_block(__amdgpu_buffer_rsrc_t rsrc, int ld4, int voff, int k0, float (&v)[8]) {
    if (KCONTIG) {  // voff already carries +4h floats: x = k0+4h.., y = k0+8+4h..
        const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, k0 * 4, 0));
        const f32x4 y = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, k0 * 4 + 32, 0));
        v[0] = x[0]; v[1] = y[0]; v[2] = x[1]; v[3] = y[1]; v[4] = x[2]; v[5] = y[2]; v[6] = x[3]; v[7] = y[3];
    } else {        // voff carries +4h rows
#pragma unroll
        for (int j = 0; j < 8; j++)
            v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (k0 + (j >> 1) + 8 * (j & 1)) * ld4, 0));
    }
}

// ragged last block (K % 16 != 0): per-element loads, lanes whose k is past K aim out of range and read 0
template <bool KCONTIG>
__device__ __forceinline__ void load_tail(__amdgpu_buffer_rsrc_t rsrc, int ld4, int voff, int k0, int kend, int h, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int kk = (j >> 1) + 8 * (j & 1);  // + 4h is inside voff
        const bool ok = k0 + kk + 4 * h < kend;
        if (KCONTIG) v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, ok ? voff + 4 * kk : HV_OOB, k0 * 4, 0));
        else v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, ok ? voff : HV_OOB, (k0 + kk) * ld4, 0));
    }
}

// ---- 16x16 tiles (v_mfma_f32_16x16x4_f32): lane = (i = lane&15, q = lane>>4) supplies k = k0 + 4q + s at step s.
// Four times as many waves as the 32x32 form for the same problem and a 40-cycle dependent chain per 4 k instead of 64 per
// 2 k: this is the latency-optimised form the small-batch GEMMs of the step use.
template <bool KCONTIG>
__device__ __forceinline__ void load_block16(__amdgpu_buffer_rsrc_t rsrc, int ld4, int voff, int k0, float (&v)[4]) {
    if (KCONTIG) {
        const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, k0 * 4, 0));
        v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
    } else {
#pragma unroll
        for (int s = 0; s < 4; s++) v[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (k0 + s) * ld4, 0));
    }
}
template <bool KCONTIG>
__device__ __forceinline__ void load_tail16(__amdgpu_buffer_rsrc_t rsrc, int ld4, int voff, int k0, int kend, int q, float (&v)[4]) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const bool ok = k0 + 4 * q + s < kend;
        if (KCONTIG) v[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, ok ? voff + 4 * s : HV_OOB, k0 * 4, 0));
        else v[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, ok ? voff : HV_OOB, (k0 + s) * ld4, 0));
    }
}

template <int LAYOUT, int SPLIT, int NS>
__global__ __launch_bounds__(64 * SPLIT) void gemm_direct16_kernel(GemmArgs g) {
    constexpr bool A_KC = (LAYOUT != HIDVAE_GEMM_TN);
    constexpr bool B_KC = (LAYOUT == HIDVAE_GEMM_NT);
    __shared__ float part[SPLIT > 1 ? SPLIT * 256 : 1];
    const int lane = threadIdx.x & 63;
    const int w = SPLIT == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    int bx = blockIdx.x, by = blockIdx.y;
    xcd_tile(gridDim.x, gridDim.y, g.M >= g.N, bx, by);
    const int64_t m0 = (int64_t)by * 16, n0 = (int64_t)bx * 16;
    const int64_t ra = (m0 + i16 < g.M) ? m0 + i16 : g.M - 1;
    const int64_t rb = (n0 + i16 < g.N) ? n0 + i16 : g.N - 1;
    const int va = 4 * (A_KC ? (int)(ra * g.lda) + 4 * q : (int)(4 * q * g.lda + ra));
    const int vb = 4 * (B_KC ? (int)(rb * g.ldb) + 4 * q : (int)(4 * q * g.ldb + rb));
    const int lda4 = (int)g.lda * 4, ldb4 = (int)g.ldb * 4, Ki = (int)g.K;
    const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(g.A), 0, (int)(4 * (A_KC ? (g.M - 1) * g.lda + g.K : (g.K - 1) * g.lda + g.M)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(g.B), 0, (int)(4 * (B_KC ? (g.N - 1) * g.ldb + g.K : (g.K - 1) * g.ldb + g.N)), 0x00020000);
    const int nfull = Ki / 16;
    const int b_lo = nfull * w / SPLIT, b_hi = nfull * (w + 1) / SPLIT;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float ra_[NS][4], rb_[NS][4];
    auto load = [&](float (&av)[4], float (&bv)[4], int blk) {
        const bool in = blk < b_hi;
        load_block16<A_KC>(ra_rsrc, lda4, in ? va : HV_OOB, blk * 16, av);
        load_block16<B_KC>(rb_rsrc, ldb4, in ? vb : HV_OOB, blk * 16, bv);
    };
#pragma unroll
    for (int st = 0; st < NS - 1; st++) load(ra_[st], rb_[st], b_lo + st);
    for (int blk = b_lo; blk < b_hi; blk += NS) {
#pragma unroll
        for (int st = 0; st < NS; st++) {
            load(ra_[(st + NS - 1) % NS], rb_[(st + NS - 1) % NS], blk + st + NS - 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ra_[st][s4], rb_[st][s4], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (w == SPLIT - 1 && nfull * 16 < Ki) {
        load_tail16<A_KC>(ra_rsrc, lda4, va, nfull * 16, Ki, q, ra_[0]);
        load_tail16<B_KC>(rb_rsrc, ldb4, vb, nfull * 16, Ki, q, rb_[0]);
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ra_[0][s4], rb_[0][s4], acc, 0, 0, 0);
    }
    const float *mk = g.mask;
    // C/D map of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
    if (SPLIT == 1) {
        const int64_t col = n0 + i16;
        if (col >= g.N) return;
        const float bias = g.bias != nullptr ? g.bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int64_t row = m0 + 4 * q + r;
            if (row >= g.M) continue;
            float v = acc[r] + bias;
            if (g.aux != nullptr && g.epilogue < HIDVAE_EPI_DSILU && g.epilogue != HIDVAE_EPI_NONE) g.aux[row * g.ldaux + col] = v;
            v = apply_epilogue(g.epilogue, v, g.aux, row * g.ldaux + col, g.mask_scale);
            if (has_dropout(g)) v = v * dropout_factor(g, row, col);
            float *dst = g.C + row * g.ldc + col;
            *dst = g.accumulate ? *dst + v : v;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++) part[w * 256 + (4 * q + r) * 16 + i16] = acc[r];
        __syncthreads();
        for (int e = threadIdx.x; e < 256; e += 64 * SPLIT) {
            float v = part[e];
#pragma unroll
            for (int sidx = 1; sidx < SPLIT; sidx++) v += part[sidx * 256 + e];  // fixed order: bit-reproducible
            const int64_t row = m0 + (e >> 4), col = n0 + (e & 15);
            if (row >= g.M || col >= g.N) continue;
            v += g.bias != nullptr ? g.bias[col] : 0.0f;
            if (g.aux != nullptr && g.epilogue < HIDVAE_EPI_DSILU && g.epilogue != HIDVAE_EPI_NONE) g.aux[row * g.ldaux + col] = v;
            v = apply_epilogue(g.epilogue, v, g.aux, row * g.ldaux + col, g.mask_scale);
            if (has_dropout(g)) v = v * dropout_factor(g, row, col);
            float *dst = g.C + row * g.ldc + col;
            *dst = g.accumulate ? *dst + v : v;
        }
    }
}

template <int SPLIT, int NS>
void launch_direct16(int layout, const GemmArgs &g, hipStream_t s) {
    dim3 grid((unsigned)hv_cdiv(g.N, 16), (unsigned)hv_cdiv(g.M, 16));
    dim3 block(64 * SPLIT);
    if (layout == HIDVAE_GEMM_NT) hipLaunchKernelGGL((gemm_direct16_kernel<HIDVAE_GEMM_NT, SPLIT, NS>), grid, block, 0, s, g);
    else if (layout == HIDVAE_GEMM_NN) hipLaunchKernelGGL((gemm_direct16_kernel<HIDVAE_GEMM_NN, SPLIT, NS>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_direct16_kernel<HIDVAE_GEMM_TN, SPLIT, NS>), grid, block, 0, s, g);
}

// ---- two independent 16x16-tile problems in ONE launch (the dW / dX pair of a Linear layer's backward) ------------------
// Same arithmetic as gemm_direct16_kernel with the split taken at run time: wave w < split owns blocks [nfull*w/split,
// nfull*(w+1)/split) and the partials are added in ascending wave order, so a pair launch is bit-identical to the two
// separate launches it replaces.  The workgroup has max(split0, split1) waves; the surplus waves of the narrower problem
// only keep the barrier company.
template <int LAYOUT, int NS>
__device__ __forceinline__ void direct16_body(const GemmArgs &g, int split, int64_t tile0, int64_t ntiles, int nbx, float *part) {
    constexpr bool A_KC = (LAYOUT != HIDVAE_GEMM_TN);
    constexpr bool B_KC = (LAYOUT == HIDVAE_GEMM_NT);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the workgroup's waves form groups of `split`; group t works on tile tile0 + t, wave w of the group on its K share
    const int tw = wv / split, w = wv - tw * split;
    const int i16 = lane & 15, q = lane >> 4;
    const int64_t tile = tile0 + tw;
    if (tile < ntiles) {
        const int64_t m0 = (tile / nbx) * 16, n0 = (tile % nbx) * 16;
        const int64_t ra = (m0 + i16 < g.M) ? m0 + i16 : g.M - 1;
        const int64_t rb = (n0 + i16 < g.N) ? n0 + i16 : g.N - 1;
        const int va = 4 * (A_KC ? (int)(ra * g.lda) + 4 * q : (int)(4 * q * g.lda + ra));
        const int vb = 4 * (B_KC ? (int)(rb * g.ldb) + 4 * q : (int)(4 * q * g.ldb + rb));
        const int lda4 = (int)g.lda * 4, ldb4 = (int)g.ldb * 4, Ki = (int)g.K;
        const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(g.A), 0, (int)(4 * (A_KC ? (g.M - 1) * g.lda + g.K : (g.K - 1) * g.lda + g.M)), 0x00020000);
        const __amdgpu_buffer_rsrc_t rb_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(g.B), 0, (int)(4 * (B_KC ? (g.N - 1) * g.ldb + g.K : (g.K - 1) * g.ldb + g.N)), 0x00020000);
        const int nfull = Ki / 16;
        const int b_lo = nfull * w / split, b_hi = nfull * (w + 1) / split;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float ra_[NS][4], rb_[NS][4];
        auto load = [&](float (&av)[4], float (&bv)[4], int blk) {
            const bool in = blk < b_hi;
            load_block16<A_KC>(ra_rsrc, lda4, in ? va : HV_OOB, blk * 16, av);
            load_block16<B_KC>(rb_rsrc, ldb4, in ? vb : HV_OOB, blk * 16, bv);
        };
#pragma unroll
        for (int st = 0; st < NS - 1; st++) load(ra_[st], rb_[st], b_lo + st);
        for (int blk = b_lo; blk < b_hi; blk += NS) {
#pragma unroll
            for (int st = 0; st < NS; st++) {
                load(ra_[(st + NS - 1) % NS], rb_[(st + NS - 1) % NS], blk + st + NS - 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ra_[st][s4], rb_[st][s4], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (w == split - 1 && nfull * 16 < Ki) {
            load_tail16<A_KC>(ra_rsrc, lda4, va, nfull * 16, Ki, q, ra_[0]);
            load_tail16<B_KC>(rb_rsrc, ldb4, vb, nfull * 16, Ki, q, rb_[0]);
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ra_[0][s4], rb_[0][s4], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) part[wv * 256 + (4 * q + r) * 16 + i16] = acc[r];
    }
    __syncthreads();
    const float *mk = g.mask;
    const int groups = (int)(blockDim.x >> 6) / split;
    for (int e = threadIdx.x; e < groups * 256; e += blockDim.x) {
        const int t = e >> 8, ee = e & 255;
        const int64_t tl = tile0 + t;
        if (tl >= ntiles) break;
        float v = part[(t * split) * 256 + ee];
        for (int sidx = 1; sidx < split; sidx++) v += part[(t * split + sidx) * 256 + ee];  // fixed order: bit-reproducible
        const int64_t row = (tl / nbx) * 16 + (ee >> 4), col = (tl % nbx) * 16 + (ee & 15);
        if (row >= g.M || col >= g.N) continue;
        v += g.bias != nullptr ? g.bias[col] : 0.0f;
        if (g.aux != nullptr && g.epilogue < HIDVAE_EPI_DSILU && g.epilogue != HIDVAE_EPI_NONE) g.aux[row * g.ldaux + col] = v;
        v = apply_epilogue(g.epilogue, v, g.aux, row * g.ldaux + col, g.mask_scale);
        if (has_dropout(g)) v = v * dropout_factor(g, row, col);
        float *dst = g.C + row * g.ldc + col;
        *dst = g.accumulate ? *dst + v : v;
    }
}

// workgroup slot -> slot of the XCD's contiguous band (see xcd_tile); tiles are listed row-major, so bands are row bands
__device__ __forceinline__ int xcd_slot(int slot, int nslots) {
    return (nslots & 7) == 0 ? (slot & 7) * (nslots >> 3) + (slot >> 3) : slot;
}

struct PairArgs {
    GemmArgs g0, g1;       // g0: TN (dW = g^T x), g1: NN (dX = g W)
    int split0, split1;    // waves per tile
    int nbx0, nbx1;        // column tiles
    int64_t nt0, nt1;      // tiles
    int nb0;               // workgroups of problem 0 (each takes waves/split tiles)
    int nb1;               // workgroups of problem 1; the rest of the grid sums columns of g (bias gradient), 32 per workgroup
    const float *cs_x;     // g [rows, cols]
    int64_t cs_ld, cs_rows, cs_cols;
    float *cs_out;         // db [cols] (nullptr: no bias)
    int cs_accumulate;
    int lds_a;             // pair32: stage the NN product's k-contiguous A operand through LDS (coalesced loads)
};

// bias gradient db[c] = sum_b g[b, c]: 32 columns per workgroup, blockDim/32 row groups, partials added in ascending group order
__device__ __forceinline__ void colsum32_body(const PairArgs &p, int64_t c0, float *part) {
    const int col = threadIdx.x & 31, rg = threadIdx.x >> 5, nrg = (int)(blockDim.x >> 5);
    const int64_t c = c0 + col;
    float acc = 0.0f;
    if (c < p.cs_cols)
        for (int64_t r0 = rg; r0 < p.cs_rows; r0 += 8 * (int64_t)nrg) {  // eight loads in flight, same ascending order of additions
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = r0 + (int64_t)nrg * j < p.cs_rows ? p.cs_x[(r0 + (int64_t)nrg * j) * p.cs_ld + c] : 0.0f;
#pragma unroll
            for (int j = 0; j < 8; j++) acc += v[j];
        }
    part[rg * 32 + col] = acc;
    __syncthreads();
    if (threadIdx.x < 32 && c < p.cs_cols) {
        float v = part[col];
        for (int j = 1; j < nrg; j++) v += part[j * 32 + col];
        p.cs_out[c] = p.cs_accumulate ? p.cs_out[c] + v : v;
    }
}

__global__ __launch_bounds__(1024) void gemm_pair16_kernel(PairArgs p) {
    __shared__ float part[16 * 256];
    const int bid = blockIdx.x;
    const int waves = (int)(blockDim.x >> 6);
    if (bid < p.nb0) direct16_body<HIDVAE_GEMM_TN, 3>(p.g0, p.split0, (int64_t)xcd_slot(bid, p.nb0) * (waves / p.split0), p.nt0, p.nbx0, part);
    else if (bid < p.nb0 + p.nb1)
        direct16_body<HIDVAE_GEMM_NN, 3>(p.g1, p.split1, (int64_t)xcd_slot(bid - p.nb0, p.nb1) * (waves / p.split1), p.nt1, p.nbx1, part);
    else colsum32_body(p, (int64_t)(bid - p.nb0 - p.nb1) * 32, part);
}

// NWN > 1 (only with SPLIT == 1): the workgroup's NWN waves take NWN neighbouring column tiles of the SAME row tile, so the
// A rows they all read are fetched into the CU's L1 once instead of NWN times.
template <int LAYOUT, int SPLIT, int NS, int NWN = 1>
__global__ __launch_bounds__(64 * SPLIT * NWN) void gemm_direct_kernel(GemmArgs g) {
    constexpr bool A_KC = (LAYOUT != HIDVAE_GEMM_TN);
    constexpr bool B_KC = (LAYOUT == HIDVAE_GEMM_NT);
    static_assert(SPLIT == 1 || NWN == 1, "column-tile sharing is for the unsplit kernel");
    __shared__ float part[SPLIT > 1 ? SPLIT * 1024 : 1];
    const int lane = threadIdx.x & 63;
    // wave index as a SCALAR: the k offsets derived from it feed the buffer loads' soffset operand, which must be
    // provably uniform or the compiler wraps every load in a waterfall loop
    const int w = SPLIT == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i32 = lane & 31, h = lane >> 5;
    const int wn = NWN == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bx = blockIdx.x, by = blockIdx.y;
    if (NWN == 1) xcd_tile(gridDim.x, gridDim.y, g.M >= g.N, bx, by);
    const int64_t m0 = (int64_t)by * 32, n0 = ((int64_t)bx * NWN + wn) * 32;
    if (NWN > 1 && n0 >= g.N) return;  // (no barrier follows in the unsplit kernel)
    const int64_t ra = (m0 + i32 < g.M) ? m0 + i32 : g.M - 1;
    const int64_t rb = (n0 + i32 < g.N) ? n0 + i32 : g.N - 1;
    const int va = 4 * (A_KC ? (int)(ra * g.lda) + 4 * h : (int)(4 * h * g.lda + ra));  // ORDER-G16: lane half h starts at k0+4h
    const int vb = 4 * (B_KC ? (int)(rb * g.ldb) + 4 * h : (int)(4 * h * g.ldb + rb));
    const int lda4 = (int)g.lda * 4, ldb4 = (int)g.ldb * 4, Ki = (int)g.K;
    // logical extent of each operand in bytes (works for column-slice views too)
    const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(g.A), 0, (int)(4 * (A_KC ? (g.M - 1) * g.lda + g.K : (g.K - 1) * g.lda + g.M)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(g.B), 0, (int)(4 * (B_KC ? (g.N - 1) * g.ldb + g.K : (g.K - 1) * g.ldb + g.N)), 0x00020000);
    const int nfull = Ki / 16;  // complete 16-wide k-blocks, shared out over the SPLIT waves; the ragged rest goes last
    const int b_lo = nfull * w / SPLIT, b_hi = nfull * (w + 1) / SPLIT;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    // NS-deep register ring of k-blocks: NS-1 blocks of loads stay in flight behind the MFMAs of the current one
    float ra_[NS][8], rb_[NS][8];
    auto load = [&](float (&av)[8], float (&bv)[8], int blk) {
        const bool in = blk < b_hi;
        load_block<A_KC>(ra_rsrc, lda4, in ? va : HV_OOB, blk * 16, av);
        load_block<B_KC>(rb_rsrc, ldb4, in ? vb : HV_OOB, blk * 16, bv);
    };
#pragma unroll
    for (int st = 0; st < NS - 1; st++) load(ra_[st], rb_[st], b_lo + st);
    for (int blk = b_lo; blk < b_hi; blk += NS) {
#pragma unroll
        for (int st = 0; st < NS; st++) {
            // keep THIS order in the instruction stream: issue block (blk+st+NS-1)'s loads, then the MFMAs of block blk+st.
            // Left to itself hipcc sinks all loads of an iteration behind its MFMAs and waits vmcnt(0) at the loop top,
            // which serialises load latency and compute (seen in the .s: 2.3x the chain bound).
            load(ra_[(st + NS - 1) % NS], rb_[(st + NS - 1) % NS], blk + st + NS - 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s8 = 0; s8 < 8; s8++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra_[st][s8], rb_[st][s8], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (w == SPLIT - 1 && nfull * 16 < Ki) {
        load_tail<A_KC>(ra_rsrc, lda4, va, nfull * 16, Ki, h, ra_[0]);
        load_tail<B_KC>(rb_rsrc, ldb4, vb, nfull * 16, Ki, h, rb_[0]);
#pragma unroll
        for (int s8 = 0; s8 < 8; s8++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra_[0][s8], rb_[0][s8], acc, 0, 0, 0);
    }
    const float *mk = g.mask;
    if (SPLIT == 1) {
        const int64_t col = n0 + i32;
        if (col >= g.N) return;
        const float bias = g.bias != nullptr ? g.bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int64_t row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= g.M) continue;
            float v = acc[r] + bias;
            if (g.aux != nullptr && g.epilogue < HIDVAE_EPI_DSILU && g.epilogue != HIDVAE_EPI_NONE) g.aux[row * g.ldaux + col] = v;
            v = apply_epilogue(g.epilogue, v, g.aux, row * g.ldaux + col, g.mask_scale);
            if (has_dropout(g)) v = v * dropout_factor(g, row, col);
            float *dst = g.C + row * g.ldc + col;
            *dst = g.accumulate ? *dst + v : v;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; r++) part[w * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i32] = acc[r];
        __syncthreads();
        for (int e = threadIdx.x; e < 1024; e += 64 * SPLIT) {
            float v = part[e];
#pragma unroll
            for (int sidx = 1; sidx < SPLIT; sidx++) v += part[sidx * 1024 + e];  // fixed order: bit-reproducible
            const int64_t row = m0 + (e >> 5), col = n0 + (e & 31);
            if (row >= g.M || col >= g.N) continue;
            v += g.bias != nullptr ? g.bias[col] : 0.0f;
            if (g.aux != nullptr && g.epilogue < HIDVAE_EPI_DSILU && g.epilogue != HIDVAE_EPI_NONE) g.aux[row * g.ldaux + col] = v;
            v = apply_epilogue(g.epilogue, v, g.aux, row * g.ldaux + col, g.mask_scale);
            if (has_dropout(g)) v = v * dropout_factor(g, row, col);
            float *dst = g.C + row * g.ldc + col;
            *dst = g.accumulate ? *dst + v : v;
        }
    }
}

// ---- the direct kernel with COALESCED operand loads (k-contiguous operands) ------------------------------------------------------
// What bounds gemm_direct_kernel at the step's sizes is how its operands arrive, not the MFMA pipe: a lane reads 16
// bytes of ITS OWN row, so one wave instruction touches 32 different cache lines and uses 32 bytes of each; a stand-alone model of
// exactly that stream (scratch/mb/loadpat.hip) pulls 4-6.7 TB/s out of L2 chip-wide whatever the ring depth, the same strips read
// as 8 rows x 128 contiguous bytes per instruction 10-20 TB/s.  The MFMA fragment layout puts neighbouring ROWS on neighbouring
// lanes, so a coalesced read needs a transposition on the way: here through a PRIVATE LDS strip per wave -- no other wave ever touches
// it, the LDS unit executes one wave's instructions in order, so there is no barrier and no wait beyond the data's own.
//   stage = 32 k (two 16-blocks): lane -> row (lane >> 3) + 8 t, bytes [16 (lane & 7), +16) of the row's 128, t = 0..3 : four 16-byte
//   loads per operand, written as ds_write_b128 at [row][k] (144-byte rows: 16-byte aligned, bank groups rotate by one per row);
//   the fragment of a 16-block is then the same two 16-byte pieces per lane the register path loads from memory (x = k0 + 4h ..,
//   y = k0 + 8 + 4h ..) as two ds_read_b128 -- identical registers, identical ORDER-G16 chain, bit-identical results.
// Operands whose ROWS are contiguous in memory (A of TN, B of NN / TN) already read 128 contiguous bytes per half-wave and keep the
// register path.
constexpr int LSTR = 36;  // floats per staged row: 32 k + 4 (144 bytes)

// one wave's accumulation of tile (m0, n0) over 16-blocks [b_lo, b_hi) (+ the ragged tail block when do_tail): sA / sB = the wave's
// private strips (32 * LSTR floats each; sB unused for a row-contiguous B)
template <int LAYOUT>
__device__ __forceinline__ void directL_accumulate(const GemmArgs &g, int64_t m0, int64_t n0, int b_lo, int b_hi, bool do_tail, float *sA,
                                                   float *sB, f32x16 &acc) {
    constexpr bool B_KC = (LAYOUT == HIDVAE_GEMM_NT);
    static_assert(LAYOUT != HIDVAE_GEMM_TN, "the LDS-transposed loader serves k-contiguous A operands (NT, NN)");
    const int lane = threadIdx.x & 63;
    const int i32 = lane & 31, h = lane >> 5;
    const int lda4 = (int)g.lda * 4, ldb4 = (int)g.ldb * 4, Ki = (int)g.K;
    const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(g.A), 0, (int)(4 * ((g.M - 1) * g.lda + g.K)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(g.B), 0, (int)(4 * (B_KC ? (g.N - 1) * g.ldb + g.K : (g.K - 1) * g.ldb + g.N)), 0x00020000);
    // coalesced slots: t-th load of this lane = row (lane >> 3) + 8 t of the tile, bytes 16 (lane & 7) of the 32-k stage
    const int lr = lane >> 3, lk = lane & 7;
    int offA[4], offB[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int64_t ra = (m0 + lr + 8 * t < g.M) ? m0 + lr + 8 * t : g.M - 1;
        const int64_t rb = (n0 + lr + 8 * t < g.N) ? n0 + lr + 8 * t : g.N - 1;
        offA[t] = (int)(ra * g.lda) * 4 + 16 * lk;
        offB[t] = B_KC ? (int)(rb * g.ldb) * 4 + 16 * lk : 0;
    }
    // register-path offsets (the ragged tail block, and a row-contiguous B)
    const int64_t rra = (m0 + i32 < g.M) ? m0 + i32 : g.M - 1;
    const int64_t rrb = (n0 + i32 < g.N) ? n0 + i32 : g.N - 1;
    const int va = 4 * ((int)(rra * g.lda) + 4 * h);
    const int vb = 4 * (B_KC ? (int)(rrb * g.ldb) + 4 * h : (int)(4 * h * g.ldb + rrb));
    const int nfull = Ki / 16;
    f32x4 pa[4], pb[4];       // the next stage's coalesced loads
    float rbv[2][8];          // a row-contiguous B keeps the register path: the stage's two blocks
    auto fetch = [&](int blk) {  // stage starting at 16-block `blk`: blocks blk and blk+1 where they lie inside [b_lo, b_hi)
        const int nb = b_hi - blk;  // <= 0: nothing, 1: half a stage
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const bool in = nb >= 2 || (nb == 1 && lk < 4);
            pa[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, in ? offA[t] : HV_OOB, blk * 64, 0));
            if (B_KC) pb[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb_rsrc, in ? offB[t] : HV_OOB, blk * 64, 0));
        }
        if (!B_KC) {
            load_block<false>(rb_rsrc, ldb4, nb >= 1 ? vb : HV_OOB, blk * 16, rbv[0]);
            load_block<false>(rb_rsrc, ldb4, nb >= 2 ? vb : HV_OOB, blk * 16 + 16, rbv[1]);
        }
    };
    fetch(b_lo);
    for (int blk = b_lo; blk < b_hi; blk += 2) {
        // this wave's staged strip: written and read by this wave only, in program order (no barrier)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            *reinterpret_cast<f32x4 *>(sA + (lr + 8 * t) * LSTR + 4 * lk) = pa[t];
            if (B_KC) *reinterpret_cast<f32x4 *>(sB + (lr + 8 * t) * LSTR + 4 * lk) = pb[t];
        }
        float bcur[2][8];
        if (!B_KC) {
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int j = 0; j < 8; j++) bcur[u][j] = rbv[u][j];
        }
        __builtin_amdgcn_wave_barrier();
        float fa[2][8], fb[2][8];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(sA + i32 * LSTR + 16 * u + 4 * h);
            const f32x4 y = *reinterpret_cast<const f32x4 *>(sA + i32 * LSTR + 16 * u + 8 + 4 * h);
            fa[u][0] = x[0]; fa[u][1] = y[0]; fa[u][2] = x[1]; fa[u][3] = y[1]; fa[u][4] = x[2]; fa[u][5] = y[2]; fa[u][6] = x[3]; fa[u][7] = y[3];
            if (B_KC) {
                const f32x4 p = *reinterpret_cast<const f32x4 *>(sB + i32 * LSTR + 16 * u + 4 * h);
                const f32x4 q = *reinterpret_cast<const f32x4 *>(sB + i32 * LSTR + 16 * u + 8 + 4 * h);
                fb[u][0] = p[0]; fb[u][1] = q[0]; fb[u][2] = p[1]; fb[u][3] = q[1]; fb[u][4] = p[2]; fb[u][5] = q[2]; fb[u][6] = p[3]; fb[u][7] = q[3];
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) fb[u][j] = bcur[u][j];
            }
        }
        __builtin_amdgcn_wave_barrier();
        fetch(blk + 2);  // the next stage's loads fly behind this stage's 16 MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s8 = 0; s8 < 8; s8++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][s8], fb[0][s8], acc, 0, 0, 0);
        if (blk + 1 < b_hi) {
#pragma unroll
            for (int s8 = 0; s8 < 8; s8++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][s8], fb[1][s8], acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (do_tail && nfull * 16 < Ki) {  // the ragged last block: per-element register path, as gemm_direct_kernel
        float ta[8], tb[8];
        load_tail<true>(ra_rsrc, lda4, va, nfull * 16, Ki, h, ta);
        load_tail<B_KC>(rb_rsrc, ldb4, vb, nfull * 16, Ki, h, tb);
#pragma unroll
        for (int s8 = 0; s8 < 8; s8++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[s8], tb[s8], acc, 0, 0, 0);
    }
}

// the 32x32-tile twin of direct16_body / gemm_pair16_kernel: same arithmetic as gemm_direct_kernel with a run-time split
// stageL (NN only, optional): 32 * LSTR floats per wave -- the k-contiguous A operand then arrives through directL_accumulate
template <int LAYOUT, int NS>
__device__ __forceinline__ void direct32_body(const GemmArgs &g, int split, int64_t tile0, int64_t ntiles, int nbx, float *part,
                                              float *stageL = nullptr) {
    constexpr bool A_KC = (LAYOUT != HIDVAE_GEMM_TN);
    constexpr bool B_KC = (LAYOUT == HIDVAE_GEMM_NT);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tw = wv / split, w = wv - tw * split;
    const int i32 = lane & 31, h = lane >> 5;
    const int64_t tile = tile0 + tw;
    if (tile < ntiles) {
        const int64_t m0 = (tile / nbx) * 32, n0 = (tile % nbx) * 32;
        const int64_t ra = (m0 + i32 < g.M) ? m0 + i32 : g.M - 1;
        const int64_t rb = (n0 + i32 < g.N) ? n0 + i32 : g.N - 1;
        const int va = 4 * (A_KC ? (int)(ra * g.lda) + 4 * h : (int)(4 * h * g.lda + ra));
        const int vb = 4 * (B_KC ? (int)(rb * g.ldb) + 4 * h : (int)(4 * h * g.ldb + rb));
        const int lda4 = (int)g.lda * 4, ldb4 = (int)g.ldb * 4, Ki = (int)g.K;
        const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(g.A), 0, (int)(4 * (A_KC ? (g.M - 1) * g.lda + g.K : (g.K - 1) * g.lda + g.M)), 0x00020000);
        const __amdgpu_buffer_rsrc_t rb_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(g.B), 0, (int)(4 * (B_KC ? (g.N - 1) * g.ldb + g.K : (g.K - 1) * g.ldb + g.N)), 0x00020000);
        const int nfull = Ki / 16;
        const int b_lo = nfull * w / split, b_hi = nfull * (w + 1) / split;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.0f;
        if (LAYOUT == HIDVAE_GEMM_NN && stageL != nullptr) {  // coalesced A through this wave's private LDS strip: same chain, same bits
            if constexpr (LAYOUT == HIDVAE_GEMM_NN)
                directL_accumulate<HIDVAE_GEMM_NN>(g, m0, n0, b_lo, b_hi, w == split - 1, stageL + wv * (32 * LSTR), nullptr, acc);
        } else {
        float ra_[NS][8], rb_[NS][8];
        auto load = [&](float (&av)[8], float (&bv)[8], int blk) {
            const bool in = blk < b_hi;
            load_block<A_KC>(ra_rsrc, lda4, in ? va : HV_OOB, blk * 16, av);
            load_block<B_KC>(rb_rsrc, ldb4, in ? vb : HV_OOB, blk * 16, bv);
        };
#pragma unroll
        for (int st = 0; st < NS - 1; st++) load(ra_[st], rb_[st], b_lo + st);
        for (int blk = b_lo; blk < b_hi; blk += NS) {
#pragma unroll
            for (int st = 0; st < NS; st++) {
                load(ra_[(st + NS - 1) % NS], rb_[(st + NS - 1) % NS], blk + st + NS - 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s8 = 0; s8 < 8; s8++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra_[st][s8], rb_[st][s8], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (w == split - 1 && nfull * 16 < Ki) {
            load_tail<A_KC>(ra_rsrc, lda4, va, nfull * 16, Ki, h, ra_[0]);
            load_tail<B_KC>(rb_rsrc, ldb4, vb, nfull * 16, Ki, h, rb_[0]);
#pragma unroll
            for (int s8 = 0; s8 < 8; s8++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra_[0][s8], rb_[0][s8], acc, 0, 0, 0);
        }
        }
#pragma unroll
        for (int r = 0; r < 16; r++) part[wv * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i32] = acc[r];
    }
    __syncthreads();
    const float *mk = g.mask;
    const int groups = (int)(blockDim.x >> 6) / split;
    for (int e = threadIdx.x; e < groups * 1024; e += blockDim.x) {
        const int t = e >> 10, ee = e & 1023;
        const int64_t tl = tile0 + t;
        if (tl >= ntiles) break;
        float v = part[(t * split) * 1024 + ee];
        for (int sidx = 1; sidx < split; sidx++) v += part[(t * split + sidx) * 1024 + ee];  // fixed order: bit-reproducible
        const int64_t row = (tl / nbx) * 32 + (ee >> 5), col = (tl % nbx) * 32 + (ee & 31);
        if (row >= g.M || col >= g.N) continue;
        v += g.bias != nullptr ? g.bias[col] : 0.0f;
        if (g.aux != nullptr && g.epilogue < HIDVAE_EPI_DSILU && g.epilogue != HIDVAE_EPI_NONE) g.aux[row * g.ldaux + col] = v;
        v = apply_epilogue(g.epilogue, v, g.aux, row * g.ldaux + col, g.mask_scale);
        if (has_dropout(g)) v = v * dropout_factor(g, row, col);
        float *dst = g.C + row * g.ldc + col;
        *dst = g.accumulate ? *dst + v : v;
    }
}

// dynamic LDS: 4 KB of split-K partials per wave (+ a 32 x LSTR strip per wave with lds_a) -- sized by the launch, because a static
// 8-wave allocation (68 KB with the strips) held the usual 4-wave workgroups at two per CU where registers allow four
template <int NS1>
__global__ __launch_bounds__(512) void gemm_pair32_kernel(PairArgs p) {
    extern __shared__ __attribute__((aligned(16))) float pair_lds[];
    const int bid = blockIdx.x;
    const int waves = (int)(blockDim.x >> 6);
    float *part = pair_lds;
    float *stageL = pair_lds + waves * 1024;  // dX = g W: the k-contiguous g rows arrive coalesced (directL), only with lds_a
    if (bid < p.nb0) direct32_body<HIDVAE_GEMM_TN, 3>(p.g0, p.split0, (int64_t)xcd_slot(bid, p.nb0) * (waves / p.split0), p.nt0, p.nbx0, part);
    else if (bid < p.nb0 + p.nb1)
        direct32_body<HIDVAE_GEMM_NN, NS1>(p.g1, p.split1, (int64_t)xcd_slot(bid - p.nb0, p.nb1) * (waves / p.split1), p.nt1, p.nbx1, part,
                                          p.lds_a ? stageL : nullptr);
    else colsum32_body(p, (int64_t)(bid - p.nb0 - p.nb1) * 32, part);
}

// ---- the dW / dX pair of a Linear backward on LDS-SHARED 64x64 tiles, the work dealt out evenly (gemm_mid_sk_kernel) ------------
// What the per-wave kernels above cannot do is reuse an operand byte: every wave pulls its own 32 x 16 pieces through L1 (8 FLOP per
// loaded byte), the launches are bound by that flow (the dW and dX halves of a paired launch take exactly as long together as alone,
// MFMA pipe 33 % busy); and one LDS-tiled workgroup per 64x64 tile leaves SIMDs idle at these sizes (96 + 128 tiles for a 768 x 512
// layer at B = 1024 on 256 CUs) while a tile's 16 k-steps of 2,048 MFMA cycles each set the launch's length.  Here
//   * a workgroup is 4 KG waves: wave (kg, wm, wn) owns the 32x32 quarter (wm, wn) of a tile for the k-blocks 16 kg + 16 KG s.  Per
//     STEP (16 KG consecutive k of one tile) the whole workgroup stages a 64 x 16KG slab of each operand -- one coalesced 16-byte
//     load per thread and operand, 16 FLOP per loaded byte --, register-prefetched PF steps ahead, three LDS buffers, the next step's
//     fragments read while this step's MFMAs run, one barrier per step;
//   * the launch is a list of S = tiles x steps-per-tile steps (dW tiles, then dX tiles, then the bias column sums priced in steps),
//     cut into G equal contiguous ranges, one per workgroup (G = the workgroup slots of the chip or fewer): every SIMD gets the same
//     number of MFMAs whatever the tile count.  A range that ends inside a tile leaves a partial 64x64 tile in the workspace; the
//     workgroup that arrives LAST at a tile (one counter per tile) adds the tile's partials in ascending k order -- always the same
//     order, whoever is last: deterministic -- and applies the epilogue.  Nobody waits for anybody.
// Not the ORDER-G16 chain: gradients only (the forward layers keep one exact chain per output).
// k-major slabs (A of TN, B of NN / TN) live as [k][64 + 4]: lane (i32, h) of a quarter reads k = 8h + j at step j, the two lane
// halves 8 rows = 544 words = 32 banks apart; k-contiguous slabs (A of NN) as [64][16KG + 4]: two ds_read_b128 per block.
constexpr int MID_STR = 68;

struct SkArgs {
    PairArgs p;         // g0 (TN: dW), g1 (NN: dX), nb0 / nb1 = 64x64 tiles, nbx0 / nbx1 = column tiles, cs_* = the bias column sums
    int n0, n1;         // steps per tile
    int nbc, cu;        // column-sum units (32 columns each) and their price in steps
    int S, q, G;        // total steps, steps per workgroup, workgroups
    float *slabs;       // [G][2][4096] partial tiles
    int *counters;      // [nb0 + nb1]
};

template <int KG>
struct SkGeom {
    static constexpr int BKS = 16 * KG, T = 256 * KG, KC_STR = BKS + 4, NB = 3, NG = 1024 / T;
    static constexpr int A_KC_SZ = 64 * KC_STR, KM_SZ = BKS * MID_STR;
    static constexpr int BUF = (A_KC_SZ > KM_SZ ? A_KC_SZ : KM_SZ) + KM_SZ;
    static constexpr size_t lds_bytes = (size_t)NB * BUF * 4;
    static_assert(NB * BUF >= KG * 4096, "the stage buffers must hold the KG partial tiles");
};

// k-steps [ks0, ks0 + len) of tile (m0, n0) -> the wave's 32x32 accumulator of its k-blocks
template <int LAYOUT, int KG, int PF>
__device__ __forceinline__ void sk_accumulate(const GemmArgs &g, int64_t m0, int64_t n0, int ks0, int len, float *lds, f32x16 &acc) {
    static_assert(LAYOUT != HIDVAE_GEMM_NT && PF >= 2, "gradient products only: TN (dW) and NN (dX)");
    using G_ = SkGeom<KG>;
    constexpr bool A_KC = (LAYOUT == HIDVAE_GEMM_NN);
    constexpr int BKS = G_::BKS, KC_STR = G_::KC_STR, NB = G_::NB, BUF = G_::BUF;
    constexpr int A_SZ = A_KC ? G_::A_KC_SZ : G_::KM_SZ;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int i32 = lane & 31, h = lane >> 5;
    const int Ki = (int)g.K;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(g.A), 0, (int)(4 * (A_KC ? (g.M - 1) * g.lda + g.K : (g.K - 1) * g.lda + g.M)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(g.B), 0, (int)(4 * ((g.K - 1) * g.ldb + g.N)), 0x00020000);
    // this thread's 16-byte slot of a slab
    const int kr = tid >> 4, c4 = (tid & 15) * 4;                      // k-major: row kr of the step, columns c4 .. c4 + 3
    const int ar = tid / (BKS / 4), ak4 = (tid % (BKS / 4)) * 4;       // k-contiguous: tile row ar, k = ak4 .. ak4 + 3 of the step
    const int64_t arow = (m0 + ar < g.M) ? m0 + ar : g.M - 1;
    const int offA = A_KC ? 4 * ((int)(arow * g.lda) + ak4) : 4 * (kr * (int)g.lda + (int)m0 + c4);
    const int offB = 4 * (kr * (int)g.ldb + (int)n0 + c4);
    const int stepA = A_KC ? 4 * BKS : 4 * BKS * (int)g.lda, stepB = 4 * BKS * (int)g.ldb;
    const int ldsA = A_KC ? ar * KC_STR + ak4 : kr * MID_STR + c4, ldsB = kr * MID_STR + c4;
    f32x4 pa[PF], pb[PF];
    auto fetch = [&](int slot, int s) {  // (steps past the segment and k rows past K aim out of range: zeros, no traffic)
        const int ks = ks0 + s;
        const bool ka = s < len && (A_KC ? ks * BKS + ak4 < Ki : ks * BKS + kr < Ki);
        const bool kb = s < len && ks * BKS + kr < Ki;
        pa[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, ka ? offA : HV_OOB, ks * stepA, 0));
        pb[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, kb ? offB : HV_OOB, ks * stepB, 0));
    };
    auto stash = [&](int slot, int s, int buf) {
        float *dst = lds + buf * BUF;
        f32x4 a = pa[slot];
        if (A_KC) {  // a 16-byte piece that straddles the end of the row: what follows K belongs to the next row
            const int k = (ks0 + s) * BKS + ak4;
            if (k + 4 > Ki) {
#pragma unroll
                for (int e = 0; e < 4; e++) a[e] = k + e < Ki ? a[e] : 0.0f;
            }
        }
        *reinterpret_cast<f32x4 *>(dst + ldsA) = a;
        *reinterpret_cast<f32x4 *>(dst + A_SZ + ldsB) = pb[slot];
    };
    float fa[2][8], fb[2][8];
    auto frags = [&](int fs, int buf) {
        const float *As = lds + buf * BUF, *Bs = As + A_SZ;
        if (A_KC) {
            const float *ap = As + (32 * wm + i32) * KC_STR + 16 * kg + 8 * h;
            const f32x4 x = *reinterpret_cast<const f32x4 *>(ap), y = *reinterpret_cast<const f32x4 *>(ap + 4);
#pragma unroll
            for (int j = 0; j < 4; j++) { fa[fs][j] = x[j]; fa[fs][4 + j] = y[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) fa[fs][j] = As[(16 * kg + 8 * h + j) * MID_STR + 32 * wm + i32];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) fb[fs][j] = Bs[(16 * kg + 8 * h + j) * MID_STR + 32 * wn + i32];
    };
    // prologue: steps 0 and 1 staged, fragments of step 0 in registers, loads of steps 2 .. PF + 1 in flight
#pragma unroll
    for (int p = 0; p < PF; p++) fetch(p, p);
    stash(0, 0, 0);
    fetch(0, PF);
    stash(1 % PF, 1, 1);
    fetch(1 % PF, PF + 1);
    __syncthreads();
    frags(0, 0);
    constexpr int U2 = PF % 2 == 0 ? PF : 2 * PF, UN = U2 % 3 == 0 ? U2 : 3 * U2;  // lcm(2, NB, PF)
    constexpr int FRAG_AT = 2;
    for (int s0 = 0; s0 < len; s0 += UN) {
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int s = s0 + u;
            if (s < len) {  // (uniform)
                // this step's MFMAs lead (their fragments arrived during the previous step); the next step's fragment reads are
                // issued behind the first of them, so the matrix pipe starts right after the barrier and the reads fly under it
#pragma unroll
                for (int j = 0; j < FRAG_AT; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u & 1][j], fb[u & 1][j], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < len) frags((u + 1) & 1, (u + 1) % NB);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = FRAG_AT; j < 8; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u & 1][j], fb[u & 1][j], acc, 0, 0, 0);
                if (s + 2 < len) {
                    stash((u + 2) % PF, s + 2, (u + 2) % NB);
                    fetch((u + 2) % PF, s + 2 + PF);
                }
                __syncthreads();
            }
        }
    }
}

// epilogue + store of the workgroup's tile: thread t holds the float4 groups e4 = t + T i (row e4 >> 4, columns 4 (e4 & 15) ..)
template <int KG>
__device__ __forceinline__ void sk_output(const GemmArgs &g, int64_t m0, int64_t n0, const f32x4 (&v)[SkGeom<KG>::NG]) {
    const bool vec = (g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0);
#pragma unroll
    for (int i = 0; i < SkGeom<KG>::NG; i++) {
        const int e4 = threadIdx.x + SkGeom<KG>::T * i;
        const int64_t row = m0 + (e4 >> 4), col = n0 + 4 * (e4 & 15);
        if (row >= g.M || col >= g.N) continue;
        float *dst = g.C + row * g.ldc + col;
        f32x4 o = v[i];
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (col + e < g.N) {
                o[e] = apply_epilogue(g.epilogue, o[e], g.aux, row * g.ldaux + col + e, g.mask_scale);
                if (g.accumulate) o[e] = dst[e] + o[e];
            }
        if (vec && col + 4 <= g.N) *reinterpret_cast<f32x4 *>(dst) = o;
        else {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (col + e < g.N) dst[e] = o[e];
        }
    }
}

// Partial tiles cross workgroups -- possibly XCDs, whose L2s do not see each other's dirty lines -- through memory: system-scope
// (sc0 sc1) stores and loads that bypass the caches, ordered by the store acknowledgements (s_waitcnt vmcnt(0)) and a relaxed
// device-scope counter.  A release / acquire FENCE pair would do the same, but on gfx950 that is a write-back of the whole L2 and an
// invalidate of it per wave: measured 10x slower (220 us for the 768 x 512 layer), every operand re-fetched from HBM.
__device__ __forceinline__ void sk_store_through(float *p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 sk_load_through(const float *p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// one segment: steps [ks0, ks0 + len) of tile `tl` (of nT steps, first global step tstart) of problem g; v = this workgroup's index
template <int LAYOUT, int KG, int PF>
__device__ __forceinline__ void sk_segment(const SkArgs &a, const GemmArgs &g, int tl, int nbx, int counter, int ks0, int len, int nT,
                                           int tstart, int v, float *lds, int *s_old) {
    using G_ = SkGeom<KG>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int i32 = lane & 31, h = lane >> 5;
    const int by = tl / nbx, bx = tl - by * nbx;
    const int64_t m0 = (int64_t)by * 64, n0 = (int64_t)bx * 64;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    sk_accumulate<LAYOUT, KG, PF>(g, m0, n0, ks0, len, lds, acc);
    // the KG partial tiles -> LDS (in the stage buffers: every wave is past its last fragment read), summed in ascending kg order
#pragma unroll
    for (int r = 0; r < 16; r++) lds[kg * 4096 + (32 * wm + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + 32 * wn + i32] = acc[r];
    __syncthreads();
    f32x4 vv[G_::NG];
#pragma unroll
    for (int i = 0; i < G_::NG; i++) {
        const int e4 = tid + G_::T * i;
        vv[i] = *reinterpret_cast<const f32x4 *>(lds + 4 * e4);
#pragma unroll
        for (int q = 1; q < KG; q++) vv[i] = vv[i] + *reinterpret_cast<const f32x4 *>(lds + q * 4096 + 4 * e4);
    }
    if (ks0 == 0 && len == nT) {  // the whole tile
        sk_output<KG>(g, m0, n0, vv);
    } else {
        // a piece of the tile: park it (slot 2v: the tile began before this range; 2v + 1: it begins here and goes on), count arrivals
        float *mine = a.slabs + ((int64_t)2 * v + (ks0 == 0 ? 1 : 0)) * 4096;
#pragma unroll
        for (int i = 0; i < G_::NG; i++) sk_store_through(mine + 4 * (tid + G_::T * i), vv[i]);
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");  // this thread's piece has been acknowledged by memory
        __syncthreads();                                     // ... and so has every thread's
        const int v_first = tstart / a.q, v_last = (tstart + nT - 1) / a.q;
        if (tid == 0) *s_old = __hip_atomic_fetch_add(a.counters + counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*s_old == v_last - v_first) {  // last to arrive: every piece is in memory
            for (int u = v_first; u <= v_last; u++) {  // ascending k, whoever finishes
                const float *src = a.slabs + ((int64_t)2 * u + (u * a.q <= tstart ? 1 : 0)) * 4096;
                f32x4 w[G_::NG];
#pragma unroll
                for (int i = 0; i < G_::NG; i++) w[i] = sk_load_through(src + 4 * (tid + G_::T * i));
#pragma unroll
                for (int i = 0; i < G_::NG; i++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[i]) : : "memory");  // (ties the uses of w to the wait)
#pragma unroll
                for (int i = 0; i < G_::NG; i++) vv[i] = u == v_first ? w[i] : vv[i] + w[i];
            }
            sk_output<KG>(g, m0, n0, vv);
            if (tid == 0) __hip_atomic_store(a.counters + counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // clean for the next launch
        }
    }
    __syncthreads();  // the stage buffers (and s_old) are free again
}

template <int KG, int PF>
__global__ __launch_bounds__(256 * KG) void gemm_mid_sk_kernel(SkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float mid_lds[];
    __shared__ int s_old;
    const int v = xcd_slot((int)blockIdx.x, a.G);  // XCD j works on the j-th contiguous eighth of the step list
    if (a.q == 0) {  // one whole tile (or one strip of column sums) per workgroup: nothing to hand over
        if (v < a.p.nb0) sk_segment<HIDVAE_GEMM_TN, KG, PF>(a, a.p.g0, v, a.p.nbx0, 0, 0, a.n0, a.n0, 0, v, mid_lds, &s_old);
        else if (v < a.p.nb0 + a.p.nb1) sk_segment<HIDVAE_GEMM_NN, KG, PF>(a, a.p.g1, v - a.p.nb0, a.p.nbx1, 0, 0, a.n1, a.n1, 0, v, mid_lds, &s_old);
        else colsum32_body(a.p, (int64_t)(v - a.p.nb0 - a.p.nb1) * 32, mid_lds);
        return;
    }
    int step = v * a.q;
    const int Sg = a.p.nb0 * a.n0 + a.p.nb1 * a.n1;
    const int hi = step + a.q < a.S ? step + a.q : a.S;
    const int S0 = a.p.nb0 * a.n0;
    while (step < hi) {
        if (step >= Sg) {  // bias column sums: a unit belongs to the range that holds its first step
            const int rel = step - Sg, u = (rel + a.cu - 1) / a.cu;
            if (u >= a.nbc || Sg + u * a.cu >= hi) break;
            colsum32_body(a.p, (int64_t)u * 32, mid_lds);
            __syncthreads();
            step = Sg + u * a.cu + 1;
            continue;
        }
        const bool second = step >= S0;
        const int nT = second ? a.n1 : a.n0;
        const int rel = second ? step - S0 : step;
        const int tl = rel / nT, ks0 = rel - tl * nT;
        int len = nT - ks0;
        if (len > hi - step) len = hi - step;
        if (!second) sk_segment<HIDVAE_GEMM_TN, KG, PF>(a, a.p.g0, tl, a.p.nbx0, tl, ks0, len, nT, step - ks0, v, mid_lds, &s_old);
        else sk_segment<HIDVAE_GEMM_NN, KG, PF>(a, a.p.g1, tl, a.p.nbx1, a.p.nb0 + tl, ks0, len, nT, step - ks0, v, mid_lds, &s_old);
        step += len;
    }
}

template <int KG, int PF>
int launch_mid_sk(const SkArgs &a, hipStream_t s) {
    constexpr size_t bytes = SkGeom<KG>::lds_bytes;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_mid_sk_kernel<KG, PF>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (attr != hipSuccess) return -1;
    hipLaunchKernelGGL((gemm_mid_sk_kernel<KG, PF>), dim3((unsigned)a.G), dim3(256 * KG), bytes, s, a);
    return 0;
}

template <int LAYOUT, int SPLIT>
__global__ __launch_bounds__(64 * SPLIT) void gemm_directL_kernel(GemmArgs g) {
    // per wave: the A strip, the B strip (NT only), and -- after the K loop, in the same words -- the wave's 32x32 partial tile
    constexpr int WSTRIDE = (LAYOUT == HIDVAE_GEMM_NT ? 2 : 1) * 32 * LSTR;
    static_assert(WSTRIDE >= 1024, "a wave's strips must hold its partial tile");
    __shared__ __attribute__((aligned(16))) float stage[SPLIT * WSTRIDE];
    const int lane = threadIdx.x & 63;
    const int w = SPLIT == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i32 = lane & 31, h = lane >> 5;
    int bx = blockIdx.x, by = blockIdx.y;
    xcd_tile(gridDim.x, gridDim.y, g.M >= g.N, bx, by);
    const int64_t m0 = (int64_t)by * 32, n0 = (int64_t)bx * 32;
    const int nfull = (int)g.K / 16;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    directL_accumulate<LAYOUT>(g, m0, n0, nfull * w / SPLIT, nfull * (w + 1) / SPLIT, w == SPLIT - 1, stage + w * WSTRIDE,
                               stage + w * WSTRIDE + 32 * LSTR, acc);
    const float *mk = g.mask;
    if (SPLIT == 1) {
        const int64_t col = n0 + i32;
        if (col >= g.N) return;
        const float bias = g.bias != nullptr ? g.bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int64_t row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= g.M) continue;
            float v = acc[r] + bias;
            if (g.aux != nullptr && g.epilogue < HIDVAE_EPI_DSILU && g.epilogue != HIDVAE_EPI_NONE) g.aux[row * g.ldaux + col] = v;
            v = apply_epilogue(g.epilogue, v, g.aux, row * g.ldaux + col, g.mask_scale);
            if (has_dropout(g)) v = v * dropout_factor(g, row, col);
            float *dst = g.C + row * g.ldc + col;
            *dst = g.accumulate ? *dst + v : v;
        }
    } else {
        __builtin_amdgcn_wave_barrier();  // (this wave's last fragment reads precede its partial tile in program order)
#pragma unroll
        for (int r = 0; r < 16; r++) stage[w * WSTRIDE + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i32] = acc[r];
        __syncthreads();
        for (int e = threadIdx.x; e < 1024; e += 64 * SPLIT) {
            float v = stage[e];
#pragma unroll
            for (int sidx = 1; sidx < SPLIT; sidx++) v += stage[sidx * WSTRIDE + e];  // fixed order: bit-reproducible
            const int64_t row = m0 + (e >> 5), col = n0 + (e & 31);
            if (row >= g.M || col >= g.N) continue;
            v += g.bias != nullptr ? g.bias[col] : 0.0f;
            if (g.aux != nullptr && g.epilogue < HIDVAE_EPI_DSILU && g.epilogue != HIDVAE_EPI_NONE) g.aux[row * g.ldaux + col] = v;
            v = apply_epilogue(g.epilogue, v, g.aux, row * g.ldaux + col, g.mask_scale);
            if (has_dropout(g)) v = v * dropout_factor(g, row, col);
            float *dst = g.C + row * g.ldc + col;
            *dst = g.accumulate ? *dst + v : v;
        }
    }
}

// ---- the same idea on 16x16 tiles (v_mfma_f32_16x16x4_f32), NT, one exact chain per output: four times the waves of the 32x32 form.
// Per 32-k stage a wave reads its 16 A rows and 16 B rows as 8 rows x 128 contiguous bytes per instruction (2 + 2 loads), parks them
// in its private strips [16][LSTR] and takes each 16-block's fragment as ONE ds_read_b128 per operand (lane (i, q): k0 + 4q .. + 3,
// what load_block16 reads from memory).  WAVES waves per workgroup, each on its own tile (no barrier anywhere).
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void gemm_directL16_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float stage[WAVES * 2 * 16 * LSTR];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    float *sA = stage + wv * (2 * 16 * LSTR), *sB = sA + 16 * LSTR;
    const int nbx = (int)((g.N + 15) / 16), nby = (int)((g.M + 15) / 16);
    const int64_t ntiles = (int64_t)nbx * nby;
    const int64_t nslots = (ntiles + WAVES - 1) / WAVES;
    const int64_t tile = (int64_t)xcd_slot((int)blockIdx.x, (int)nslots) * WAVES + wv;  // neighbouring column tiles share a workgroup
    if (tile >= ntiles) return;
    const int64_t m0 = (tile / nbx) * 16, n0 = (tile % nbx) * 16;
    const int lda4 = (int)g.lda * 4, ldb4 = (int)g.ldb * 4, Ki = (int)g.K;
    const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(g.A), 0, (int)(4 * ((g.M - 1) * g.lda + g.K)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(g.B), 0, (int)(4 * ((g.N - 1) * g.ldb + g.K)), 0x00020000);
    const int lr = lane >> 3, lk = lane & 7;
    int offA[2], offB[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int64_t ra = (m0 + lr + 8 * t < g.M) ? m0 + lr + 8 * t : g.M - 1;
        const int64_t rb = (n0 + lr + 8 * t < g.N) ? n0 + lr + 8 * t : g.N - 1;
        offA[t] = (int)(ra * g.lda) * 4 + 16 * lk;
        offB[t] = (int)(rb * g.ldb) * 4 + 16 * lk;
    }
    const int64_t rra = (m0 + i16 < g.M) ? m0 + i16 : g.M - 1;
    const int64_t rrb = (n0 + i16 < g.N) ? n0 + i16 : g.N - 1;
    const int va = 4 * ((int)(rra * g.lda) + 4 * q), vb = 4 * ((int)(rrb * g.ldb) + 4 * q);
    const int nfull = Ki / 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 pa[2][2], pb[2][2];  // two stages of coalesced loads in flight
    auto fetch = [&](int st, int blk) {
        const int nb = nfull - blk;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const bool in = nb >= 2 || (nb == 1 && lk < 4);
            pa[st][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, in ? offA[t] : HV_OOB, blk * 64, 0));
            pb[st][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb_rsrc, in ? offB[t] : HV_OOB, blk * 64, 0));
        }
    };
    auto consume = [&](int st, int blk) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            *reinterpret_cast<f32x4 *>(sA + (lr + 8 * t) * LSTR + 4 * lk) = pa[st][t];
            *reinterpret_cast<f32x4 *>(sB + (lr + 8 * t) * LSTR + 4 * lk) = pb[st][t];
        }
        __builtin_amdgcn_wave_barrier();
        f32x4 fa[2], fb[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            fa[u] = *reinterpret_cast<const f32x4 *>(sA + i16 * LSTR + 16 * u + 4 * q);
            fb[u] = *reinterpret_cast<const f32x4 *>(sB + i16 * LSTR + 16 * u + 4 * q);
        }
        __builtin_amdgcn_wave_barrier();
        fetch(st, blk + 4);  // this stage's registers are free again: request the stage after next
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0][s4], fb[0][s4], acc, 0, 0, 0);
        if (blk + 1 < nfull) {
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1][s4], fb[1][s4], acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    fetch(0, 0);
    fetch(1, 2);
    for (int blk = 0; blk < nfull; blk += 4) {
        consume(0, blk);
        if (blk + 2 < nfull) consume(1, blk + 2);
    }
    if (nfull * 16 < Ki) {  // the ragged last block: per-element register path
        float ta[4], tb[4];
        load_tail16<true>(ra_rsrc, lda4, va, nfull * 16, Ki, q, ta);
        load_tail16<true>(rb_rsrc, ldb4, vb, nfull * 16, Ki, q, tb);
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[s4], tb[s4], acc, 0, 0, 0);
    }
    // C/D map of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
    const int64_t col = n0 + i16;
    if (col >= g.N) return;
    const float bias = g.bias != nullptr ? g.bias[col] : 0.0f;
    const float *mk = g.mask;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int64_t row = m0 + 4 * q + r;
        if (row >= g.M) continue;
        float v = acc[r] + bias;
        if (g.aux != nullptr && g.epilogue < HIDVAE_EPI_DSILU && g.epilogue != HIDVAE_EPI_NONE) g.aux[row * g.ldaux + col] = v;
        v = apply_epilogue(g.epilogue, v, g.aux, row * g.ldaux + col, g.mask_scale);
        if (has_dropout(g)) v = v * dropout_factor(g, row, col);
        float *dst = g.C + row * g.ldc + col;
        *dst = g.accumulate ? *dst + v : v;
    }
}

// ---- the exact-chain forward layers on LDS-SHARED tiles (gemm_tile16_kernel): NT, one ORDER-G16 chain per output -----------------
// gemm_directL16_kernel gives every wave its own 16x16 tile and its own copy of the operand rows: 4 FLOP per byte through L1, and
// the step's forward layers run at the rate that flow allows (60 TFLOP/s).  Here a workgroup of WM x WN waves owns a (16 WM) x (16 WN)
// tile and stages a (16 WM + 16 WN) x 64-k slab per step with coalesced 16-byte loads (one row = 256 contiguous bytes per 16 lanes),
// register-prefetched two steps ahead into one of three LDS buffers; wave (wm, wn) reads its two 16 x 16 fragments per block as one
// ds_read_b128 each (rows 68 floats apart: conflict-free) and runs the SAME v_mfma_f32_16x16x4_f32 sequence over ascending k as
// gemm_directL16_kernel -- bit-identical results, 2 WM WN / (WM + WN) times fewer operand bytes per FLOP.
constexpr int T16_BKS = 64, T16_STR = T16_BKS + 4;

template <int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_tile16_kernel(GemmArgs g) {
    constexpr int T = 64 * WM * WN, RA = 16 * WM, RB = 16 * WN, NB = 3, PF = 2;
    constexpr int A_SZ = RA * T16_STR, BUF = (RA + RB) * T16_STR;
    constexpr int SLOTS = (RA + RB) * (T16_BKS / 4), NV = (SLOTS + T - 1) / T;  // 16-byte slots of a slab, per thread
    extern __shared__ __attribute__((aligned(16))) float t16_lds[];
    float *lds = t16_lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int i16 = lane & 15, q = lane >> 4;
    const int nbx = (int)((g.N + RB - 1) / RB), nby = (int)((g.M + RA - 1) / RA);
    const int tile = xcd_slot((int)blockIdx.x, nbx * nby);
    const int by = tile / nbx, bx = tile - by * nbx;
    const int64_t m0 = (int64_t)by * RA, n0 = (int64_t)bx * RB;
    const int Ki = (int)g.K;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.A), 0, (int)(4 * ((g.M - 1) * g.lda + g.K)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.B), 0, (int)(4 * ((g.N - 1) * g.ldb + g.K)), 0x00020000);
    // this thread's slots: slot s < RA * 16 -> A row s / 16, else B row; k piece 4 (s % 16)
    // (WN == 4: the A slab's slots are exactly the threads' first slot, every later slot is B's -- the buffer resource of a load is
    //  then known at compile time, as the scalar operand it is)
    static_assert(WN == 4 && SLOTS == NV * T && RA * (T16_BKS / 4) == T, "slot v == 0 <-> A, v >= 1 <-> B");
    int off[NV], dst[NV], k4[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
        const int sl = tid + v * T;
        const bool a_slot = v == 0;
        const int r = (a_slot ? sl : sl - T) / (T16_BKS / 4);
        k4[v] = 4 * (sl % (T16_BKS / 4));
        const int64_t row = a_slot ? ((m0 + r < g.M) ? m0 + r : g.M - 1) : ((n0 + r < g.N) ? n0 + r : g.N - 1);
        off[v] = 4 * ((int)(row * (a_slot ? g.lda : g.ldb)) + k4[v]);
        dst[v] = (a_slot ? 0 : A_SZ) + r * T16_STR + k4[v];
    }
    const int nsteps = (Ki + T16_BKS - 1) / T16_BKS;
    f32x4 pf[PF][NV];
    auto fetch = [&](int slot, int s) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const bool in = s * T16_BKS + k4[v] < Ki;
            pf[slot][v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(v == 0 ? ra : rb, in ? off[v] : HV_OOB, s * (4 * T16_BKS), 0));
        }
    };
    auto stash = [&](int slot, int s, int buf) {
        float *base = lds + buf * BUF;
#pragma unroll
        for (int v = 0; v < NV; v++) {
            f32x4 x = pf[slot][v];
            const int k = s * T16_BKS + k4[v];
            if (k + 4 > Ki) {  // the piece straddles the end of the row: what follows K belongs to the next row
#pragma unroll
                for (int e = 0; e < 4; e++) x[e] = k + e < Ki ? x[e] : 0.0f;
            }
            *reinterpret_cast<f32x4 *>(base + dst[v]) = x;
        }
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 fa[2][4], fb[2][4];
    auto frags = [&](int fs, int buf) {
        const float *As = lds + buf * BUF + (16 * wm + i16) * T16_STR + 4 * q;
        const float *Bs = lds + buf * BUF + A_SZ + (16 * wn + i16) * T16_STR + 4 * q;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            fa[fs][u] = *reinterpret_cast<const f32x4 *>(As + 16 * u);
            fb[fs][u] = *reinterpret_cast<const f32x4 *>(Bs + 16 * u);
        }
    };
#pragma unroll
    for (int p = 0; p < PF; p++) fetch(p, p);
    stash(0, 0, 0);
    fetch(0, PF);
    stash(1 % PF, 1, 1);
    fetch(1 % PF, PF + 1);
    __syncthreads();
    frags(0, 0);
    constexpr int UN = 6;  // lcm(2 fragment sets, NB, PF)
    for (int s0 = 0; s0 < nsteps; s0 += UN) {
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int s = s0 + u;
            if (s < nsteps) {  // (uniform)
                const int nblk = (Ki - s * T16_BKS + 15) / 16;  // 16-blocks of this step that hold any k < K (>= 4: all of them)
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u & 1][0][s4], fb[u & 1][0][s4], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < nsteps) frags((u + 1) & 1, (u + 1) % NB);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 1; b < 4; b++)
                    if (b < nblk) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; s4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u & 1][b][s4], fb[u & 1][b][s4], acc, 0, 0, 0);
                    }
                if (s + 2 < nsteps) {
                    stash((u + 2) % PF, s + 2, (u + 2) % NB);
                    fetch((u + 2) % PF, s + 2 + PF);
                }
                __syncthreads();
            }
        }
    }
    // C/D map of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
    const int64_t col = n0 + 16 * wn + i16;
    if (col >= g.N) return;
    const float bias = g.bias != nullptr ? g.bias[col] : 0.0f;
    const float *mk = g.mask;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int64_t row = m0 + 16 * wm + 4 * q + r;
        if (row >= g.M) continue;
        float v = acc[r] + bias;
        if (g.aux != nullptr && g.epilogue < HIDVAE_EPI_DSILU && g.epilogue != HIDVAE_EPI_NONE) g.aux[row * g.ldaux + col] = v;
        v = apply_epilogue(g.epilogue, v, g.aux, row * g.ldaux + col, g.mask_scale);
        if (has_dropout(g)) v = v * dropout_factor(g, row, col);
        float *dstp = g.C + row * g.ldc + col;
        *dstp = g.accumulate ? *dstp + v : v;
    }
}

template <int WM, int WN>
int launch_tile16(const GemmArgs &g, hipStream_t s) {
    constexpr size_t bytes = (size_t)3 * (16 * WM + 16 * WN) * T16_STR * 4;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tile16_kernel<WM, WN>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (attr != hipSuccess) return -1;
    const int64_t tiles = hv_cdiv(g.M, 16 * WM) * hv_cdiv(g.N, 16 * WN);
    hipLaunchKernelGGL((gemm_tile16_kernel<WM, WN>), dim3((unsigned)tiles), dim3(64 * WM * WN), bytes, s, g);
    return 0;
}

template <int SPLIT>
void launch_directL(int layout, const GemmArgs &g, hipStream_t s) {
    dim3 grid((unsigned)hv_cdiv(g.N, 32), (unsigned)hv_cdiv(g.M, 32));
    dim3 block(64 * SPLIT);
    if (layout == HIDVAE_GEMM_NT) hipLaunchKernelGGL((gemm_directL_kernel<HIDVAE_GEMM_NT, SPLIT>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_directL_kernel<HIDVAE_GEMM_NN, SPLIT>), grid, block, 0, s, g);
}

template <int SPLIT, int NS, int NWN = 1>
void launch_direct(int layout, const GemmArgs &g, hipStream_t s) {
    dim3 grid((unsigned)hv_cdiv(hv_cdiv(g.N, 32), NWN), (unsigned)hv_cdiv(g.M, 32));
    dim3 block(64 * SPLIT * NWN);
    if (layout == HIDVAE_GEMM_NT) hipLaunchKernelGGL((gemm_direct_kernel<HIDVAE_GEMM_NT, SPLIT, NS, NWN>), grid, block, 0, s, g);
    else if (layout == HIDVAE_GEMM_NN) hipLaunchKernelGGL((gemm_direct_kernel<HIDVAE_GEMM_NN, SPLIT, NS, NWN>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_direct_kernel<HIDVAE_GEMM_TN, SPLIT, NS, NWN>), grid, block, 0, s, g);
}

template <int WM, int WN>
void launch_tile(int layout, const GemmArgs &g, int splits, hipStream_t s) {
    dim3 grid((unsigned)hv_cdiv(g.N, 32 * WN), (unsigned)hv_cdiv(g.M, 32 * WM), (unsigned)splits);
    dim3 block(64 * WM * WN);
    if (layout == HIDVAE_GEMM_NT) hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, HIDVAE_GEMM_NT>), grid, block, 0, s, g);
    else if (layout == HIDVAE_GEMM_NN) hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, HIDVAE_GEMM_NN>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, HIDVAE_GEMM_TN>), grid, block, 0, s, g);
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// dispatch rules shared by hidvae_gemm_f32 and the pair launch (so both choose the same split and give the same bits)
inline bool fits32bit(int layout, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb) {
    const int64_t a_rows = (layout == HIDVAE_GEMM_TN) ? K : M, b_rows = (layout == HIDVAE_GEMM_NT) ? N : K;
    return a_rows * lda < (1ll << 29) && b_rows * ldb < (1ll << 29);  // the direct kernels use 32-bit BYTE offsets
}
inline bool use_direct16(int64_t M, int64_t N, int64_t K) {
    const int64_t tiles32 = hv_cdiv(M, 32) * hv_cdiv(N, 32);
    return tiles32 <= 256 || (tiles32 <= 512 && K <= 256);
}
inline int pick_split32(int64_t M, int64_t N, int64_t K, int split_k) {
    const int64_t tiles32 = hv_cdiv(M, 32) * hv_cdiv(N, 32);
    int sp = 1;
    if (split_k != 1) {
        const int cap = split_k == 0 ? 16 : split_k;
        while (sp < cap && sp < 16 && tiles32 * sp < 2048 && K / (sp * 2) >= 32) sp *= 2;
    }
    return sp;
}
inline int pick_split16(int64_t M, int64_t N, int64_t K, int split_k) {
    const int64_t tiles16 = hv_cdiv(M, 16) * hv_cdiv(N, 16);
    int sp16 = 1;
    if (split_k != 1) {
        const int cap = split_k == 0 ? 16 : split_k;
        while (sp16 < cap && sp16 < 16 && tiles16 * sp16 < 4096 && K / (sp16 * 2) >= 32) sp16 *= 2;
    }
    return sp16;
}

// ---- grouped launch: up to GROUP_MAX independent sub-problems (forward Linear layers of the three tag-head levels; or their
// backward: dW, dX and the bias column sums of every level) in ONE grid.  A sub-problem runs the same device bodies as
// gemm_pair16/32_kernel (16x16 or 32x32 tiles chosen by use_direct16, K shared out over `split` <= 8 waves of an 8-wave workgroup,
// partials added in ascending wave order), so every sub-problem is deterministic; the point of the group is that the small
// problems' launch / drain latencies hide under the big problems' arithmetic instead of being paid one after the other.
constexpr int GROUP_MAX = 12;
constexpr int GROUP_WAVES = 8;
struct GroupSub {
    GemmArgs g;    // kind 2 (column sums): A = X [M rows, N cols] at lda, C = out [N], accumulate
    int kind;      // 0: 16x16-tile GEMM, 1: 32x32-tile GEMM, 2: column sums
    int layout, split, deep, nbx;
    int64_t nt;    // tiles
    int nb;        // workgroups
};
struct GroupArgs {
    int n;
    GroupSub s[GROUP_MAX];
};

__device__ __forceinline__ void colsum_group_body(const GemmArgs &g, int64_t c0, float *part) {
    const int col = threadIdx.x & 31, rg = threadIdx.x >> 5, nrg = (int)(blockDim.x >> 5);
    const int64_t c = c0 + col;
    float acc = 0.0f;
    if (c < g.N)
        for (int64_t r0 = rg; r0 < g.M; r0 += 8 * (int64_t)nrg) {  // eight loads in flight, ascending order of additions
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = r0 + (int64_t)nrg * j < g.M ? g.A[(r0 + (int64_t)nrg * j) * g.lda + c] : 0.0f;
#pragma unroll
            for (int j = 0; j < 8; j++) acc += v[j];
        }
    part[rg * 32 + col] = acc;
    __syncthreads();
    if (threadIdx.x < 32 && c < g.N) {
        float v = part[col];
        for (int j = 1; j < nrg; j++) v += part[j * 32 + col];
        g.C[c] = g.accumulate ? g.C[c] + v : v;
    }
}

__global__ __launch_bounds__(64 * GROUP_WAVES) void gemm_group_kernel(GroupArgs a) {
    __shared__ float part[GROUP_WAVES * 1024];
    int bid = blockIdx.x, p = 0;
    while (p + 1 < a.n && bid >= a.s[p].nb) {
        bid -= a.s[p].nb;
        p++;
    }
    const GroupSub &s = a.s[p];
    if (s.kind == 2) {
        colsum_group_body(s.g, (int64_t)bid * 32, part);
        return;
    }
    const int64_t tile0 = (int64_t)xcd_slot(bid, s.nb) * (GROUP_WAVES / s.split);
    if (s.kind == 0) {
        if (s.layout == HIDVAE_GEMM_NT) {
            if (s.deep) direct16_body<HIDVAE_GEMM_NT, 6>(s.g, s.split, tile0, s.nt, s.nbx, part);
            else direct16_body<HIDVAE_GEMM_NT, 3>(s.g, s.split, tile0, s.nt, s.nbx, part);
        } else if (s.layout == HIDVAE_GEMM_NN) direct16_body<HIDVAE_GEMM_NN, 3>(s.g, s.split, tile0, s.nt, s.nbx, part);
        else direct16_body<HIDVAE_GEMM_TN, 3>(s.g, s.split, tile0, s.nt, s.nbx, part);
    } else {
        if (s.layout == HIDVAE_GEMM_NT) {
            if (s.deep) direct32_body<HIDVAE_GEMM_NT, 6>(s.g, s.split, tile0, s.nt, s.nbx, part);
            else direct32_body<HIDVAE_GEMM_NT, 3>(s.g, s.split, tile0, s.nt, s.nbx, part);
        } else if (s.layout == HIDVAE_GEMM_NN) {
            if (s.deep) direct32_body<HIDVAE_GEMM_NN, 6>(s.g, s.split, tile0, s.nt, s.nbx, part);
            else direct32_body<HIDVAE_GEMM_NN, 3>(s.g, s.split, tile0, s.nt, s.nbx, part);
        } else direct32_body<HIDVAE_GEMM_TN, 3>(s.g, s.split, tile0, s.nt, s.nbx, part);
    }
}

// fill one GEMM sub-problem with the group's dispatch rule; false: the problem is outside the direct kernels' regime
inline bool group_gemm_sub(GroupSub &s, int layout, const GemmArgs &g) {
    if (!fits32bit(layout, g.M, g.N, g.K, g.lda, g.ldb) || hv_cdiv(g.M, 32) * hv_cdiv(g.N, 32) >= 2048) return false;
    s.g = g;
    s.layout = layout;
    const bool f16 = use_direct16(g.M, g.N, g.K);
    s.kind = f16 ? 0 : 1;
    int sp = f16 ? pick_split16(g.M, g.N, g.K, 0) : pick_split32(g.M, g.N, g.K, 0);
    // in-workgroup K split of a grouped sub-problem (measured: 1 -> 63 us, 2 -> 58 us, 4 -> 52 us for the three levels' widest layer);
    // a handful of tiles with a long K (the attention gate's weight gradients: 2-8 tiles, K = batch) takes all eight waves
    const int cap = hv_cdiv(g.M, 16) * hv_cdiv(g.N, 16) <= 16 ? GROUP_WAVES : 4;
    if (sp > GROUP_WAVES) sp = GROUP_WAVES;
    while (sp > cap && sp > 1) sp /= 2;
    s.split = sp;
    s.deep = g.K / (16 * sp) >= 12;
    const int T = f16 ? 16 : 32;
    s.nbx = (int)hv_cdiv(g.N, T);
    s.nt = (int64_t)s.nbx * hv_cdiv(g.M, T);
    s.nb = (int)hv_cdiv(s.nt, GROUP_WAVES / sp);
    return true;
}

inline int launch_group(GroupArgs &a, hipStream_t s) {
    int64_t blocks = 0;
    for (int i = 0; i < a.n; i++) blocks += a.s[i].nb;
    if (blocks == 0) return HIDVAE_OK;
    hipLaunchKernelGGL(gemm_group_kernel, dim3((unsigned)blocks), dim3(64 * GROUP_WAVES), 0, s, a);
    HV_LAUNCH_CHECK("gemm_group");
    return HIDVAE_OK;
}

}  // namespace

extern "C" int hidvae_gemm_f32(int layout, int64_t M, int64_t N, int64_t K, const float *A, int64_t lda,
                               const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc, int epilogue,
                               float *aux, int64_t ldaux, const float *mask, int64_t ldmask, float mask_scale,
                               const unsigned long long *rng_state, unsigned rng_site, unsigned drop_threshold,
                               int split_k, float *workspace, int accumulate, void *stream) {
    HV_REQUIRE(layout >= 0 && layout <= 2, "gemm: layout %d", layout);
    HV_REQUIRE(!(mask && rng_state), "gemm: a keep-mask OR the in-kernel generator, not both");
    HV_REQUIRE(M >= 1 && N >= 1 && K >= 1, "gemm: empty problem M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    HV_REQUIRE(A && B && C, "gemm: null operand");
    const int64_t a_min = (layout == HIDVAE_GEMM_TN) ? M : K, b_min = (layout == HIDVAE_GEMM_NT) ? K : N;
    HV_REQUIRE(lda >= a_min && ldb >= b_min && ldc >= N, "gemm: leading dimension too small (lda=%lld ldb=%lld ldc=%lld)",
               (long long)lda, (long long)ldb, (long long)ldc);
    HV_REQUIRE(epilogue < HIDVAE_EPI_DSILU || (aux != nullptr && ldaux >= N), "gemm: backward epilogue %d needs aux", epilogue);
    HV_REQUIRE(mask == nullptr || ldmask >= N, "gemm: ldmask=%lld", (long long)ldmask);
    HV_REQUIRE(split_k >= 0, "gemm: split_k=%d", split_k);
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.bias = bias; g.C = C; g.ldc = ldc;
    g.epilogue = epilogue; g.aux = aux; g.ldaux = aux ? ldaux : 0; g.accumulate = accumulate;
    g.vecA = (lda % 4 == 0) && aligned16(A);
    g.vecB = (ldb % 4 == 0) && aligned16(B);
    g.mask = mask; g.ldmask = ldmask; g.mask_scale = mask_scale;
    g.drop = HvDrop{rng_state, rng_site, drop_threshold};
    hipStream_t s = (hipStream_t)stream;
    const int64_t tiles32 = hv_cdiv(M, 32) * hv_cdiv(N, 32);
    const bool fits32 = fits32bit(layout, M, N, K, lda, ldb);
    // enough tiles that LDS sharing beats per-wave operand loads (measured crossover: 2048 tiles -- at 4096 rows the encoder's
    // first layer takes 46 us tiled vs 52 us direct, at 2048 rows 33 vs 30)
    // deep-K problems with few output tiles (weight gradients at large batch: K = batch): the LDS-tiled kernel with the K range
    // cut into grid-level slabs, summed in slab order by splitk_reduce_kernel.  Needs the workspace (slabs * M * N floats).
    int deep_splits = 0;
    if (split_k == 0 && workspace != nullptr && K >= 4096 && tiles32 < 2048 && M > 32 && N > 32) {
        deep_splits = (int)(K / 512 < 16 ? K / 512 : 16);
        const int64_t wgs = hv_cdiv(M, 64) * hv_cdiv(N, 64);
        while (deep_splits > 2 && wgs * deep_splits > 2048) deep_splits /= 2;
    }
    const bool big = (tiles32 >= 2048 && K >= 64) || !fits32 || deep_splits > 1;
    if (!big || split_k == 1) {
        if (!big) {
            // direct path.  split_k == 1: one wave per tile, sequential (ORDER-G) chain; otherwise spread K over up to 16
            // waves of the workgroup until the chip has ~2 waves per SIMD or the chunks get shorter than 32
            // few tiles: 16x16 tiles (4x the waves, shorter dependent MFMA chains) -- the latency-optimised form
            // NT with ONE exact chain per output (the id-determining forward layers), and the small NT problems the 16x16 form serves
            // anyway: 16x16 tiles with coalesced operand loads through per-wave LDS strips (gemm_directL16_kernel).  Bit-identical to
            // the register-path kernels (same ORDER-G16 chain); measured at B = 1024: 1024x512x768 19.5 -> 15.5 us, 1024x768x512
            // 19.5 -> 13.2, 1024x256x512 10.8 -> 7.9, 1024x512x256 11.3 -> 7.4, 1000x300x333 12.7 -> 8.6
            const int64_t nt16 = hv_cdiv(M, 16) * hv_cdiv(N, 16);
            // the widest of them on LDS-shared 32x64 tiles (gemm_tile16_kernel): same chain, same bits, a third of the operand traffic
            // -- where the tiles fill the chip in ONE round (192 .. 256 of them): 1024x512x768 16.1 -> 13.5 us (32x64), 2048x512x768
            // 24.9 -> 20.7 (64x64); with 1.5 rounds (1024x768x512 on 32x64 tiles) the finer-grained kernel below wins, 14.8 vs 16.3
            const int64_t t_small = hv_cdiv(M, 32) * hv_cdiv(N, 64), t_big = hv_cdiv(M, 64) * hv_cdiv(N, 64);
            const bool fit_small = t_small >= 192 && t_small <= 256, fit_big = t_big >= 224 && t_big <= 256;  // (192 64x64 tiles: no gain)
            // the wide layers that may split K (the tag predictors of levels 1 / 2: 460 .. 768 wide) on the same kernel's 64x64 tiles.
            // Stand-alone gemm_directL_kernel<NT, 4> (every wave its own operand strips, K over four waves) is as fast; inside the
            // three-lane step it is not -- per-wave operand traffic is what the lanes compete for -- and the shared tiles take the
            // tagged step from 1.045 to 1.032 ms at B = 1024, 1.702 to 1.690 at B = 2048 (profiles/r04_forward_shared_tiles_ab.log;
            // not kept there: 32x64 tiles (equal), the 2x2-wave 32x32-MFMA tile kernel (+10 %), the same move for 224 .. 447 tiles
            // or for the exact-chain layers outside the one-round window (no change))
            if (layout == HIDVAE_GEMM_NT && split_k == 0 && K >= 256 && tiles32 >= 448) {
                HV_REQUIRE((launch_tile16<4, 4>(g, s)) == 0, "gemm_f32: could not size the LDS of gemm_tile16_kernel");
                HV_LAUNCH_CHECK("gemm_f32 tile16");
                return HIDVAE_OK;
            }
            if (layout == HIDVAE_GEMM_NT && split_k == 1 && K >= 256 && (fit_small || fit_big)) {
                const int rc = !fit_small ? launch_tile16<4, 4>(g, s) : launch_tile16<2, 4>(g, s);
                HV_REQUIRE(rc == 0, "gemm_f32: could not size the LDS of gemm_tile16_kernel");
                HV_LAUNCH_CHECK("gemm_f32 tile16");
                return HIDVAE_OK;
            }
            if (layout == HIDVAE_GEMM_NT && nt16 >= 512 && nt16 <= 8192 && K >= 64 && (split_k == 1 || (use_direct16(M, N, K) && K >= 128))) {
                constexpr int W16 = 4;
                hipLaunchKernelGGL(gemm_directL16_kernel<W16>, dim3((unsigned)hv_cdiv(nt16, W16)), dim3(64 * W16), 0, s, g);
                HV_LAUNCH_CHECK("gemm_f32 directL16");
                return HIDVAE_OK;
            }
            if (use_direct16(M, N, K)) {
                const int sp16 = pick_split16(M, N, K, split_k);
                const bool deep16 = K / (16 * sp16) >= 12;
                switch (sp16) {
                    case 1: if (deep16) launch_direct16<1, 6>(layout, g, s); else launch_direct16<1, 3>(layout, g, s); break;
                    case 2: if (deep16) launch_direct16<2, 6>(layout, g, s); else launch_direct16<2, 3>(layout, g, s); break;
                    case 4: if (deep16) launch_direct16<4, 6>(layout, g, s); else launch_direct16<4, 3>(layout, g, s); break;
                    case 8: launch_direct16<8, 3>(layout, g, s); break;
                    default: launch_direct16<16, 3>(layout, g, s); break;
                }
                HV_LAUNCH_CHECK("gemm_f32 direct16");
                return HIDVAE_OK;
            }
            const int sp = pick_split32(M, N, K, split_k);
            // the LDS-transposed loader where it measured faster: K shared out over >= 2 waves per tile, or >= 1024 tiles (one exact chain
            // per tile at 512 tiles -- the encoder's first layer at B = 1024 -- is bound by the chain itself: 21.5 us against 18.4 us)
            if (layout != HIDVAE_GEMM_TN && sp <= 8 && (sp >= 2 || tiles32 >= 1024)) {
                switch (sp) {
                    case 1: launch_directL<1>(layout, g, s); break;
                    case 2: launch_directL<2>(layout, g, s); break;
                    case 4: launch_directL<4>(layout, g, s); break;
                    default: launch_directL<8>(layout, g, s); break;
                }
                HV_LAUNCH_CHECK("gemm_f32 directL");
                return HIDVAE_OK;
            }
            const bool deep = K / (16 * sp) >= 12;  // long chains per wave: keep 5 blocks of loads in flight instead of 2
            switch (sp) {
                case 1: if (deep) launch_direct<1, 6>(layout, g, s); else launch_direct<1, 3>(layout, g, s); break;
                case 2: if (deep) launch_direct<2, 6>(layout, g, s); else launch_direct<2, 3>(layout, g, s); break;
                case 4: if (deep) launch_direct<4, 6>(layout, g, s); else launch_direct<4, 3>(layout, g, s); break;
                case 8: launch_direct<8, 3>(layout, g, s); break;
                default: launch_direct<16, 3>(layout, g, s); break;
            }
            HV_LAUNCH_CHECK("gemm_f32 direct");
            return HIDVAE_OK;
        }
    }
    // LDS-tiled path (large batches); grid-level split-K through the workspace when asked for and available
    int splits = (split_k > 1 && workspace != nullptr) ? split_k : (deep_splits > 1 ? deep_splits : 1);
    int64_t kps = hv_cdiv(hv_cdiv(K, splits), BK) * BK;
    splits = (int)hv_cdiv(K, kps);
    g.k_per_split = kps;
    g.partial = splits > 1 ? workspace : nullptr;
    if (N <= 32) launch_tile<2, 1>(layout, g, splits, s);
    else if (M <= 32) launch_tile<1, 2>(layout, g, splits, s);
    else launch_tile<2, 2>(layout, g, splits, s);
    HV_LAUNCH_CHECK("gemm_f32 tiled");
    if (splits > 1) {
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)hv_cdiv(M * N, 256)), dim3(256), 0, s, g, splits);
        HV_LAUNCH_CHECK("splitk_reduce");
    }
    return HIDVAE_OK;
}

extern "C" int hidvae_colsum(const float *X, int64_t M, int64_t N, int64_t ldx, float *out, int accumulate, float *workspace, void *stream);

// balanced_ok: the workspace follows the contract of HIDVAE_WS_LINEAR_BWD_ZEROED (hidvae_linear_bwd); the grouped entry point's
// per-problem fallback passes false -- its workspaces are plain scratch
static int linear_bwd_impl(const float *g, int64_t ldg, const float *x, int64_t ldx, const float *W, int64_t ldw, int64_t B,
                           int64_t n_out, int64_t n_in, float *dW, int64_t lddw, int accumulate_dw, float *dX, int64_t lddx,
                           int dx_epilogue, float *aux, int64_t ldaux, float dx_scale, float *db, int accumulate_db, float *workspace,
                           void *stream, bool balanced_ok, bool co_resident = false) {
    HV_REQUIRE(g && x && dW && B >= 1 && n_out >= 1 && n_in >= 1, "linear_bwd: bad arguments");
    HV_REQUIRE(ldg >= n_out && ldx >= n_in && lddw >= n_in, "linear_bwd: leading dimension too small");
    HV_REQUIRE(dX == nullptr || (W != nullptr && ldw >= n_in && lddx >= n_in), "linear_bwd: dX needs W");
    HV_REQUIRE(dX == nullptr || dx_epilogue == HIDVAE_EPI_NONE || (dx_epilogue >= HIDVAE_EPI_DSILU && aux != nullptr && ldaux >= n_in),
               "linear_bwd: dX epilogue %d", dx_epilogue);
    // dW [n_out, n_in] = g^T x : TN with M = n_out, N = n_in, K = B;   dX [B, n_in] = g W : NN with M = B, N = n_in, K = n_out
    const bool small = dX != nullptr && fits32bit(HIDVAE_GEMM_TN, n_out, n_in, B, ldg, ldx) && fits32bit(HIDVAE_GEMM_NN, B, n_in, n_out, ldg, ldw) &&
                       hv_cdiv(n_out, 32) * hv_cdiv(n_in, 32) < 2048 && hv_cdiv(B, 32) * hv_cdiv(n_in, 32) < 2048;
    const bool pair16 = small && use_direct16(n_out, n_in, B) && use_direct16(B, n_in, n_out);
    const int s32_0 = pick_split32(n_out, n_in, B, 0), s32_1 = pick_split32(B, n_in, n_out, 0);
    const bool pair32 = small && !use_direct16(n_out, n_in, B) && !use_direct16(B, n_in, n_out) && s32_0 <= 8 && s32_1 <= 8;
    const bool pair = pair16 || pair32;
    // LDS-shared 64x64 tiles, evenly dealt (gemm_mid_sk_kernel<4, 2>: 16 waves = 2x2 quarters x 4 k-groups, register prefetch two steps
    // ahead; <2, 2>: 8 waves, for launches that share the chip with other streams' -- see co_resident below; not kept: four prefetch stages)
    constexpr int mid_minq = 6;  // shortest range worth a workgroup (steps)
    // (the rule looks at the layer, not at whether this call wants dX: a weight-gradient-only call on a balanced shape -- the one-launch
    //  predictor backward's, a projector's first layer -- takes the ring kernel too, and the workspace query says the same)
    if (balanced_ok && workspace != nullptr && hv_lbwd_balanced(B, n_out, n_in, true) &&
        fits32bit(HIDVAE_GEMM_TN, n_out, n_in, B, ldg, ldx) && (dX == nullptr || fits32bit(HIDVAE_GEMM_NN, B, n_in, n_out, ldg, ldw))) {
        // the LDS-DMA ring kernel (gemm_ring.hip) wherever it takes the problem; the kernels below are what runs when it declines.
        // (First shipped from B = 2048 on, where it won stand-alone; in the step it also wins at B = 1024: tagged 1.156 -> 1.088 ms
        // together with the lower threshold of rules.h -- profiles/r04_ring_threshold_ab.log.)
        {
            const int rc = hv_ring_linear_bwd(g, ldg, x, ldx, W, ldw, B, n_out, n_in, dW, lddw, accumulate_dw, dX, lddx, dx_epilogue, aux, ldaux,
                                              dx_scale, db, accumulate_db, workspace, 512, (hipStream_t)stream);
            if (rc != 1) return rc;
        }
        auto run_mid = [&](auto kg_tag) -> int {
        constexpr int KG = decltype(kg_tag)::value, BKS = 16 * KG;
        SkArgs a{};
        PairArgs &p = a.p;
        p.g0.M = n_out; p.g0.N = n_in; p.g0.K = B; p.g0.A = g; p.g0.lda = ldg; p.g0.B = x; p.g0.ldb = ldx; p.g0.C = dW; p.g0.ldc = lddw;
        p.g0.epilogue = HIDVAE_EPI_NONE; p.g0.mask_scale = 1.0f; p.g0.accumulate = accumulate_dw;
        p.nbx0 = (int)hv_cdiv(n_in, 64);
        p.nb0 = p.nbx0 * (int)hv_cdiv(n_out, 64);
        a.n0 = (int)hv_cdiv(B, BKS);
        if (dX != nullptr) {
            p.g1.M = B; p.g1.N = n_in; p.g1.K = n_out; p.g1.A = g; p.g1.lda = ldg; p.g1.B = W; p.g1.ldb = ldw; p.g1.C = dX; p.g1.ldc = lddx;
            p.g1.epilogue = dx_epilogue; p.g1.aux = aux; p.g1.ldaux = aux ? ldaux : 0; p.g1.mask_scale = dx_scale;
            p.nbx1 = (int)hv_cdiv(n_in, 64);
            p.nb1 = p.nbx1 * (int)hv_cdiv(B, 64);
            a.n1 = (int)hv_cdiv(n_out, BKS);
        } else a.n1 = 1;
        p.cs_x = g; p.cs_ld = ldg; p.cs_rows = B; p.cs_cols = n_out; p.cs_out = db; p.cs_accumulate = accumulate_db;
        a.nbc = db != nullptr ? (int)hv_cdiv(n_out, 32) : 0;
        a.cu = (int)hv_cdiv(B, 128 * KG);  // a 32-column strip of B rows, priced against a step's MFMAs (~1 us per 512 rows either way)
        a.S = p.nb0 * a.n0 + p.nb1 * a.n1 + a.nbc * a.cu;
        constexpr int slots = 1024 / KG;  // workgroups the chip holds at once: one sixteen-wave workgroup per CU (KG = 4), two eight-wave ones (KG = 2), four four-wave ones (KG = 1)
        int G = a.S / mid_minq;
        if (G > slots) G = slots;
        if (G < 1) G = 1;
        a.q = (int)hv_cdiv(a.S, G);
        a.G = (int)hv_cdiv(a.S, a.q);
        // one tile per workgroup when the tiles fit the chip in one round and are short (B = 1024: 24.8 us against 29.1 for the 768 x 512
        // layer -- a range that ends inside a tile costs ~3.5 us of pipeline refill and hand-over); even ranges otherwise (B = 2048:
        // 45.6 against 80.3 us)
        const int longest = a.n0 > (p.nb1 ? a.n1 : 0) ? a.n0 : a.n1;
        const bool tiles = p.nb0 + p.nb1 + a.nbc <= slots && longest * KG <= 96;
        if (tiles) {
            a.q = 0;
            a.G = p.nb0 + p.nb1 + a.nbc;
        }
        if (p.nb0 + p.nb1 <= HV_SK_COUNTERS && (tiles || a.G <= HV_SK_MAX_G)) {
            a.counters = reinterpret_cast<int *>(workspace);
            a.slabs = workspace + HV_SK_COUNTERS;
            const int rc = launch_mid_sk<KG, 2>(a, (hipStream_t)stream);
            HV_REQUIRE(rc == 0, "linear_bwd: could not size the LDS of gemm_mid_sk_kernel");
            HV_LAUNCH_CHECK("linear_bwd mid");
            return HIDVAE_OK;
        }
        return 1;  // not taken
        };
        // co_resident (the caller runs launches on other streams beside this one -- the tag heads' level streams): EIGHT-wave workgroups,
        // one whole tile each, two fit a CU.  Alone such a launch is 0-6 % slower than the sixteen-wave form (768 x 512 at B = 1024:
        // 29.2 vs 27.7 us; 691 x 768: 41.4 vs 44.8), but the sixteen-wave workgroup takes every register of its CU for 30-45 us, so
        // nothing of another stream can start there; with the smaller workgroups the launches of the three level streams overlap each
        // other's fill and drain: tagged step 1.385 -> 1.320 ms (four-wave workgroups, four per CU: 1.340).  Same scheduling rules with
        // 512 workgroup slots instead of 256 (whole tiles when they fit and K <= 1536, equal ranges otherwise: B = 2048 2.32 -> 2.27 ms).
        int rc_mid = co_resident ? run_mid(std::integral_constant<int, 2>{}) : 1;
        if (rc_mid == 1) rc_mid = run_mid(std::integral_constant<int, 4>{});
        if (rc_mid != 1) return rc_mid;
    }
    // (a workspace of a balanced-kernel shape keeps its leading counters to itself, whichever path this call takes)
    if (balanced_ok && workspace != nullptr && hv_lbwd_balanced(B, n_out, n_in, true)) workspace += HV_SK_COUNTERS;
    HV_REQUIRE(db == nullptr || pair || workspace != nullptr || B <= 16384, "linear_bwd: the unpaired bias gradient needs the colsum workspace");
    if (!pair) {
        // (the workspace serves the bias column sums first -- only for B > 16384 -- and then, in stream order, the slabs of dW)
        int rc = HIDVAE_OK;
        if (db != nullptr) rc = hidvae_colsum(g, B, n_out, ldg, db, accumulate_db, workspace, stream);
        if (rc == HIDVAE_OK)
            rc = hidvae_gemm_f32(HIDVAE_GEMM_TN, n_out, n_in, B, g, ldg, x, ldx, nullptr, dW, lddw, HIDVAE_EPI_NONE, nullptr, 0, nullptr, 0,
                                 1.0f, nullptr, 0u, 0u, 0, workspace, accumulate_dw, stream);
        if (rc != HIDVAE_OK || dX == nullptr) return rc;
        return hidvae_gemm_f32(HIDVAE_GEMM_NN, B, n_in, n_out, g, ldg, W, ldw, nullptr, dX, lddx, dx_epilogue, aux, ldaux, nullptr, 0, dx_scale,
                               nullptr, 0u, 0u, 0, nullptr, 0, stream);
    }
    PairArgs p{};
    p.g0.M = n_out; p.g0.N = n_in; p.g0.K = B; p.g0.A = g; p.g0.lda = ldg; p.g0.B = x; p.g0.ldb = ldx; p.g0.C = dW; p.g0.ldc = lddw;
    p.g0.epilogue = HIDVAE_EPI_NONE; p.g0.mask_scale = 1.0f; p.g0.accumulate = accumulate_dw;
    p.g1.M = B; p.g1.N = n_in; p.g1.K = n_out; p.g1.A = g; p.g1.lda = ldg; p.g1.B = W; p.g1.ldb = ldw; p.g1.C = dX; p.g1.ldc = lddx;
    p.g1.epilogue = dx_epilogue; p.g1.aux = aux; p.g1.ldaux = aux ? ldaux : 0; p.g1.mask_scale = dx_scale;
    const int T = pair16 ? 16 : 32;
    p.split0 = pair16 ? pick_split16(n_out, n_in, B, 0) : s32_0;
    p.split1 = pair16 ? pick_split16(B, n_in, n_out, 0) : s32_1;
    p.nbx0 = (int)hv_cdiv(n_in, T);
    p.nt0 = (int64_t)p.nbx0 * hv_cdiv(n_out, T);
    p.nbx1 = (int)hv_cdiv(n_in, T);
    p.nt1 = (int64_t)p.nbx1 * hv_cdiv(B, T);
    int waves = p.split0 > p.split1 ? p.split0 : p.split1;  // both splits are powers of two (<= 16, <= 8 for 32x32 tiles)
    if (waves < 4) waves = 4;
    p.nb0 = (int)hv_cdiv(p.nt0, waves / p.split0);
    p.nb1 = (int)hv_cdiv(p.nt1, waves / p.split1);
    p.cs_x = g; p.cs_ld = ldg; p.cs_rows = B; p.cs_cols = n_out; p.cs_out = db; p.cs_accumulate = accumulate_db;
    const int nbc = db != nullptr ? (int)hv_cdiv(n_out, 32) : 0;
    // (measured in round 2 and removed: staging the NN half's A operand through LDS made the paired launch SLOWER, tagged step 1.913 ->
    //  1.959 ms -- its row-contiguous B operand still takes 8 scalar loads per block and the 2-block stage is shallower than the 6-deep
    //  register ring)
    p.lds_a = 0;
    const dim3 grid((unsigned)(p.nb0 + p.nb1 + nbc)), block(64 * waves);
    const size_t lds32 = (size_t)waves * (4096 + (p.lds_a ? 32 * LSTR * 4 : 0));
    if (pair16) hipLaunchKernelGGL(gemm_pair16_kernel, grid, block, 0, (hipStream_t)stream, p);
    else if (n_out / (16 * p.split1) >= 12) hipLaunchKernelGGL(gemm_pair32_kernel<6>, grid, block, lds32, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gemm_pair32_kernel<3>, grid, block, lds32, (hipStream_t)stream, p);
    HV_LAUNCH_CHECK("linear_bwd pair");
    return HIDVAE_OK;
}

extern "C" int hidvae_linear_bwd(const float *g, int64_t ldg, const float *x, int64_t ldx, const float *W, int64_t ldw, int64_t B,
                                 int64_t n_out, int64_t n_in, float *dW, int64_t lddw, int accumulate_dw, float *dX, int64_t lddx,
                                 int dx_epilogue, float *aux, int64_t ldaux, float dx_scale, float *db, int accumulate_db, float *workspace,
                                 int co_resident, void *stream) {
    return linear_bwd_impl(g, ldg, x, ldx, W, ldw, B, n_out, n_in, dW, lddw, accumulate_dw, dX, lddx, dx_epilogue, aux, ldaux, dx_scale, db,
                           accumulate_db, workspace, stream, true, co_resident != 0);
}

extern "C" int hidvae_colsum(const float *X, int64_t M, int64_t N, int64_t ldx, float *out, int accumulate,
                             float *workspace, void *stream) {
    HV_REQUIRE(X && out && M >= 1 && N >= 1 && ldx >= N, "colsum: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (M <= 16384) {
        hipLaunchKernelGGL(colsum_one_kernel, dim3((unsigned)hv_cdiv(N, 32)), dim3(1024), 0, s, X, M, N, ldx, out, accumulate);
        HV_LAUNCH_CHECK("colsum_one");
        return HIDVAE_OK;
    }
    HV_REQUIRE(workspace != nullptr, "colsum: M > 16384 needs the workspace");
    const int64_t chunks = hv_cdiv(M, 64);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)hv_cdiv(N, 64), (unsigned)chunks), dim3(256), 0, s, X, M, N,
                       ldx, workspace);
    HV_LAUNCH_CHECK("colsum_partial");
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)hv_cdiv(N, 256)), dim3(256), 0, s, workspace, chunks, N, out,
                       accumulate);
    HV_LAUNCH_CHECK("colsum_final");
    return HIDVAE_OK;
}


// ---- grouped entry points (see gemm_group_kernel) -----------------------------------------------------------------------------
extern "C" int hidvae_linear_bwd_group(const hidvae_linear_bwd_problem *pr, int n, void *stream) {
    HV_REQUIRE(pr != nullptr && n >= 1, "linear_bwd_group: bad arguments");
    GroupArgs a{};
    int k = 0;
    bool ok = true;
    for (int i = 0; i < n && ok; i++) {
        const hidvae_linear_bwd_problem &q = pr[i];
        HV_REQUIRE(q.g && q.x && q.dW && q.B >= 1 && q.n_out >= 1 && q.n_in >= 1, "linear_bwd_group: problem %d is malformed", i);
        HV_REQUIRE(q.ldg >= q.n_out && q.ldx >= q.n_in && q.lddw >= q.n_in, "linear_bwd_group: problem %d: leading dimension too small", i);
        HV_REQUIRE(q.dX == nullptr || (q.W != nullptr && q.ldw >= q.n_in && q.lddx >= q.n_in), "linear_bwd_group: problem %d: dX needs W", i);
        HV_REQUIRE(q.dX == nullptr || q.dx_epilogue == HIDVAE_EPI_NONE || (q.dx_epilogue >= HIDVAE_EPI_DSILU && q.aux != nullptr && q.ldaux >= q.n_in),
                   "linear_bwd_group: problem %d: dX epilogue %d", i, q.dx_epilogue);
        if (k + 3 > GROUP_MAX) { ok = false; break; }
        GemmArgs g0{};  // dW [n_out, n_in] = g^T x : TN with M = n_out, N = n_in, K = B
        g0.M = q.n_out; g0.N = q.n_in; g0.K = q.B; g0.A = q.g; g0.lda = q.ldg; g0.B = q.x; g0.ldb = q.ldx; g0.C = q.dW; g0.ldc = q.lddw;
        g0.epilogue = HIDVAE_EPI_NONE; g0.mask_scale = 1.0f; g0.accumulate = q.accumulate_dw;
        ok = group_gemm_sub(a.s[k++], HIDVAE_GEMM_TN, g0);
        if (ok && q.dX != nullptr) {  // dX [B, n_in] = g W : NN with M = B, N = n_in, K = n_out
            GemmArgs g1{};
            g1.M = q.B; g1.N = q.n_in; g1.K = q.n_out; g1.A = q.g; g1.lda = q.ldg; g1.B = q.W; g1.ldb = q.ldw; g1.C = q.dX; g1.ldc = q.lddx;
            g1.epilogue = q.dx_epilogue; g1.aux = q.aux; g1.ldaux = q.aux ? q.ldaux : 0; g1.mask_scale = 1.0f;
            ok = group_gemm_sub(a.s[k++], HIDVAE_GEMM_NN, g1);
        }
        if (ok && q.db != nullptr) {
            GroupSub &c = a.s[k++];
            c.kind = 2;
            c.g.A = q.g; c.g.lda = q.ldg; c.g.M = q.B; c.g.N = q.n_out; c.g.C = q.db; c.g.accumulate = q.accumulate_db;
            c.nb = (int)hv_cdiv(q.n_out, 32);
        }
    }
    if (!ok) {
        for (int i = 0; i < n; i++) {
            const hidvae_linear_bwd_problem &q = pr[i];
            const int rc = linear_bwd_impl(q.g, q.ldg, q.x, q.ldx, q.W, q.ldw, q.B, q.n_out, q.n_in, q.dW, q.lddw, q.accumulate_dw, q.dX, q.lddx,
                                           q.dx_epilogue, q.aux, q.ldaux, 1.0f, q.db, q.accumulate_db, q.workspace, stream, false);
            if (rc != HIDVAE_OK) return rc;
        }
        return HIDVAE_OK;
    }
    a.n = k;
    return launch_group(a, (hipStream_t)stream);
}
