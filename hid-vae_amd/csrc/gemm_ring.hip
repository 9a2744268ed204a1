// Linear backward (dW = g^T x, dX = epi(g W), db = colsum(g)) on an LDS-DMA ring: every layer of the tag heads and of the encoder /
// decoder (reference modules/h_rqvae.py:132-188,322-331; modules/encoder.py:23-36) from 0.2 GFLOP on (rules.h), at every batch size.
//
// What differs from gemm_mid_sk_kernel (gemm.hip), which stages every operand global -> VGPR -> ds_write, runs sixteen (eight) waves per
// 64x64 tile in four (two) k-groups whose partial tiles meet in LDS at every segment end, and restarts its pipeline at every segment:
//   * a workgroup is FOUR waves, one 32x32 quarter of a 64x64 tile each and the whole K range of its segment: one accumulator chain
//     per wave, no k-groups, nothing to add up in LDS;
//   * operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR, no ds_write, one instruction per KiB), into a ring
//     of RB_NS stages of 32 k; the loads of step s + 3 are issued right after the barrier of step s and run through segment and tile
//     boundaries -- the ring never drains inside a workgroup's range except where a partial tile is handed over; counted
//     s_waitcnt vmcnt(N), one raw s_barrier per step;
//   * the work is gemm_mid_sk_kernel's list of k-steps cut into equal contiguous ranges (every SIMD the same number of MFMAs whatever
//     the tile count; partial tiles through the workspace, the last arriver adds them in ascending k order: deterministic), over a
//     tile list ordered in square-ish bands;
//   * the loader state (buffer descriptors, per-lane offsets, the cursor) lives in registers and is rebuilt only at a tile change:
//     indexing the argument struct by a run-time product index turns every field into a scalar load from the argument segment, eight
//     dependent ones per step (the first build: 1.7 us per step of nothing but that).
// Measured (scratch/r4): alone, 10-25 % faster than gemm_mid_sk_kernel on most shapes from B = 1024 up (1024 x 691 x 768:
// 31.6 vs 37.5 us; 2048 x 768 x 691: 55.6 vs 64.0 us = 0.50 of the fp32 MFMA peak; 4096 x 768 x 512: 71.8 vs 80.6 = 0.57; 8192 x 512 x 256:
// 51.9 vs 60.8; slower on 1024 x 768 x 512: 27.4 vs 22.9).  Inside the step it first replaced gemm_mid_sk_kernel from B = 2048 (tagged step
// 2.008 -> 1.881 ms, the B = 8192 shard 0.870 -> 0.830 ms) and tied at B = 1024; what moved the B = 1024 step was giving it the layers
// of 0.2-0.9 GFLOP that the per-wave pair kernels had (those run 1.5-2x slower in the three-lane step than alone): 1.156 -> 1.088 ms
// (profiles/r04_ring_threshold_ab.log).  Variants that were built, tested and NOT kept (sources: scratch/r4/gemm_ring_v3c.hip.txt): a
// chunk-major step list (L2 hits 39 % -> 81 %, fabric traffic 114 -> 27 MB per launch) with pieces parked in slabs and a second,
// combining launch -- slower in the step for the second launch's gap; two dedicated loader waves beside four MFMA waves -- no change:
// what does not overlap is not the DMA instructions' issue (ablation: MFMAs + LDS reads alone 17.4 us above the launch's fixed 8 us =
// the matrix pipe's floor at the ~2.0 GHz the chip holds; DMAs alone 9.7 us; together 23.6).
// LDS images (a stage = 16 KB = operand A, then operand B):
//   k-major operand (g and x of dW; W of dX), element (k, c) at P[k * ld + c]:   [32 k][64 c] floats, filled by 8 pieces of 4 k-rows;
//       the MFMA fragment of k-pair j is the dword at row 2j + h: 32 consecutive dwords per half-wave, conflict-free ds_read_b32;
//   k-contiguous operand (g of dX), element (m, k) at P[m * ld + k]:             [64 m][32 k], 128-byte rows, filled by 8 pieces of
//       8 rows; the 16-byte chunk c of row m sits at position c ^ ((m >> 1) & 7) -- the permutation is applied to the SOURCE address
//       of the DMA (its LDS side is lane-linear) and again on the read: conflict-free ds_read_b128.
// Edges: k rows past K are aimed out of the buffer (LDS-DMA writes zeros for them: scratch/r4/probe.hip); a 16-byte chunk of a
// k-contiguous row that straddles K brings in finite neighbours which meet those zeros; rows / columns past M / N only reach
// accumulator rows / columns that are never stored; the hardware's range check is per dword (same probe).  16-byte DMA needs dword
// alignment only, so 691-wide rows are fine.
// Not the ORDER-G16 chain: gradients only (their tests compare against float64 and demand launch-to-launch bit identity).
#include "common.h"
#include "rules.h"
#include "gemm_ring.h"

namespace {

constexpr int RB_BK = 32;            // k per step
constexpr int RB_NS = 4;             // ring stages
constexpr int RB_D = 3;              // steps the loads run ahead
constexpr int RB_STAGE = 4096;       // floats per stage: A image (2048) then B image (2048)
constexpr int RB_LDS_FLOATS = RB_NS * RB_STAGE + 64;  // + the flag word and the column sums' partials
constexpr int RB_OOB = 0x7FFFFFF0;   // a byte offset no buffer reaches
#define RB_LDSP(p) ((__attribute__((address_space(3))) void *)(p))

struct RingProb {
    const float *A, *B;
    float *C;
    const float *aux;
    int lda, ldb, ldc, ldaux;
    int M, N, K;
    int nbx, nby, ntiles, nsteps, bh;  // 64x64 tiles, steps of RB_BK per tile, band height of the tile order
    int epilogue, accumulate;
    float scale;
    unsigned a_bytes, b_bytes;
};

struct RingArgs {
    RingProb p[2];  // p[0]: TN (dW = g^T x), p[1]: NN (dX = g W; ntiles == 0: absent)
    const float *cs_x;
    float *cs_out;
    int cs_ld, cs_rows, cs_cols, cs_accumulate;
    int nbc, cu;    // column-sum units (32 columns each) and their price in steps
    int S, q, G;    // total steps, steps per workgroup, workgroups
    float *slabs;   // [G][2][4096] partial tiles (accumulator layout)
    int *counters;  // [ntiles0 + ntiles1], zero between launches
};

// tile t of the band order -> (by, bx): bands of `bh` tile rows, column-major inside a band
__device__ __forceinline__ void ring_tile_xy(const RingProb &P, int t, int &by, int &bx) {
    const int band_sz = P.bh * P.nbx;
    const int band = t / band_sz, r = t - band * band_sz;
    int hb = P.nby - band * P.bh;
    hb = hb < P.bh ? hb : P.bh;
    bx = r / hb;
    by = band * P.bh + (r - bx * hb);
}

struct Cursor {  // a position in the workgroup's list of k-steps (all wave-uniform)
    int prob, tile, ks;
};

__device__ __forceinline__ Cursor ring_locate(const RingArgs &a, int step) {
    Cursor c;
    const int S0 = a.p[0].ntiles * a.p[0].nsteps;
    c.prob = step >= S0 ? 1 : 0;
    const int rel = step - (c.prob ? S0 : 0), nT = a.p[c.prob].nsteps;
    c.tile = rel / nT;
    c.ks = rel - c.tile * nT;
    return c;
}
// The loader: where the DMA stream stands (it runs RB_D steps ahead of the MFMAs, through segment and tile boundaries), with everything a
// step needs held in registers -- the kernel arguments are read again only when the stream moves to another tile (indexing a.p[prob] at
// run time makes every field a scalar load from the argument segment: eight dependent ones per step cost more than the step's MFMAs)
struct Loader {
    int prob, tile, ks;       // cursor (wave-uniform)
    int nsteps, ntiles, K;    // of the cursor's problem
    int stepA, stepB;         // bytes per step
    __amdgpu_buffer_rsrc_t ra, rb;
    int voffA[2], voffB[2];   // per lane: byte offsets of this wave's two pieces per operand at k0 = 0 (RB_OOB: a row outside the matrix)
    int kA[2], kB[2];         // per lane: the piece's first k relative to the step (for the K edge)
};

__device__ __forceinline__ void loader_tile(const RingArgs &a, Loader &L, int wave, int lane) {
    const RingProb &P = a.p[L.prob];
    const bool NN = L.prob != 0;
    L.nsteps = P.nsteps; L.ntiles = P.ntiles; L.K = P.K;
    L.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.A), 0, (int)P.a_bytes, 0x00020000);
    L.rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.B), 0, (int)P.b_bytes, 0x00020000);
    int by, bx;
    ring_tile_xy(P, L.tile, by, bx);
    const int m0 = 64 * by, n0 = 64 * bx, lda = P.lda, ldb = P.ldb, M = P.M;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int p = wave + 4 * i;
        const int kr = 4 * p + (lane >> 4), c4 = 4 * (lane & 15);  // k-major image: row kr of the step, columns c4 .. c4 + 3
        L.kB[i] = kr;
        L.voffB[i] = 4 * (kr * ldb + n0 + c4);
        const int row = 8 * p + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);  // k-contiguous image: row, source chunk
        L.kA[i] = NN ? 4 * c : kr;
        L.voffA[i] = NN ? (m0 + row < M ? 4 * ((m0 + row) * lda + 4 * c) : RB_OOB) : 4 * (kr * lda + m0 + c4);
    }
    L.stepA = NN ? 4 * RB_BK : 4 * RB_BK * lda;
    L.stepB = 4 * RB_BK * ldb;
}

// the wave's four DMA instructions of the loader's step into ring buffer `buf`, then the cursor moves on
__device__ __forceinline__ void loader_issue(const RingArgs &a, Loader &L, int wave, int lane, float *lds, int buf) {
    float *dst = lds + buf * RB_STAGE;
    const int k0 = L.ks * RB_BK;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int p = wave + 4 * i;
        // (the whole byte offset rides in the VECTOR offset: the hardware's per-dword range check covers voffset, not soffset, so
        //  anything past the end of the matrix -- the overhang of its last rows' last chunks -- reads as zero instead of leaving the buffer)
        const int va = (k0 + L.kA[i] < L.K && L.voffA[i] != RB_OOB) ? L.voffA[i] + L.ks * L.stepA : RB_OOB;
        const int vb = k0 + L.kB[i] < L.K ? L.voffB[i] + L.ks * L.stepB : RB_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(L.ra, RB_LDSP(dst + 256 * p), 16, va, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(L.rb, RB_LDSP(dst + 2048 + 256 * p), 16, vb, 0, 0, 0);
    }
    if (++L.ks == L.nsteps) {
        L.ks = 0;
        if (++L.tile == L.ntiles) {
            L.tile = 0;
            L.prob++;
        }
        if (L.prob < 2) loader_tile(a, L, wave, lane);  // (past the last problem the stream has ended: nothing is issued any more)
    }
}

// one step's sixteen MFMAs of the wave's quarter from ring buffer `buf`
template <bool NN>
__device__ __forceinline__ void ring_mfma(const float *lds, int buf, int wm, int wn, int i32, int h, f32x16 &acc) {
    const float *As = lds + buf * RB_STAGE, *Bs = As + 2048;
    if (NN) {
        const int m = 32 * wm + i32, f = (m >> 1) & 7;
        f32x4 x[4];
#pragma unroll
        for (int t = 0; t < 4; t++) x[t] = *reinterpret_cast<const f32x4 *>(As + m * 32 + 4 * ((2 * t + h) ^ f));
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float b = Bs[(8 * t + 4 * h + e) * 64 + 32 * wn + i32];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[t][e], b, acc, 0, 0, 0);
            }
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float a = As[(2 * j + h) * 64 + 32 * wm + i32];
            const float b = Bs[(2 * j + h) * 64 + 32 * wn + i32];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
}

// write-through stores / cache-bypassing loads for the partial tiles (the hand-over of gemm_mid_sk_kernel: gemm.hip, sk_store_through)
__device__ __forceinline__ void ring_store_through(float *p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 ring_load_through(const float *p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// epilogue + store of the wave's quarter.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
// Every load and store is a buffer operation whose lanes outside the matrix are aimed out of range (loads give 0, stores are dropped):
// sixteen back-to-back loads per input and sixteen stores, no branch per element (a per-element "load or not" makes the compiler wait
// for every element's round trip on its own).
template <int EPI>
__device__ __forceinline__ void ring_output_epi(const RingProb &P, int m0, int n0, int wm, int wn, int i32, int h, const f32x16 &acc) {
    const int col = n0 + 32 * wn + i32, row0 = m0 + 32 * wm + 4 * h;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(P.C, 0, (int)(4 * ((int64_t)(P.M - 1) * P.ldc + P.N)), 0x00020000);
    int offc[16];
    float ax[16], old[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = row0 + (r & 3) + 8 * (r >> 2);
        offc[r] = (row < P.M && col < P.N) ? 4 * (row * P.ldc + col) : RB_OOB;
    }
    if (EPI != HIDVAE_EPI_NONE) {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.aux), 0,
                                                                            (int)(4 * ((int64_t)(P.M - 1) * P.ldaux + P.N)), 0x00020000);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = row0 + (r & 3) + 8 * (r >> 2);
            ax[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (row < P.M && col < P.N) ? 4 * (row * P.ldaux + col) : RB_OOB, 0, 0));
        }
    }
    if (P.accumulate) {
#pragma unroll
        for (int r = 0; r < 16; r++) old[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, offc[r], 0, 0));
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        float v = EPI == HIDVAE_EPI_NONE ? acc[r] : hv_apply_epilogue(EPI, acc[r], &ax[r], 0, P.scale);
        if (P.accumulate) v = old[r] + v;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, offc[r], 0, 0);
    }
}
__device__ __forceinline__ void ring_output(const RingProb &P, int m0, int n0, int wm, int wn, int i32, int h, const f32x16 &acc) {
    switch (P.epilogue) {  // (uniform)
        case HIDVAE_EPI_DSILU: ring_output_epi<HIDVAE_EPI_DSILU>(P, m0, n0, wm, wn, i32, h, acc); break;
        case HIDVAE_EPI_DRELU: ring_output_epi<HIDVAE_EPI_DRELU>(P, m0, n0, wm, wn, i32, h, acc); break;
        case HIDVAE_EPI_DGELU: ring_output_epi<HIDVAE_EPI_DGELU>(P, m0, n0, wm, wn, i32, h, acc); break;
        case HIDVAE_EPI_DSIGMOID: ring_output_epi<HIDVAE_EPI_DSIGMOID>(P, m0, n0, wm, wn, i32, h, acc); break;
        default: ring_output_epi<HIDVAE_EPI_NONE>(P, m0, n0, wm, wn, i32, h, acc); break;
    }
}

// end of a segment: steps [ks0, ks0 + len) of tile `tile` (global steps from tstart, nT of them) are in `acc`
__device__ __forceinline__ void ring_flush(const RingArgs &a, const RingProb &P, int tile, int counter, int ks0, int len, int tstart, int v,
                                           int wave, int lane, f32x16 &acc, int *flag) {
    const int wm = wave >> 1, wn = wave & 1, i32 = lane & 31, h = lane >> 5;
    int by, bx;
    ring_tile_xy(P, tile, by, bx);
    const int m0 = 64 * by, n0 = 64 * bx, nT = P.nsteps;
    if (ks0 == 0 && len == nT) {  // the whole tile
        ring_output(P, m0, n0, wm, wn, i32, h, acc);
        return;
    }
    // a piece of the tile: park it (slot 2v: the tile began before this range; 2v + 1: it begins here and goes on), count arrivals
    float *mine = a.slabs + ((int64_t)2 * v + (ks0 == 0 ? 1 : 0)) * 4096 + (wave * 4 * 64 + lane) * 4;
#pragma unroll
    for (int gi = 0; gi < 4; gi++) {
        f32x4 w = {acc[4 * gi], acc[4 * gi + 1], acc[4 * gi + 2], acc[4 * gi + 3]};
        ring_store_through(mine + gi * 256, w);
    }
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");  // this thread's piece has been acknowledged by memory (and every DMA has landed)
    __syncthreads();                                     // ... and so has every thread's
    const int v_first = tstart / a.q, v_last = (tstart + nT - 1) / a.q;
    if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(a.counters + counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool last = *flag == v_last - v_first;
    if (last) {  // last to arrive: every piece is in memory; ascending k, whoever finishes
        f32x16 sum;
        for (int u = v_first; u <= v_last; u++) {
            const float *src = a.slabs + ((int64_t)2 * u + (u * a.q <= tstart ? 1 : 0)) * 4096 + (wave * 4 * 64 + lane) * 4;
            f32x4 w[4];
#pragma unroll
            for (int gi = 0; gi < 4; gi++) w[gi] = ring_load_through(src + gi * 256);
#pragma unroll
            for (int gi = 0; gi < 4; gi++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[gi]) : : "memory");  // (ties the uses of w to the wait)
#pragma unroll
            for (int gi = 0; gi < 4; gi++)
#pragma unroll
                for (int e = 0; e < 4; e++) sum[4 * gi + e] = u == v_first ? w[gi][e] : sum[4 * gi + e] + w[gi][e];
        }
        ring_output(P, m0, n0, wm, wn, i32, h, sum);
        if (threadIdx.x == 0) __hip_atomic_store(a.counters + counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // clean for the next launch
    }
    __syncthreads();  // the flag word is free again
}

// bias gradient db[c] = sum_b g[b, c]: 32 columns per unit.  Thread t takes the column quad 4 (t & 7) of the unit and the rows
// t >> 3, t >> 3 + 32, ...: sixteen-byte loads, eight in flight, each thread's rows added in ascending order; the 32 row groups' partials
// are then added in ascending group order through LDS (fixed order: bit-reproducible).  Memory latency, not arithmetic: ~2.5 us per unit
// at 1024 rows, which is what a.cu prices.
__device__ __forceinline__ void ring_colsum32(const RingArgs &a, int c0, float *part) {
    const int cq = threadIdx.x & 7, rg = threadIdx.x >> 3;  // 8 column quads x 32 row groups
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.cs_x), 0,
                                                                        (int)(4 * ((int64_t)(a.cs_rows - 1) * a.cs_ld + a.cs_cols)), 0x00020000);
    const int col = c0 + 4 * cq;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = rg; r0 < a.cs_rows; r0 += 8 * 32) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int r = r0 + 32 * j;
            v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (r < a.cs_rows && col < a.cs_cols) ? 4 * (r * a.cs_ld + col) : RB_OOB, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < 8; j++) acc = acc + v[j];
    }
    *reinterpret_cast<f32x4 *>(part + (rg * 8 + cq) * 4) = acc;
    __syncthreads();
    if (threadIdx.x < 32 && c0 + (int)threadIdx.x < a.cs_cols) {
        float v = part[threadIdx.x];
        for (int j = 1; j < 32; j++) v += part[j * 32 + threadIdx.x];
        float *dst = a.cs_out + c0 + threadIdx.x;
        *dst = a.cs_accumulate ? *dst + v : v;
    }
    __syncthreads();
}

__device__ __forceinline__ int ring_xcd_slot(int slot, int nslots) {  // XCD j works on the j-th contiguous eighth of the step list
    return (nslots & 7) == 0 ? (slot & 7) * (nslots >> 3) + (slot >> 3) : slot;
}

// the steps [ks0, ks0 + len) of one tile: the tight loop.  `s` = the workgroup's step counter (ring position), `nst` its total.
template <bool NN>
__device__ __forceinline__ void ring_segment(const RingArgs &a, Loader &L, int len, int &s, int nst, int wave, int lane, float *lds, f32x16 &acc) {
    const int wm = wave >> 1, wn = wave & 1, i32 = lane & 31, h = lane >> 5;
    for (int i = 0; i < len; i++, s++) {
        // stage s has landed once at most the younger stages' DMAs (4 per stage and wave) are outstanding
        const int rem = nst - 1 - s;
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(8)" : : : "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" : : : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        __builtin_amdgcn_s_barrier();  // every wave's share of stage s is in LDS; everybody is done reading stage s - 1
        if (s + RB_D < nst) loader_issue(a, L, wave, lane, lds, (s + RB_D) % RB_NS);
        ring_mfma<NN>(lds, s % RB_NS, wm, wn, i32, h, acc);
    }
}

__global__ __launch_bounds__(256) void gemm_ring_bwd_kernel(RingArgs a) {
    extern __shared__ __attribute__((aligned(16))) float ring_lds[];
    int *flag = reinterpret_cast<int *>(ring_lds + RB_NS * RB_STAGE);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int v = ring_xcd_slot((int)blockIdx.x, a.G);
    const int lo = v * a.q;
    const int hi = lo + a.q < a.S ? lo + a.q : a.S;
    const int n0 = a.p[0].nsteps, n1 = a.p[1].nsteps, T0 = a.p[0].ntiles;
    const int S0 = T0 * n0, Sg = S0 + a.p[1].ntiles * n1;
    const int ghi = hi < Sg ? hi : Sg;
    const int nst = ghi - lo;  // GEMM steps of this workgroup
    if (nst > 0) {
        Loader L;
        {
            const Cursor c = ring_locate(a, lo);
            L.prob = c.prob; L.tile = c.tile; L.ks = c.ks;
            loader_tile(a, L, wave, lane);
        }
        int prob = L.prob, tile = L.tile, ks = L.ks;  // the consumer's position
#pragma unroll
        for (int i = 0; i < RB_D; i++)
            if (i < nst) loader_issue(a, L, wave, lane, ring_lds, i);
        int s = 0;
        while (s < nst) {
            const int nT = prob ? n1 : n0;
            int len = nT - ks;
            if (len > nst - s) len = nst - s;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.0f;
            if (prob) ring_segment<true>(a, L, len, s, nst, wave, lane, ring_lds, acc);
            else ring_segment<false>(a, L, len, s, nst, wave, lane, ring_lds, acc);
            ring_flush(a, a.p[prob], tile, (prob ? T0 : 0) + tile, ks, len, (prob ? S0 : 0) + tile * nT, v, wave, lane, acc, flag);
            ks += len;
            if (ks == nT) {
                ks = 0;
                if (++tile == (prob ? a.p[1].ntiles : T0)) {
                    tile = 0;
                    prob++;
                }
            }
        }
    }
    // bias column sums: a unit belongs to the range that holds its first step
    if (hi > Sg && a.nbc > 0) {
        __syncthreads();
        int step = lo > Sg ? lo : Sg;
        while (step < hi) {
            const int rel = step - Sg, u = (rel + a.cu - 1) / a.cu;
            if (u >= a.nbc || Sg + u * a.cu >= hi) break;
            ring_colsum32(a, u * 32, ring_lds);
            step = Sg + u * a.cu + 1;
        }
    }
}

}  // namespace

int hv_ring_linear_bwd(const float *g, int64_t ldg, const float *x, int64_t ldx, const float *W, int64_t ldw, int64_t B, int64_t n_out,
                       int64_t n_in, float *dW, int64_t lddw, int accumulate_dw, float *dX, int64_t lddx, int dx_epilogue, float *aux,
                       int64_t ldaux, float dx_scale, float *db, int accumulate_db, float *workspace, int slots, hipStream_t stream) {
    RingArgs a{};
    RingProb &p0 = a.p[0], &p1 = a.p[1];
    auto band = [](int tiles, int nby) {  // band height ~ sqrt(tiles / 8): the eighth of the list an XCD takes is then about square
        int bh = 1;
        while ((bh + 1) * (bh + 1) * 8 <= tiles) bh++;
        return bh < nby ? bh : nby;
    };
    p0.A = g; p0.lda = (int)ldg; p0.B = x; p0.ldb = (int)ldx; p0.C = dW; p0.ldc = (int)lddw;
    p0.M = (int)n_out; p0.N = (int)n_in; p0.K = (int)B;
    p0.nbx = (int)hv_cdiv(n_in, 64); p0.nby = (int)hv_cdiv(n_out, 64); p0.ntiles = p0.nbx * p0.nby; p0.nsteps = (int)hv_cdiv(B, RB_BK);
    p0.bh = band(p0.ntiles, p0.nby);
    p0.epilogue = HIDVAE_EPI_NONE; p0.scale = 1.0f; p0.accumulate = accumulate_dw; p0.aux = nullptr; p0.ldaux = 0;
    p0.a_bytes = (unsigned)(4 * ((B - 1) * ldg + n_out)); p0.b_bytes = (unsigned)(4 * ((B - 1) * ldx + n_in));
    if (dX != nullptr) {
        p1.A = g; p1.lda = (int)ldg; p1.B = W; p1.ldb = (int)ldw; p1.C = dX; p1.ldc = (int)lddx;
        p1.M = (int)B; p1.N = (int)n_in; p1.K = (int)n_out;
        p1.nbx = (int)hv_cdiv(n_in, 64); p1.nby = (int)hv_cdiv(B, 64); p1.ntiles = p1.nbx * p1.nby; p1.nsteps = (int)hv_cdiv(n_out, RB_BK);
        p1.bh = band(p1.ntiles, p1.nby);
        p1.epilogue = dx_epilogue; p1.aux = aux; p1.ldaux = aux ? (int)ldaux : 0; p1.scale = dx_scale; p1.accumulate = 0;
        p1.a_bytes = (unsigned)(4 * ((B - 1) * ldg + n_out)); p1.b_bytes = (unsigned)(4 * ((n_out - 1) * ldw + n_in));
    } else {
        p1.ntiles = 0; p1.nsteps = 1; p1.nbx = p1.nby = p1.bh = 1;
    }
    a.cs_x = g; a.cs_ld = (int)ldg; a.cs_rows = (int)B; a.cs_cols = (int)n_out; a.cs_out = db; a.cs_accumulate = accumulate_db;
    a.nbc = db != nullptr ? (int)hv_cdiv(n_out, 32) : 0;
    a.cu = 4 + (int)hv_cdiv(B, 512);  // a 32-column strip of B rows: latency-bound, ~2 us + 0.5 us per 512 rows, in steps of ~0.45 us
    a.S = p0.ntiles * p0.nsteps + p1.ntiles * p1.nsteps + a.nbc * a.cu;
    if (p0.ntiles + p1.ntiles > HV_SK_COUNTERS) return 1;
    constexpr int minq = 8;  // shortest range worth a workgroup (steps of 32 k)
    int G = a.S / minq;
    if (G > slots) G = slots;
    if (G > HV_SK_MAX_G) G = (int)HV_SK_MAX_G;
    if (G < 1) G = 1;
    a.q = (int)hv_cdiv(a.S, G);
    a.G = (int)hv_cdiv(a.S, a.q);
    a.counters = reinterpret_cast<int *>(workspace);
    a.slabs = workspace + HV_SK_COUNTERS;
    constexpr size_t bytes = (size_t)RB_LDS_FLOATS * 4;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_ring_bwd_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (attr != hipSuccess) return hv_fail(HIDVAE_ELAUNCH, "linear_bwd ring: could not size the LDS of gemm_ring_bwd_kernel");
    hipLaunchKernelGGL(gemm_ring_bwd_kernel, dim3((unsigned)a.G), dim3(256), bytes, stream, a);
    HV_LAUNCH_CHECK("linear_bwd ring");
    return HIDVAE_OK;
}
