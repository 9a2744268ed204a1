// Dispatch rules that more than one translation unit must agree on (host code only).
#pragma once
#include <stdint.h>

// hidvae_linear_bwd on the balanced LDS-shared kernel (gemm_mid_sk_kernel, gemm.hip): which problems take it, and what it needs at
// the head of the caller's workspace -- HV_SK_COUNTERS arrival counters (ints, zero between launches) and two 64x64 partial-tile
// slabs per workgroup.
constexpr int64_t HV_SK_COUNTERS = 4096;
constexpr int64_t HV_SK_MAX_G = 512;
constexpr int64_t HV_SK_WS_FLOATS = HV_SK_COUNTERS + HV_SK_MAX_G * 2 * 4096;

static inline bool hv_lbwd_balanced(int64_t B, int64_t n_out, int64_t n_in, bool has_dx = true) {
    // which Linear backwards take the LDS-shared kernels (gemm_ring.hip, fallback gemm_mid_sk) instead of the per-wave pair kernels.
    // Round 3 drew the line at 0.9 GFLOP from stand-alone timings (B = 1024: 460 x 512 wins, 512 x 256 loses).  Inside the tagged step
    // the per-wave kernels -- every wave streams its own operands from L2 -- run 1.5-2x slower than alone while three lanes share the
    // chip (1024 x 230 x 256: 11 us alone, 14-25 us in the step; 1024 x 348 x 345: 47 us), the LDS-shared ones ~8 %.  Measured in the
    // step with the ring kernel (profiles/r04_ring_threshold_ab.log): threshold 900 / 300 / 200 / 100 / 10 MFLOP -> tagged B = 1024
    // 1.156 / 1.126 / 1.088 / 1.093 / 1.093 ms, untagged 0.234 / 0.231 / 0.231 / 0.239 / 0.237 ms.
    const int64_t t64 = ((n_out + 63) / 64) * ((n_in + 63) / 64), u64 = ((B + 63) / 64) * ((n_in + 63) / 64);
    const double mflop = 2.0e-6 * (double)B * (double)n_out * (double)n_in * (has_dx ? 2.0 : 1.0);
    constexpr double min_mflop = 200.0;
    constexpr int64_t max_b = 16384;
    return n_out >= 64 && n_in >= 64 && B >= 256 && B <= max_b && t64 + u64 <= HV_SK_COUNTERS && mflop >= min_mflop;
}
