// Dispatch rules that more than one translation unit must agree on (host code only).
#pragma once
#include <stdint.h>
#include <stdlib.h>

// hidvae_linear_bwd on the balanced LDS-shared kernel (gemm_mid_sk_kernel, gemm.hip): which problems take it, and what it needs at
// the head of the caller's workspace -- HV_SK_COUNTERS arrival counters (ints, zero between launches) and two 64x64 partial-tile
// slabs per workgroup.
constexpr int64_t HV_SK_COUNTERS = 4096;
constexpr int64_t HV_SK_MAX_G = 512;
constexpr int64_t HV_SK_WS_FLOATS = HV_SK_COUNTERS + HV_SK_MAX_G * 2 * 4096;

static inline bool hv_lbwd_balanced(int64_t B, int64_t n_out, int64_t n_in, bool has_dx = true) {
    // enough arithmetic to occupy the chip with 64x64 tiles (below, the per-wave pair kernels spread a small problem better):
    // measured crossover ~0.9 GFLOP (B = 1024: 460 x 512 wins, 512 x 256 loses; B = 2048: 512 x 256 wins, 256 x 128 loses)
    const int64_t t64 = ((n_out + 63) / 64) * ((n_in + 63) / 64), u64 = ((B + 63) / 64) * ((n_in + 63) / 64);
    const double mflop = 2.0e-6 * (double)B * (double)n_out * (double)n_in * (has_dx ? 2.0 : 1.0);
    constexpr double min_mflop = 900.0;
    constexpr int64_t max_b = 16384;
    return n_out >= 64 && n_in >= 64 && B >= 256 && B <= max_b && t64 + u64 <= HV_SK_COUNTERS && mflop >= min_mflop;
}
