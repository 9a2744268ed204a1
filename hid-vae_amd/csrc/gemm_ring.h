// Linear backward on the LDS-DMA ring kernel (gemm_ring.hip), called by hidvae_linear_bwd's dispatch in gemm.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// dW [n_out, n_in] (+)= g^T x;  dX [B, n_in] = epi(g W) (dX == nullptr: skipped);  db [n_out] (+)= column sums of g (db == nullptr: skipped).
// workspace: the HIDVAE_WS_LINEAR_BWD buffer of a balanced-kernel shape (rules.h): HV_SK_COUNTERS zeroed arrival counters, then the
// partial-tile slabs.  slots: workgroups the launch may use (256 = one per CU, 512 = two).  Byte offsets are 32-bit: the caller checks
// rows * ld < 2^29 for every operand.  -> HIDVAE_OK, a negative error code, or 1 when the shape is not taken.
int hv_ring_linear_bwd(const float *g, int64_t ldg, const float *x, int64_t ldx, const float *W, int64_t ldw, int64_t B, int64_t n_out,
                       int64_t n_in, float *dW, int64_t lddw, int accumulate_dw, float *dX, int64_t lddx, int dx_epilogue, float *aux,
                       int64_t ldaux, float dx_scale, float *db, int accumulate_db, float *workspace, int slots, hipStream_t stream);
