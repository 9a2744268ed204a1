// Padded -> jagged row copy and its adjoint (reference ops/triton/jagged.py:9-124, the stage-2 transformer's only custom kernel;
// a Triton kernel there, replaced by HIP here).  x [B, N, D] (any element type, D contiguous) with lengths[b] valid rows per
// batch entry  ->  values [sum(lengths), D] = the valid rows back to back; offsets[b] = first row of entry b (exclusive scan).
// Pure HBM traffic: every valid row is read once and written once as 16-byte vectors (or bytes when a row is not 16-B granular).
#include "common.h"

namespace {

// one wave per row of (b, n); rows past lengths[b] are skipped (forward) or zero-filled (backward)
template <bool TO_JAGGED, bool VEC16>
__global__ __launch_bounds__(256) void jagged_rows_kernel(char *padded, int64_t stride_b, int64_t stride_n, const int64_t *offsets,
                                                          char *values, int64_t B, int64_t N, int64_t row_bytes) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * N) return;
    const int64_t b = row / N, n = row - b * N;
    const int64_t start = offsets[b], len = offsets[b + 1] - start;
    char *p = padded + b * stride_b + n * stride_n;
    if (n >= len) {
        if (!TO_JAGGED) {  // gradient of a padding row is zero
            if (VEC16) for (int64_t i = lane; i < row_bytes / 16; i += 64) reinterpret_cast<uint4 *>(p)[i] = make_uint4(0, 0, 0, 0);
            else for (int64_t i = lane; i < row_bytes; i += 64) p[i] = 0;
        }
        return;
    }
    char *v = values + (start + n) * row_bytes;
    if (VEC16) {
        for (int64_t i = lane; i < row_bytes / 16; i += 64) {
            if (TO_JAGGED) reinterpret_cast<uint4 *>(v)[i] = reinterpret_cast<const uint4 *>(p)[i];
            else reinterpret_cast<uint4 *>(p)[i] = reinterpret_cast<const uint4 *>(v)[i];
        }
    } else {
        for (int64_t i = lane; i < row_bytes; i += 64) {
            if (TO_JAGGED) v[i] = p[i];
            else p[i] = v[i];
        }
    }
}

template <bool TO_JAGGED>
int launch(void *padded, int64_t stride_b, int64_t stride_n, const int64_t *offsets, void *values, int64_t B, int64_t N, int64_t row_bytes,
           hipStream_t s) {
    const bool vec = row_bytes % 16 == 0 && stride_b % 16 == 0 && stride_n % 16 == 0 &&
                     ((reinterpret_cast<uintptr_t>(padded) | reinterpret_cast<uintptr_t>(values)) & 15) == 0;
    const unsigned grid = (unsigned)hv_cdiv(B * N, 4);
    if (vec)
        hipLaunchKernelGGL((jagged_rows_kernel<TO_JAGGED, true>), dim3(grid), dim3(256), 0, s, (char *)padded, stride_b, stride_n, offsets,
                           (char *)values, B, N, row_bytes);
    else
        hipLaunchKernelGGL((jagged_rows_kernel<TO_JAGGED, false>), dim3(grid), dim3(256), 0, s, (char *)padded, stride_b, stride_n, offsets,
                           (char *)values, B, N, row_bytes);
    return 0;
}

}  // namespace

extern "C" int hidvae_padded_to_jagged(const void *x, int64_t stride_b_bytes, int64_t stride_n_bytes, const int64_t *offsets, void *values,
                                       int64_t B, int64_t N, int64_t row_bytes, void *stream) {
    HV_REQUIRE(x && offsets && B >= 1 && N >= 1 && row_bytes >= 1 && stride_n_bytes >= row_bytes && stride_b_bytes >= N * stride_n_bytes,
               "padded_to_jagged: bad arguments");
    launch<true>(const_cast<void *>(x), stride_b_bytes, stride_n_bytes, offsets, values, B, N, row_bytes, (hipStream_t)stream);
    HV_LAUNCH_CHECK("padded_to_jagged");
    return HIDVAE_OK;
}

extern "C" int hidvae_jagged_to_padded(const void *values, const int64_t *offsets, void *x, int64_t stride_b_bytes, int64_t stride_n_bytes,
                                       int64_t B, int64_t N, int64_t row_bytes, void *stream) {
    HV_REQUIRE(x && offsets && B >= 1 && N >= 1 && row_bytes >= 1 && stride_n_bytes >= row_bytes && stride_b_bytes >= N * stride_n_bytes,
               "jagged_to_padded: bad arguments");
    launch<false>(x, stride_b_bytes, stride_n_bytes, offsets, const_cast<void *>(values), B, N, row_bytes, (hipStream_t)stream);
    HV_LAUNCH_CHECK("jagged_to_padded");
    return HIDVAE_OK;
}
