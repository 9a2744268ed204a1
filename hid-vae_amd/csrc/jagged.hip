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

// ---- a batch gathered from the resident item tables in ONE launch (reference data/tags_processed.py:112-150: __getitem__ indexes
// item_data, tags_emb and tags_indices with the batch's ids; train_hidvae.py:698-701 hands the result to the step).  Up to
// HIDVAE_GATHER_MAX tables of any element type share the index vector: destination row r of table t = source row idx[r] of table t.
// A workgroup takes one destination row of every table, dwords or 16-byte vectors; pure HBM traffic (12.3 KB per tagged item).
namespace {

struct GatherArgs {
    const char *src[HIDVAE_GATHER_MAX];
    char *dst[HIDVAE_GATHER_MAX];
    int64_t row_bytes[HIDVAE_GATHER_MAX], src_rows[HIDVAE_GATHER_MAX];
    int vec16[HIDVAE_GATHER_MAX];
    int n;
    const int64_t *idx;
    int64_t rows;
};

__global__ __launch_bounds__(256) void gather_rows_kernel(GatherArgs a) {
    const int64_t r = blockIdx.x;
    const int64_t i = a.idx[r];
#pragma unroll
    for (int t = 0; t < HIDVAE_GATHER_MAX; t++) {
        if (t >= a.n) break;
        if (i < 0 || i >= a.src_rows[t]) continue;  // (the caller's contract: ids in range; an id outside leaves the row as it was)
        const char *s = a.src[t] + i * a.row_bytes[t];
        char *d = a.dst[t] + r * a.row_bytes[t];
        if (a.vec16[t]) {
            for (int64_t k = threadIdx.x; k < a.row_bytes[t] / 16; k += 256) reinterpret_cast<uint4 *>(d)[k] = reinterpret_cast<const uint4 *>(s)[k];
        } else {
            for (int64_t k = threadIdx.x; k < a.row_bytes[t] / 4; k += 256) reinterpret_cast<unsigned *>(d)[k] = reinterpret_cast<const unsigned *>(s)[k];
        }
    }
}

}  // namespace

extern "C" int hidvae_gather_rows(const int64_t *idx, int64_t rows, int n_tables, const void *const *src, void *const *dst,
                                  const int64_t *row_bytes, const int64_t *src_rows, void *stream) {
    HV_REQUIRE(idx && src && dst && row_bytes && src_rows && rows >= 1, "gather_rows: bad arguments");
    HV_REQUIRE(n_tables >= 1 && n_tables <= HIDVAE_GATHER_MAX, "gather_rows: %d tables (1 .. %d)", n_tables, HIDVAE_GATHER_MAX);
    HV_REQUIRE(rows < (int64_t(1) << 31), "gather_rows: %lld rows", (long long)rows);
    GatherArgs a{};
    a.n = n_tables; a.idx = idx; a.rows = rows;
    for (int t = 0; t < n_tables; t++) {
        HV_REQUIRE(src[t] && dst[t] && row_bytes[t] >= 4 && row_bytes[t] % 4 == 0 && src_rows[t] >= 1,
                   "gather_rows: table %d (rows of whole dwords, non-empty)", t);
        HV_REQUIRE(((reinterpret_cast<uintptr_t>(src[t]) | reinterpret_cast<uintptr_t>(dst[t])) & 3) == 0, "gather_rows: table %d is not dword-aligned", t);
        a.src[t] = static_cast<const char *>(src[t]); a.dst[t] = static_cast<char *>(dst[t]);
        a.row_bytes[t] = row_bytes[t]; a.src_rows[t] = src_rows[t];
        a.vec16[t] = row_bytes[t] % 16 == 0 && ((reinterpret_cast<uintptr_t>(src[t]) | reinterpret_cast<uintptr_t>(dst[t])) & 15) == 0;
    }
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, a);
    HV_LAUNCH_CHECK("gather_rows");
    return HIDVAE_OK;
}
