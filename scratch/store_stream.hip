// Micro-model of the VQ launch's memory stream: per 32-item wave tile, load 128 B/item, then write 5 pieces of 128 B/item
// (z, emb_cat x3 at 384-B row stride, emb_sum).  Variants: where the next tile's load is issued, and how a store instruction's
// 1 KB is spread.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scratch/_store_stream.so scratch/store_stream.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int VAR>
__global__ __launch_bounds__(1024) void stream_kernel(const float *y, int64_t B, float *z, float *cat, float *esum, int spin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int64_t ntiles = (B + 511) / 512;
    float4 r[4];
    int64_t tile = blockIdx.x;
    auto load_tile = [&](int64_t t) {
        const int64_t item = t * 512 + wave * 32 + n;
        const int64_t src = item < B ? item : B - 1;
        if (VAR == 2) {  // 1 KB contiguous per instruction: lane l reads float4 number l of 8 consecutive rows
            const int64_t w0 = t * 512 + wave * 32;
#pragma unroll
            for (int k = 0; k < 4; k++) r[k] = *reinterpret_cast<const float4 *>(y + (w0 + 8 * k) * 32 + 4 * lane);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) r[k] = *reinterpret_cast<const float4 *>(y + src * 32 + 16 * h + 4 * k);
        }
    };
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const int64_t w0 = tile * 512 + wave * 32;
        const int64_t item = w0 + n;
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = r[k];
        for (int s = 0; s < spin; s++) {  // stand-in for the per-tile arithmetic (dependent chain)
#pragma unroll
            for (int k = 0; k < 4; k++) { v[k].x = fmaf(v[k].x, 1.0001f, 0.5f); v[k].y = fmaf(v[k].y, v[k].x, 0.25f); }
        }
        const int64_t nxt = tile + gridDim.x;
        if (VAR == 1 && nxt < ntiles) load_tile(nxt);  // prefetch BEFORE this tile's stores: its wait will not cover them
        if (VAR == 2) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                *reinterpret_cast<float4 *>(z + (w0 + 8 * k) * 32 + 4 * lane) = v[k];
                *reinterpret_cast<float4 *>(esum + (w0 + 8 * k) * 32 + 4 * lane) = v[k];
            }
#pragma unroll
            for (int lv = 0; lv < 3; lv++)
#pragma unroll
                for (int k = 0; k < 4; k++)  // 8 rows x 128 B at 384-B stride per instruction
                    *reinterpret_cast<float4 *>(cat + (w0 + 8 * k + (lane >> 3)) * 96 + lv * 32 + 4 * (lane & 7)) = v[k];
        } else if (item < B) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                *reinterpret_cast<float4 *>(z + item * 32 + 16 * h + 4 * k) = v[k];
                *reinterpret_cast<float4 *>(esum + item * 32 + 16 * h + 4 * k) = v[k];
            }
#pragma unroll
            for (int lv = 0; lv < 3; lv++)
#pragma unroll
                for (int k = 0; k < 4; k++) *reinterpret_cast<float4 *>(cat + item * 96 + lv * 32 + 16 * h + 4 * k) = v[k];
        }
        if (VAR != 1 && nxt < ntiles) load_tile(nxt);
    }
}

extern "C" int run_stream(int var, const float *y, int64_t B, float *z, float *cat, float *esum, int spin, int grid, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (var == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(grid), dim3(1024), 0, s, y, B, z, cat, esum, spin);
    else if (var == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(grid), dim3(1024), 0, s, y, B, z, cat, esum, spin);
    else hipLaunchKernelGGL(stream_kernel<2>, dim3(grid), dim3(1024), 0, s, y, B, z, cat, esum, spin);
    return (int)hipGetLastError();
}
