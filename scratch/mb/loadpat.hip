// microbenchmark: how fast can waves pull operand strips out of L2 with the access patterns of the GEMM kernels?
//   pattern 0: "direct" -- lane (row = lane&31, h = lane>>5) reads 16 B at row*ld + (k0 + 4h)*4 and 16 B at +32 B, k0 += 16 per step
//   pattern 1: coalesced  -- the wave reads 1 KB contiguous per instruction (4 rows x 64 floats), walking the same strip
// Every wave owns a 32-row x K strip; strips are re-read by `reuse` different waves (as column tiles re-read A rows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int PAT, int NS>
__global__ __launch_bounds__(256) void k(const float* A, int rows, int K, int reuse, float* out) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    const int strips = rows / 32;
    const int strip = wave % strips;
    const float* base = A + (size_t)strip * 32 * K;
    f4 acc = {0, 0, 0, 0};
    if (PAT == 0) {
        const int r = lane & 31, h = lane >> 5;
        const float* p = base + (size_t)r * K + 4 * h;
        f4 ring[NS][2];
        #pragma unroll
        for (int s = 0; s < NS - 1; s++) { ring[s][0] = *(const f4*)(p + 16 * s); ring[s][1] = *(const f4*)(p + 16 * s + 8); }
        for (int k0 = 0; k0 < K; k0 += 16 * NS) {
            #pragma unroll
            for (int s = 0; s < NS; s++) {
                const int kn = k0 + 16 * (s + NS - 1);
                const int slot = (s + NS - 1) % NS;
                if (kn < K) { ring[slot][0] = *(const f4*)(p + kn); ring[slot][1] = *(const f4*)(p + kn + 8); }
                acc += ring[s][0] * ring[s][1];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (PAT == 2) {
        // row-contiguous operand (A of TN): element (k, m) at A[k*ld + m]; lane (i = lane&31, h) reads 8 scalars per 16-block:
        // k = k0 + (j>>1) + 8 (j&1) + 4h -- 32 consecutive floats (128 B) per half-wave and instruction
        const int i = lane & 31, h = lane >> 5;
        const int ld = rows;  // the strip is 32 columns of a [K, rows] matrix
        const float* p = A + (size_t)(4 * h) * ld + strip * 32 + i;
        float s0 = 0.f;
        for (int k0 = 0; k0 < K; k0 += 16) {
            float v[8];
            #pragma unroll
            for (int j = 0; j < 8; j++) v[j] = p[(size_t)(k0 + (j >> 1) + 8 * (j & 1)) * ld];
            #pragma unroll
            for (int j = 0; j < 8; j++) s0 += v[j];
        }
        acc[0] = s0;
    } else if (PAT == 3) {
        // the same strip as 16-byte loads along m: lane -> k = lane>>3 (+8 per instruction), 4 floats at (lane&7)*4: 128 B per k-row
        const int kk = lane >> 3, c = (lane & 7) * 4;
        const int ld = rows;
        const float* p = A + (size_t)kk * ld + strip * 32 + c;
        for (int k0 = 0; k0 < K; k0 += 32) {
            f4 v[4];
            #pragma unroll
            for (int j = 0; j < 4; j++) v[j] = *(const f4*)(p + (size_t)(k0 + 8 * j) * ld);
            #pragma unroll
            for (int j = 0; j < 4; j++) acc += v[j];
        }
    } else {
        // 32 rows x K: per instruction 4 rows x 64 floats (K >= 64): lane -> row = lane>>4, 4 floats at (lane&15)*4
        const int rr = lane >> 4, c = (lane & 15) * 4;
        for (int k0 = 0; k0 < K; k0 += 64) {
            f4 v[8];
            #pragma unroll
            for (int j = 0; j < 8; j++) v[j] = *(const f4*)(base + (size_t)(4 * j + rr) * K + k0 + c);
            #pragma unroll
            for (int j = 0; j < 8; j++) acc += v[j];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.0f;
}
int main() {
    const int rows = 1024, Ks[] = {512, 768};
    for (int K : Ks) {
        float* A; float* out;
        hipMalloc(&A, (size_t)rows * K * 4); hipMalloc(&out, 4);
        hipMemset(A, 0, (size_t)rows * K * 4);
        for (int reuse : {16, 48}) {
            const int waves = (rows / 32) * reuse, blocks = waves / 4;
            for (int pat = 0; pat < 5; pat++) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                auto launch = [&]() {
                    if (pat == 0) hipLaunchKernelGGL((k<0, 3>), dim3(blocks), dim3(256), 0, 0, A, rows, K, reuse, out);
                    else if (pat == 1) hipLaunchKernelGGL((k<0, 6>), dim3(blocks), dim3(256), 0, 0, A, rows, K, reuse, out);
                    else if (pat == 2) hipLaunchKernelGGL((k<1, 1>), dim3(blocks), dim3(256), 0, 0, A, rows, K, reuse, out);
                    else if (pat == 3) hipLaunchKernelGGL((k<2, 1>), dim3(blocks), dim3(256), 0, 0, A, rows, K, reuse, out);
                    else hipLaunchKernelGGL((k<3, 1>), dim3(blocks), dim3(256), 0, 0, A, rows, K, reuse, out);
                };
                for (int i = 0; i < 5; i++) launch();
                hipEventRecord(e0);
                for (int i = 0; i < 50; i++) launch();
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double us = ms * 1e3 / 50, bytes = (double)waves * 32 * K * 4;
                printf("K=%d reuse=%d waves=%d pattern=%s: %.1f us, %.2f TB/s (bytes loaded by waves)\n", K, reuse, waves,
                       pat == 0 ? "direct NS3" : pat == 1 ? "direct NS6" : pat == 2 ? "coalesced" : pat == 3 ? "row-contig b32" : "row-contig b128", us, bytes / us * 1e-6);
            }
        }
        hipFree(A); hipFree(out);
    }
    return 0;
}
