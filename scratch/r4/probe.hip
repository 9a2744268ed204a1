// LDS-DMA behaviour probe (gfx950): unaligned 16-byte sources, out-of-range lanes, destination addressing
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDSP(p) ((__attribute__((address_space(3))) void *)(p))

__global__ void probe(const float *src, float *dst, int nbytes, int oob_from, int shift) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = -7.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, nbytes, 0x00020000);
    int voff = (wave * 64 + lane) * 16 + shift;
    if (lane >= oob_from) voff = 0x7FFFFFF0;  // out of range
    // wave w writes its 1 KiB at lds + 1 KiB * w  (wave-uniform base; the hardware adds lane * 16)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(lds + 256 * wave), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) dst[i] = lds[i];
}

int main() {
    const int N = 4096;
    std::vector<float> h(N);
    for (int i = 0; i < N; i++) h[i] = (float)i;
    float *src, *dst;
    hipMalloc(&src, N * 4); hipMalloc(&dst, 2048 * 4);
    hipMemcpy(src, h.data(), N * 4, hipMemcpyHostToDevice);
    std::vector<float> o(2048);
    struct { int nbytes, oob_from, shift; const char *what; } cases[] = {
        {N * 4, 64, 0, "aligned, all lanes in range"},
        {N * 4, 64, 4, "source shifted by 4 bytes (dword-aligned 16-byte loads)"},
        {N * 4, 64, 12, "source shifted by 12 bytes"},
        {N * 4, 40, 0, "lanes >= 40 aimed out of range"},
        {1024 + 8, 64, 0, "buffer ends 8 bytes into wave 1 lane 0's chunk (1032 bytes): per-dword or whole-access range check?"},
    };
    for (auto &c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(128), 8192, 0, src, dst, c.nbytes, c.oob_from, c.shift);
        hipMemcpy(o.data(), dst, 2048 * 4, hipMemcpyDeviceToHost);
        printf("== %s\n", c.what);
        int bad = 0, zeros = 0, untouched = 0;
        for (int w = 0; w < 2; w++)
            for (int l = 0; l < 64; l++)
                for (int e = 0; e < 4; e++) {
                    const float got = o[256 * w + 4 * l + e];
                    const int want_idx = (w * 64 + l) * 4 + c.shift / 4 + e;
                    const bool inr = l < c.oob_from && (want_idx * 4 + 4 <= c.nbytes) && (c.nbytes >= 4096 * 4 || c.shift == 0) ;
                    if (inr) { if (got != (float)want_idx) bad++; }
                    else { if (got == 0.0f) zeros++; else if (got == -7.0f) untouched++; else bad++; }
                }
        printf("   in-range mismatches %d; out-of-range elements: %d zero, %d untouched(-7)\n", bad, zeros, untouched);
        printf("   wave0 lane0: %g %g %g %g | wave1 lane0: %g %g %g %g | wave1 lane2: %g %g %g %g | beyond (lds[512..515]): %g %g\n", o[0], o[1], o[2], o[3],
               o[256], o[257], o[258], o[259], o[264], o[265], o[266], o[267], o[512], o[513]);
    }
    return 0;
}
