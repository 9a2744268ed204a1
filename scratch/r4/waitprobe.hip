#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDSP(p) ((__attribute__((address_space(3))) void *)(p))
extern "C" __global__ __launch_bounds__(256) void waitprobe(const float *src, float *dst, int nbytes, int steps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, nbytes, 0x00020000);
    const int voff = (wave * 64 + lane) * 16;
    f32x16 acc;
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    // prologue: stages 0,1,2
    for (int s = 0; s < 3; s++) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(lds + 4096 * s + 256 * wave), 16, voff, s * 4096, 0, 0);
    for (int s = 0; s < steps; s++) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(lds + 4096 * ((s + 3) & 3) + 256 * wave), 16, voff, (s + 3) * 4096, 0, 0);
        const float *As = lds + 4096 * (s & 3);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float a = As[j * 64 + (lane & 31)], b = As[2048 + j * 64 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    for (int i = 0; i < 16; i++) dst[(threadIdx.x * 16 + i)] = acc[i];
}
