#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
extern "C" __global__ void k(const bf16x8_t *a, const bf16x8_t *b, float *out) {
    f32x16 acc;
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    float tm = __builtin_fminf(__builtin_fminf(acc[0], acc[1]), acc[2]);
    tm = __builtin_fminf(__builtin_fminf(tm, acc[3]), acc[4]);
    tm = __builtin_fminf(__builtin_fminf(tm, acc[5]), acc[6]);
    out[threadIdx.x] = tm;
}
