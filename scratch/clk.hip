#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void chain(float* out, int n, unsigned long long* stamps) {
  f32x16 acc = {0};
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; i++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 64 + threadIdx.x] = acc[0];
  if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}
__global__ void empty() {}
int main() {
  float* out; unsigned long long* st; hipMalloc(&out, 1 << 22); hipMalloc(&st, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int n : {100, 1000, 10000, 100000}) {
    for (int grid : {1, 1024}) {
      chain<<<grid, 64>>>(out, n, st); hipDeviceSynchronize();
      hipEventRecord(e0); chain<<<grid, 64>>>(out, n, st); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
      printf("n=%6d grid=%4d: %.1f us wall, %.2f cyc/mfma (memtime), clock = %.0f MHz (memtime/realtime*100)\n", n, grid, ms * 1e3,
             (double)h[0] / n, (double)h[0] / (double)h[1] * 100.0);
    }
  }
  // empty-kernel launch cadence
  for (int i = 0; i < 10; i++) empty<<<1, 64>>>();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 1000; i++) empty<<<1, 64>>>();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("1000 empty kernels back-to-back (eager): %.2f us each\n", ms);
  // graph of 100 empty kernels
  hipStream_t s; hipStreamCreate(&s);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 100; i++) empty<<<1, 64, 0, s>>>();
  hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 3; i++) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int i = 0; i < 20; i++) hipGraphLaunch(ge, s);
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("graph of 100 empty kernels: %.2f us per kernel\n", ms * 1e3 / 2000);
  return 0;
}
