// Micro-benchmark: sustained rate of v_mfma_f32_16x16x32_bf16 against v_mfma_f32_32x32x16_bf16 on gfx950, register
// operands only (no memory), 1 or 2 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(512) void k16(float* out, int iters, float seed) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 0.001f + i); b[i] = (__bf16)(seed * 0.5f + i); }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}
template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, int iters, float seed) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 0.001f + i); b[i] = (__bf16)(seed * 0.5f + i); }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;
}
template <typename F>
static void run(const char* name, F launch, double flop_per_thread_block) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %8.3f ms  %8.1f TFLOP/s\n", name, ms / 5, flop_per_thread_block * 5 / (ms * 1e-3) / 1e12);
}
int main() {
  float* out;
  hipMalloc(&out, 4);
  const int iters = 20000, grid = 256 * 4;
  for (int threads : {256, 512}) {
    const double waves = (double)grid * threads / 64;
    char nm[96];
    snprintf(nm, 96, "16x16x32 bf16, 8 acc, %d thr/WG", threads);
    run(nm, [&] { hipLaunchKernelGGL(k16<8>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.f); }, waves * iters * 8 * 16384.0);
    snprintf(nm, 96, "16x16x32 bf16, 4 acc, %d thr/WG", threads);
    run(nm, [&] { hipLaunchKernelGGL(k16<4>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.f); }, waves * iters * 4 * 16384.0);
    snprintf(nm, 96, "32x32x16 bf16, 4 acc, %d thr/WG", threads);
    run(nm, [&] { hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.f); }, waves * iters * 4 * 32768.0);
    snprintf(nm, 96, "32x32x16 bf16, 2 acc, %d thr/WG", threads);
    run(nm, [&] { hipLaunchKernelGGL(k32<2>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.f); }, waves * iters * 2 * 32768.0);
  }
  return 0;
}
