// Micro-benchmark: MFMA rate when the operands come from LDS.  One 8-wave workgroup per CU (100 KB of LDS), every wave
// loops over steps of R ds_read_b128 fragment reads (conflict-free: lane * 16 B) feeding M v_mfma_f32_16x16x32_bf16 on
// independent accumulators, double buffered (the reads of step s + 1 are issued before the MFMAs of step s).
//   (R, M) = (6, 8)  the 64-channel persistent kernel (2 pixel rows x 64 channels per wave)
//   (12, 32)         the tall kernel (4 x 8 tiles per wave), here with 4 waves
//   (8, 16)          a 4 x 4 tile mapping
// hipcc --offload-arch=gfx950 -O3 lds_mfma.hip -o lds_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NA, int NB, int BOUND = 512>   // NA + NB reads, NA * NB MFMAs per step
__global__ __launch_bounds__(BOUND) void k(float* out, int iters) {
  extern __shared__ char smem[];
  for (int i = threadIdx.x; i < 100 * 1024 / 4; i += blockDim.x) ((float*)smem)[i] = (float)(i & 255) * 0.001f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = smem + wave * 8192 + lane * 16;
  f32x4 acc[NA][NB];
  for (int a = 0; a < NA; ++a) for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 fa0[NA], fb0[NB], fa1[NA], fb1[NB];
  auto load = [&](bf16x8 (&fa)[NA], bf16x8 (&fb)[NB], int s) {
    const char* p = base + (s & 3) * 1024;
#pragma unroll
    for (int a = 0; a < NA; ++a) fa[a] = *(const bf16x8*)(p + a * 16384 % 65536);
#pragma unroll
    for (int b = 0; b < NB; ++b) fb[b] = *(const bf16x8*)(p + 4096 + b * 16384 % 65536);
  };
  load(fa0, fb0, 0);
  for (int it = 0; it < iters; it += 2) {
    load(fa1, fb1, it + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa0[a], fb0[b], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    load(fa0, fb0, it + 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa1[a], fb1[b], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int a = 0; a < NA; ++a) for (int b = 0; b < NB; ++b) s += acc[a][b][0] + acc[a][b][3];
  if (s == 12345.678f) out[0] = s;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
// the same loop on v_mfma_f32_32x32x16_bf16: NA + NB reads (each a 32 x 16 operand, 16 bytes per lane), NA * NB MFMAs of 32 clk
template <int NA, int NB>
__global__ __launch_bounds__(512) void k32(float* out, int iters) {
  extern __shared__ char smem[];
  for (int i = threadIdx.x; i < 100 * 1024 / 4; i += blockDim.x) ((float*)smem)[i] = (float)(i & 255) * 0.001f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = smem + wave * 8192 + lane * 16;
  f32x16 acc[NA][NB];
  for (int a = 0; a < NA; ++a) for (int b = 0; b < NB; ++b) for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;
  bf16x8 fa0[NA], fb0[NB], fa1[NA], fb1[NB];
  auto load = [&](bf16x8 (&fa)[NA], bf16x8 (&fb)[NB], int s) {
    const char* p = base + (s & 3) * 1024;
#pragma unroll
    for (int a = 0; a < NA; ++a) fa[a] = *(const bf16x8*)(p + a * 16384 % 65536);
#pragma unroll
    for (int b = 0; b < NB; ++b) fb[b] = *(const bf16x8*)(p + 4096 + b * 16384 % 65536);
  };
  load(fa0, fb0, 0);
  for (int it = 0; it < iters; it += 2) {
    load(fa1, fb1, it + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[a], fb0[b], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    load(fa0, fb0, it + 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[a], fb1[b], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int a = 0; a < NA; ++a) for (int b = 0; b < NB; ++b) s += acc[a][b][0] + acc[a][b][15];
  if (s == 12345.678f) out[0] = s;
}

template <typename F>
static void run(const char* name, F launch, double flop) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) launch();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %8.3f ms  %8.1f TFLOP/s\n", name, ms / 5, flop * 5 / (ms * 1e-3) / 1e12);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 4);
  const int iters = 4000, grid = 256;
  const size_t lds = 100 * 1024;
#define CASE(NA, NB, THREADS)                                                                                   \
  {                                                                                                              \
    (void)hipFuncSetAttribute((const void*)k<NA, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
    char nm[96];                                                                                                 \
    snprintf(nm, 96, "%d+%d reads, %d MFMAs per step, %d waves/WG", NA, NB, NA* NB, THREADS / 64);                \
    run(nm, [&] { hipLaunchKernelGGL((k<NA, NB>), dim3(grid), dim3(THREADS), lds, 0, out, iters); },            \
        (double)grid * (THREADS / 64) * iters * NA * NB * 16384.0);                                             \
  }
  CASE(4, 2, 512)
  CASE(4, 4, 512)
  CASE(4, 4, 256)
  CASE(8, 4, 256)
  CASE(8, 4, 512)
  CASE(2, 2, 512)
  {   // 8 x 8 tile: 256 accumulator registers, one wave per SIMD (512-register budget)
    (void)hipFuncSetAttribute((const void*)k<8, 8, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    run("8+8 reads, 64 MFMAs per step, 4 waves/WG (1 per SIMD)", [&] { hipLaunchKernelGGL((k<8, 8, 256>), dim3(grid), dim3(256), lds, 0, out, iters); },
        (double)grid * 4 * iters * 64 * 16384.0);
    (void)hipFuncSetAttribute((const void*)k<8, 6, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    run("8+6 reads, 48 MFMAs per step, 4 waves/WG (1 per SIMD)", [&] { hipLaunchKernelGGL((k<8, 6, 256>), dim3(grid), dim3(256), lds, 0, out, iters); },
        (double)grid * 4 * iters * 48 * 16384.0);
  }
#define CASE32(NA, NB, THREADS)                                                                                 \
  {                                                                                                              \
    (void)hipFuncSetAttribute((const void*)k32<NA, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
    char nm[96];                                                                                                 \
    snprintf(nm, 96, "32x32x16: %d+%d reads, %d MFMAs per step, %d waves/WG", NA, NB, NA* NB, THREADS / 64);      \
    run(nm, [&] { hipLaunchKernelGGL((k32<NA, NB>), dim3(grid), dim3(THREADS), lds, 0, out, iters); },          \
        (double)grid * (THREADS / 64) * iters * NA * NB * 32768.0);                                             \
  }
  CASE32(2, 1, 512)   // 64 cout x 32 px per wave: the 64-channel kernel's wave tile
  CASE32(2, 2, 512)   // 64 x 64
  CASE32(4, 2, 256)   // 128 cout x 64 px: the tall kernel's wave tile
  CASE32(4, 2, 512)
  CASE32(2, 2, 256)
  return 0;
}
