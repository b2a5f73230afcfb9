// Micro-benchmark, round 3: LDS-fed v_mfma_f32_32x32x16_bf16 with LARGE wave tiles (the rows profiles/r02_ubench_lds_mfma.txt
// did not have).  One workgroup per CU (100 KB of LDS), every wave loops over K = 16 steps of NA + NB conflict-free
// ds_read_b128 fragment reads (lane * 16 B) feeding NA x NB MFMAs on independent accumulators.
//   form B ("burst"):       the reads of step s + 1 are issued in one block before the MFMAs of step s (r02's form)
//   form I ("interleaved"): the reads of step s + 1 are spread between the MFMAs of step s (sched_group_barrier:
//                           1 ds_read per MPR MFMAs), the way a hand-placed one-wave-per-SIMD stream would do it
//   (NA, NB, waves) = (4, 4, 4)  128 x 128 per wave, 256 accumulator registers, ONE wave per SIMD (512-register budget)
//                     (4, 2, 8)  128 x 64 per wave, two waves per SIMD
//                     (2, 2, 8)  64 x 64
// hipcc --offload-arch=gfx950 -O3 lds_mfma32.hip -o lds_mfma32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NA, int NB, int THREADS, int MODE>   // MODE 0 burst, 1 interleaved
__global__ __launch_bounds__(THREADS) void k32(float* out, int iters) {
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
  auto mm = [&](bf16x8 (&fa)[NA], bf16x8 (&fb)[NB]) {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
  };
  constexpr int NR = NA + NB, NM = NA * NB;
  auto interleave = [&]() {
    // NM MFMAs and NR reads in one scheduling region: 1 read after every NM / NR MFMAs
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      __builtin_amdgcn_sched_group_barrier(0x008, NM / NR, 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         // DS read
    }
  };
  load(fa0, fb0, 0);
  for (int it = 0; it < iters; it += 2) {
    if constexpr (MODE == 0) {
      load(fa1, fb1, it + 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      load(fa0, fb0, it + 2);
      __builtin_amdgcn_sched_barrier(0);
      mm(fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      load(fa1, fb1, it + 1);
      mm(fa0, fb0);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      load(fa0, fb0, it + 2);
      mm(fa1, fb1);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
    }
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
  printf("%-72s %8.3f ms  %8.1f TFLOP/s\n", name, ms / 5, flop * 5 / (ms * 1e-3) / 1e12);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 4);
  const int iters = 4000, grid = 256;
  const size_t lds = 100 * 1024;
#define CASE(NA, NB, THREADS, MODE)                                                                                  \
  {                                                                                                                   \
    auto kern = k32<NA, NB, THREADS, MODE>;                                                                           \
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
    char nm[128];                                                                                                     \
    snprintf(nm, 128, "32x32x16 %s: %d+%d reads, %2d MFMAs per K=16 step, %d waves/WG", MODE ? "interleaved" : "burst      ", \
             NA, NB, NA* NB, THREADS / 64);                                                                           \
    run(nm, [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, 0, out, iters); },                         \
        (double)grid * (THREADS / 64) * iters * NA * NB * 32768.0);                                                   \
  }
  for (int rep = 0; rep < 2; ++rep) {
    CASE(4, 4, 256, 0)
    CASE(4, 4, 256, 1)
    CASE(4, 2, 512, 0)
    CASE(4, 2, 512, 1)
    CASE(2, 4, 512, 1)
    CASE(2, 2, 512, 0)
    CASE(2, 2, 512, 1)
    CASE(4, 2, 256, 1)
  }
  return 0;
}
