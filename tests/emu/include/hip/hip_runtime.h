// TEST INFRASTRUCTURE ONLY -- never part of the product path.
//
// A host-side stand-in for <hip/hip_runtime.h> so that the *unmodified* kernel sources in
// fast-srgan_amd/csrc can be compiled with the host clang++ and executed on CPU threads:
// one OS thread per GPU thread, workgroups run one after another, `__shared__` becomes a
// function-level static, wave-collective operations (MFMA, shuffles, LDS transpose reads)
// rendezvous the 64 threads of a wave and reproduce the gfx950 lane layouts documented in
// /opt/skills/guides/cdna_hip_programming.md section 3.  It exists so index arithmetic of the
// kernels can be debugged without a GPU (`pytest -m "not gpu"`); the shipped library is built
// by hipcc for gfx950 only and nothing in the package can load the emulated build.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <functional>
#include <vector>

#define FSR_EMU_BUILD 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define FSR_LDS_PTR(T, p) ((T*)(p))
#define FSR_GLOBAL_PTR(T, p) ((T*)(p))
#define FSR_WAIT_LOADS() ((void)0)
#define FSR_WAIT_DMA() ((void)0)
typedef uintptr_t fsr_lds_addr_t;
#define FSR_LDS_ADDR(p) ((uintptr_t)(p))
#define FSR_GLDS16_AT(g, a) emu::global_load_lds((const void*)(g), (void*)(a), 16, 0)
#define FSR_GLDS16_SAT(sb, vo, a) emu::global_load_lds((const void*)((const char*)(sb) + (vo)), (void*)(a), 16, 0)
#define FSR_TOUCH(v) ((void)(v))
// buffer-addressed LDS-DMA (conv_tall3.hip): base pointer + byte count; a voffset beyond the count reads zeros
struct fsr_buf_t { const char* p; unsigned bytes; };
#define fsr_make_buf(ptr, nbytes) (fsr_buf_t{(const char*)(ptr), (unsigned)(nbytes)})
#define FSR_BLDS16(b, vo, so, a) emu::buffer_load_lds16((b).p, (b).bytes, (vo), (so), (void*)(a))
#define FSR_WAIT_VM(n) ((void)0)
#define FSR_WAIT_LGKM0() ((void)0)
#define FSR_BARRIER() emu::block_sync()
#define FSR_WAVE_SYNC() emu::wave_sync()

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct __attribute__((aligned(16))) uint4 { unsigned x, y, z, w; };
struct __attribute__((aligned(8))) uint2 { unsigned x, y; };
struct __attribute__((aligned(16))) float4 { float x, y, z, w; };
struct __attribute__((aligned(8))) float2 { float x, y; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
  memset(p, v, n);
  return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) {
  memcpy(d, s, n);
  return hipSuccess;
}
enum { hipMemcpyDeviceToDevice = 3 };
template <class F> inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
struct hipDeviceProp_t {
  char name[256];
  char gcnArchName[256];
  int multiProcessorCount;
  size_t totalGlobalMem;
  size_t sharedMemPerBlock;
  int clockRate;
};
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p));
  strcpy(p->name, "host-emulation");
  strcpy(p->gcnArchName, "emu");
  p->multiProcessorCount = 1;
  return hipSuccess;
}

namespace emu {

// Execution model: the lanes of a workgroup are FIBERS on one OS thread (a 30-line x86-64 context switch in
// emu_runtime.cpp), scheduled round-robin; wave / workgroup barriers are generation counters that yield until every
// member has arrived.  (One OS thread per lane with pthread barriers spent > 95 % of its time in futex calls.)
struct Barrier {
  int count = 0, total = 0;
  unsigned gen = 0;
};
struct WaveState {
  Barrier bar;
  float fa[64][8];
  float fb[64][8];
  uint64_t u64[64];
  float f32[64];
  int i32[64];
};
struct BlockState {
  Barrier bar;
  int nthreads;
  std::vector<WaveState*> waves;
  char* dyn_smem;
};
struct ThreadCtx {
  dim3 tid, bid, bdim, gdim;
  int lin, lane, wave;
  BlockState* blk;
  WaveState* w;
};
struct Fiber {
  void* sp = nullptr;      // saved stack pointer while the fiber is switched out
  char* stack = nullptr;
  bool done = false;
  ThreadCtx c;
};
struct Sched {
  std::vector<Fiber> fibers;
  int cur = 0, live = 0;
  void* main_sp = nullptr;
  const std::function<void()>* body = nullptr;
  dim3 grid;
};
extern "C" void fsr_emu_switch(void** save_sp, void* load_sp);   // emu_runtime.cpp
Sched& sched();                                                   // emu_runtime.cpp (one per OS thread)
void fiber_yield();                                               // run the next live fiber of the workgroup
void run_workgroups(dim3 grid, dim3 block, const std::function<void()>& body);

inline ThreadCtx& cur_ctx() { Sched& s = sched(); return s.fibers[s.cur].c; }
#define ctx (::emu::cur_ctx())

inline void barrier_wait(Barrier& b) {
  const unsigned g = b.gen;
  if (++b.count == b.total) {
    b.count = 0;
    ++b.gen;
    return;
  }
  while (b.gen == g) fiber_yield();
}
inline void wave_sync() { barrier_wait(ctx.w->bar); }
inline void block_sync() { barrier_wait(ctx.blk->bar); }

inline void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
  (void)smem;
  const int nthreads = (int)(block.x * block.y * block.z);
  if (nthreads % 64 != 0) {
    fprintf(stderr, "emu: block size must be a multiple of 64\n");
    abort();
  }
  run_workgroups(grid, block, body);
}

// ---- wave collectives ------------------------------------------------------------------
inline float shfl_idx(float v, int src) {
  ctx.w->f32[ctx.lane] = v;
  wave_sync();
  float r = ctx.w->f32[src & 63];
  wave_sync();
  return r;
}
inline int shfl_idx_i(int v, int src) {
  ctx.w->i32[ctx.lane] = v;
  wave_sync();
  int r = ctx.w->i32[src & 63];
  wave_sync();
  return r;
}

inline float bf16_bits_to_float(uint16_t h) {
  uint32_t u = ((uint32_t)h) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

typedef float f32x4_e __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x32_bf16: A[i][k] held by lane i+16*(k/8) element k%8; B[k][j] by lane
// j+16*(k/8) element k%8; D[i][j] -> lane j+16*(i/4), reg i%4.
template <class VA, class VB>
inline f32x4_e mfma_bf16_16x16x32(VA a, VB b, f32x4_e c) {
  uint16_t ua[8], ub[8];
  memcpy(ua, &a, 16);
  memcpy(ub, &b, 16);
  WaveState* w = ctx.w;
  for (int e = 0; e < 8; ++e) {
    w->fa[ctx.lane][e] = bf16_bits_to_float(ua[e]);
    w->fb[ctx.lane][e] = bf16_bits_to_float(ub[e]);
  }
  wave_sync();
  const int j = ctx.lane & 15;
  for (int r = 0; r < 4; ++r) {
    const int i = (ctx.lane >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 32; ++k)
      acc = fmaf(w->fa[i + 16 * (k >> 3)][k & 7], w->fb[j + 16 * (k >> 3)][k & 7], acc);
    c[r] = acc;
  }
  wave_sync();
  return c;
}
// v_mfma_f32_16x16x32_f16: the same lane layout with IEEE half operands
template <class VA, class VB>
inline f32x4_e mfma_f16_16x16x32(VA a, VB b, f32x4_e c) {
  _Float16 ha[8], hb[8];
  memcpy(ha, &a, 16);
  memcpy(hb, &b, 16);
  WaveState* w = ctx.w;
  for (int e = 0; e < 8; ++e) {
    w->fa[ctx.lane][e] = (float)ha[e];
    w->fb[ctx.lane][e] = (float)hb[e];
  }
  wave_sync();
  const int j = ctx.lane & 15;
  for (int r = 0; r < 4; ++r) {
    const int i = (ctx.lane >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 32; ++k)
      acc = fmaf(w->fa[i + 16 * (k >> 3)][k & 7], w->fb[j + 16 * (k >> 3)][k & 7], acc);
    c[r] = acc;
  }
  wave_sync();
  return c;
}
// v_mfma_f32_16x16x4_f32: A[i][k] lane i+16k ; B[k][j] lane j+16k
inline f32x4_e mfma_f32_16x16x4(float a, float b, f32x4_e c) {
  WaveState* w = ctx.w;
  w->fa[ctx.lane][0] = a;
  w->fb[ctx.lane][0] = b;
  wave_sync();
  const int j = ctx.lane & 15;
  for (int r = 0; r < 4; ++r) {
    const int i = (ctx.lane >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(w->fa[i + 16 * k][0], w->fb[j + 16 * k][0], acc);
    c[r] = acc;
  }
  wave_sync();
  return c;
}
// ds_read_b64_tr_b16 (assumed semantics, pinned on hardware by tests/test_hw_probes.py):
// within each 16-lane group every lane supplies the address of 4 contiguous 16-bit values;
// lane i of the group receives, for j = 0..3, element (i & 3) of the 8 bytes supplied by
// group-lane 4*j + (i >> 2).
typedef short s16x4_e __attribute__((ext_vector_type(4)));
inline s16x4_e ds_read_tr16(const void* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  ctx.w->u64[ctx.lane] = v;
  wave_sync();
  s16x4_e r;
  const int g = ctx.lane & ~15, i = ctx.lane & 15;
  for (int j = 0; j < 4; ++j) {
    uint64_t src = ctx.w->u64[g + 4 * j + (i >> 2)];
    r[j] = (short)((src >> (16 * (i & 3))) & 0xffff);
  }
  wave_sync();
  return r;
}

// global_load_lds_dwordx4 & co.: every lane copies `size` bytes from ITS global address to
// (LDS base of lane 0, the value the hardware takes from M0) + offset + lane * size.
inline void global_load_lds(const void* gsrc, void* lds_base, unsigned size, int offset) {
  ctx.w->u64[ctx.lane] = (uint64_t)(uintptr_t)lds_base;
  wave_sync();
  char* base = (char*)(uintptr_t)ctx.w->u64[0];
  wave_sync();
  memcpy(base + offset + (size_t)ctx.lane * size, gsrc, size);
}

// buffer_load_dwordx4 ... offen lds: 16 bytes per lane from base + voffset + soffset; the range check covers voffset only
inline void buffer_load_lds16(const char* base, unsigned bytes, unsigned voff, unsigned soff, void* lds_base) {
  ctx.w->u64[ctx.lane] = (uint64_t)(uintptr_t)lds_base;
  wave_sync();
  char* dst = (char*)(uintptr_t)ctx.w->u64[0];
  wave_sync();
  if ((uint64_t)voff + 16 <= (uint64_t)bytes) memcpy(dst + (size_t)ctx.lane * 16, base + voff + soff, 16);
  else memset(dst + (size_t)ctx.lane * 16, 0, 16);
}

typedef float f32x16_e __attribute__((ext_vector_type(16)));
// v_mfma_f32_32x32x16_{bf16,f16}: A[i][k] held by lane i+32*(k/8) element k%8; B[k][j] by lane j+32*(k/8) element k%8;
// D[i][j] -> lane j+32*((i/4)%2), reg (i%4) + 4*(i/8)
inline f32x16_e mfma_32x32x16_core(f32x16_e c) {
  WaveState* w = ctx.w;
  wave_sync();
  const int j = ctx.lane & 31;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (ctx.lane >> 5);
    float acc = c[r];
    for (int k = 0; k < 16; ++k) acc = fmaf(w->fa[i + 32 * (k >> 3)][k & 7], w->fb[j + 32 * (k >> 3)][k & 7], acc);
    c[r] = acc;
  }
  wave_sync();
  return c;
}
template <class VA, class VB>
inline f32x16_e mfma_bf16_32x32x16(VA a, VB b, f32x16_e c) {
  uint16_t ua[8], ub[8];
  memcpy(ua, &a, 16);
  memcpy(ub, &b, 16);
  for (int e = 0; e < 8; ++e) {
    ctx.w->fa[ctx.lane][e] = bf16_bits_to_float(ua[e]);
    ctx.w->fb[ctx.lane][e] = bf16_bits_to_float(ub[e]);
  }
  return mfma_32x32x16_core(c);
}
template <class VA, class VB>
inline f32x16_e mfma_f16_32x32x16(VA a, VB b, f32x16_e c) {
  _Float16 ha[8], hb[8];
  memcpy(ha, &a, 16);
  memcpy(hb, &b, 16);
  for (int e = 0; e < 8; ++e) {
    ctx.w->fa[ctx.lane][e] = (float)ha[e];
    ctx.w->fb[ctx.lane][e] = (float)hb[e];
  }
  return mfma_32x32x16_core(c);
}

inline float atomic_add_f32(float* addr, float v) {
  uint32_t* p = (uint32_t*)addr;
  uint32_t old = __atomic_load_n(p, __ATOMIC_RELAXED);
  for (;;) {
    float f;
    memcpy(&f, &old, 4);
    float nf = f + v;
    uint32_t nu;
    memcpy(&nu, &nf, 4);
    if (__atomic_compare_exchange_n(p, &old, nu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return f;
  }
}

#undef ctx
}  // namespace emu

#define threadIdx (emu::cur_ctx().tid)
#define blockIdx (emu::cur_ctx().bid)
#define blockDim (emu::cur_ctx().bdim)
#define gridDim (emu::cur_ctx().gdim)
#define warpSize 64

#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)emu::cur_ctx().blk->dyn_smem;

#define hipLaunchKernelGGL(kern, grid, block, smem, stream, ...) \
  emu::launch((grid), (block), (smem), [&]() { kern(__VA_ARGS__); })

inline void __syncthreads() { emu::block_sync(); }
inline float __shfl_xor(float v, int m, int = 64) { return emu::shfl_idx(v, emu::cur_ctx().lane ^ m); }
inline int __shfl_xor(int v, int m, int = 64) { return emu::shfl_idx_i(v, emu::cur_ctx().lane ^ m); }
inline float __shfl_down(float v, int d, int = 64) {
  return emu::shfl_idx(v, emu::cur_ctx().lane + d < 64 ? emu::cur_ctx().lane + d : emu::cur_ctx().lane);
}
inline float __shfl(float v, int src, int = 64) { return emu::shfl_idx(v, src); }
inline float atomicAdd(float* a, float v) { return emu::atomic_add_f32(a, v); }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }

#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu::mfma_bf16_16x16x32((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) emu::mfma_f16_16x16x32((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu::mfma_f32_16x16x4((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu::mfma_bf16_32x32x16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu::mfma_f16_32x32x16((a), (b), (c))
#define __builtin_amdgcn_sched_group_barrier(mask, n, id) ((void)0)
#define __builtin_amdgcn_s_setprio(p) ((void)0)
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) emu::ds_read_tr16((const void*)(p))
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) emu::global_load_lds((const void*)(g), (void*)(l), (size), (off))
