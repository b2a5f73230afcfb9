// Shared device-side helpers for the gfx950 (CDNA4) Fast-SRGAN kernels.
//
// Conventions used by every kernel in this directory:
//   * activations are NHWC ("pixel-major, channels contiguous"), element type
//     T = float (exact-f32 parity mode) or bf16 stored as unsigned short;
//   * a wavefront is 64 lanes; workgroups are 256 threads (4 waves, one per SIMD);
//   * MFMA shapes: v_mfma_f32_16x16x32_bf16 (bf16 mode) and v_mfma_f32_16x16x4_f32
//     (f32 mode, exact fmaf-chain numerics at the f32 vector rate);
//   * accumulator layout (both shapes): lane l, reg r -> row (l>>4)*4+r, col l&15.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

typedef unsigned short bf16_t;  // raw bf16 bits
typedef _Float16 f16_t;         // IEEE half (FSR_F16: the fp16-MFMA mode of BASELINE configs[4])

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // one 16-byte staging unit
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_hw __attribute__((ext_vector_type(8)));

#ifndef FSR_LDS_PTR
#define FSR_LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#endif
// gfx9 counts loads AND stores on vmcnt, and the two kinds complete out of order with each other, so the compiler waits
// vmcnt(0) -- for every store in flight as well -- wherever a loaded register is first used.  A persistent kernel that
// prefetches the next tile before its epilogue therefore waits for the prefetch HERE, before the epilogue's stores are
// issued (the loads have had the whole tile to land): the later use of the prefetched registers then needs no wait and
// the stores drain under the next tile.   0x0F70 = vmcnt(0), expcnt / lgkmcnt untouched.
#ifndef FSR_WAIT_LOADS
#define FSR_WAIT_LOADS() __builtin_amdgcn_s_waitcnt(0x0F70)
#endif
#ifndef FSR_GLOBAL_PTR
#define FSR_GLOBAL_PTR(T, p) ((__attribute__((address_space(1))) T*)(p))
#endif
// One LDS-DMA piece: every lane copies 16 bytes from ITS global address to (LDS address) + 16 * lane (global_load_lds_dwordx4).
// Issued from inline asm ON PURPOSE.  hipcc's waitcnt pass treats the builtin as a FLAT operation that touches two address
// spaces ("pending flat"): while one is in flight on vmcnt -- by design the whole stage -- every s_waitcnt lgkmcnt it emits is
// forced to 0, so software-pipelined ds_read -> MFMA sequences collapse into {reads, wait for ALL of them, MFMAs}.  Hidden in
// asm, the DMA is invisible to that pass: the fragment reads get counted lgkmcnt(N) waits, and the kernel waits for its DMA
// pieces itself (FSR_WAIT_DMA before the barrier that publishes the buffer).
#ifndef FSR_WAIT_DMA
#define FSR_WAIT_DMA() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
// The piece is addressed the way the hardware wants it: destination = a 32-bit LDS byte address (FSR_LDS_ADDR of a shared
// pointer, taken ONCE per kernel, plus integer offsets -- every generic-pointer -> LDS cast in a loop costs a null check and
// 64-bit scalar adds), source = a wave-uniform 64-bit base in SGPRs + a 32-bit per-lane byte offset (no vector 64-bit add).
// M0 is declared clobbered instead of saved and restored.
#ifndef FSR_GLDS16_AT
typedef unsigned fsr_lds_addr_t;
#define FSR_LDS_ADDR(p) ((unsigned)(size_t)FSR_LDS_PTR(char, (p)))
__device__ __forceinline__ void fsr_glds16_at(const void* gsrc, unsigned lds_addr) {
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(dst) : "memory", "m0");
}
__device__ __forceinline__ void fsr_glds16_sat(const void* sbase, unsigned voff, unsigned lds_addr) {
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(dst) : "memory", "m0");
}
#define FSR_GLDS16_AT(g, a) fsr_glds16_at((g), (a))
#define FSR_GLDS16_SAT(sb, vo, a) fsr_glds16_sat((sb), (vo), (a))
#endif
// ---- buffer-addressed LDS-DMA (conv_tall3.hip).  buffer_load_dwordx4 ... offen lds: every lane copies 16 bytes from
// (descriptor base + voffset + soffset) to (LDS address in M0) + 16 * lane; a lane whose voffset lies beyond the descriptor's
// byte count gets ZEROS (hardware range check on voffset only -- soffset is not checked): image borders need no zero page and
// no branch.  Issued from inline asm for the same reason as FSR_GLDS16_*: hipcc never sees the piece, so its own
// lgkmcnt / vmcnt bookkeeping stays counted, and the kernel waits for its pieces itself (FSR_WAIT_VM(n): at most n of the
// YOUNGEST vector-memory operations of this wave are still in flight; loads retire in order).
#ifndef FSR_BLDS16
struct fsr_buf_t { u32x4 v; };
__device__ __forceinline__ fsr_buf_t fsr_make_buf(const void* p, unsigned bytes) {
  const unsigned long long x = (unsigned long long)(size_t)p;
  fsr_buf_t b;
  b.v = (u32x4){(unsigned)x, (unsigned)(x >> 32) & 0xffffu, bytes, 0x00020000u};   // raw buffer, stride 0, 32-bit data format
  return b;
}
__device__ __forceinline__ void fsr_blds16(fsr_buf_t b, unsigned voff, unsigned soff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %0, %2 offen lds" ::"s"(b.v), "v"(voff), "s"(soff), "s"(lds_addr)
               : "memory", "m0");
}
#define FSR_BLDS16(b, vo, so, a) fsr_blds16((b), (vo), (so), (a))
#define FSR_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define FSR_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define FSR_BARRIER() __builtin_amdgcn_s_barrier()
#endif
// Orders the LDS traffic of ONE wave around a wave-private exchange (write in one lane mapping, read back in another): the
// hardware runs a wave's LDS instructions in order, so this only has to stop the compiler from moving them across it.
#ifndef FSR_WAVE_SYNC
#define FSR_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif
// Bulk output stores.  FSR_NT (compile-time bit mask, one bit per kernel family) puts the non-temporal hint on them: a conv
// epilogue streams hundreds of MB through 4 MB L2s whose hot content is the filter and halo lines the LDS-DMA keeps re-reading
// (measured on conv_s2d3.hip's 128-channel layer: +8..10 % with `nt` stores, profiles/r04_nt_stores.txt).
//   1 conv_tall3   2 conv64 family   4 first-layer kernels   8 elementwise   16 conv_s2d3   32 conv_igemm
#ifndef FSR_NT
#define FSR_NT 20
#endif
template <int BIT, typename V>
__device__ __forceinline__ void fsr_st(V* p, V v) {
  if constexpr ((FSR_NT & BIT) != 0) __builtin_nontemporal_store(v, p);
  else *p = v;
}
// A register "use" with no instruction: pins where the compiler places its s_waitcnt for a load's result.
#ifndef FSR_TOUCH
#define FSR_TOUCH(v) asm volatile("" : "+v"(v))
#endif

// dtype / activation / mode enums come from the public ABI header
#include "fsr_hip.h"

__device__ __forceinline__ float bf2f(bf16_t h) {
  return __uint_as_float(((unsigned)h) << 16);
}
// round-to-nearest-even (v_cvt_pk_bf16_f32 on gfx950)
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_hw));
}

template <typename T> struct ElemIO;
template <> struct ElemIO<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ElemIO<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};
template <> struct ElemIO<f16_t> {
  static __device__ __forceinline__ float ld(const f16_t* p) { return (float)*p; }
  static __device__ __forceinline__ void st(f16_t* p, float v) { *p = (f16_t)v; }
};

__device__ __forceinline__ f32x4 mfma_bf16_16x16x32(s16x8 a, s16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_hw, a),
                                                 __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
}
// ---- 16-bit storage types, generically (T = bf16_t or f16_t): two elements of a packed 32-bit word to float, two floats to
// a packed word (round to nearest even), one MFMA step v_mfma_f32_16x16x32_{bf16,f16}.
typedef _Float16 f16x8_hw __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_hw __attribute__((ext_vector_type(2)));
template <typename T> struct Num16;
template <> struct Num16<bf16_t> {
  static __device__ __forceinline__ float lo(unsigned w) { return __uint_as_float(w << 16); }
  static __device__ __forceinline__ float hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
  static __device__ __forceinline__ unsigned pack(float a, float b) { return pack_bf16x2(a, b); }
  static __device__ __forceinline__ f32x4 mfma(s16x8 a, s16x8 b, f32x4 c) { return mfma_bf16_16x16x32(a, b, c); }
};
template <> struct Num16<f16_t> {
  static __device__ __forceinline__ float lo(unsigned w) { return (float)__builtin_bit_cast(f16x2_hw, w)[0]; }
  static __device__ __forceinline__ float hi(unsigned w) { return (float)__builtin_bit_cast(f16x2_hw, w)[1]; }
  static __device__ __forceinline__ unsigned pack(float a, float b) {
    const f16x2_hw v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
  }
  static __device__ __forceinline__ f32x4 mfma(s16x8 a, s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_hw, a), __builtin_bit_cast(f16x8_hw, b), c, 0, 0, 0);
  }
};
template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b) { return Num16<T>::pack(a, b); }
template <typename T> __device__ __forceinline__ float cvt_lo(unsigned w) { return Num16<T>::lo(w); }
template <typename T> __device__ __forceinline__ float cvt_hi(unsigned w) { return Num16<T>::hi(w); }
template <typename T> __device__ __forceinline__ f32x4 mfma16(s16x8 a, s16x8 b, f32x4 c) { return Num16<T>::mfma(a, b, c); }

// ---- FSR_X3, the split-bf16 storage type (include/fsr_hip.h): a logical element is 4 bytes; per pixel and per group of 32
// channels the tensor holds 64 bytes of hi[32] (bf16(v)) followed by 64 bytes of lo[32] (bf16(v - hi)).  x3_t is only a
// pointer type: `p + e` is the logical element e (4-byte steps), and the helpers below turn that pointer into the two physical
// addresses -- byte b of the logical (float-like) layout lies in the 128-byte block b & ~127; its hi half word sits at
// (b & 127) / 2 inside the block and its lo half word 64 bytes later.  Tensor bases are 128-byte aligned (host checked), and
// every vector access covers 4 or 8 consecutive channels of one group, so a V16 unit is two 16-byte accesses 64 bytes apart.
struct x3_t { unsigned raw; };
static_assert(sizeof(x3_t) == 4, "x3_t is a 4-byte element");
template <typename P> __device__ __forceinline__ P* x3_hi_ptr(P* p) {
  const size_t b = (size_t)p;
  return (P*)((b & ~(size_t)127) | ((b & (size_t)127) >> 1));
}
// two floats -> packed bf16 pair of the hi parts and packed pair of the lo parts (both round to nearest even)
__device__ __forceinline__ void x3_split2(float a, float b, unsigned& hi, unsigned& lo) {
  hi = pack_bf16x2(a, b);
  lo = pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ float x3_join_lo(unsigned hi, unsigned lo) { return __uint_as_float(hi << 16) + __uint_as_float(lo << 16); }
__device__ __forceinline__ float x3_join_hi(unsigned hi, unsigned lo) { return __uint_as_float(hi & 0xffff0000u) + __uint_as_float(lo & 0xffff0000u); }
template <> struct ElemIO<x3_t> {
  static __device__ __forceinline__ float ld(const x3_t* p) {
    const bf16_t* h = (const bf16_t*)x3_hi_ptr(p);
    return bf2f(h[0]) + bf2f(h[32]);
  }
  static __device__ __forceinline__ void st(x3_t* p, float v) {
    bf16_t* h = (bf16_t*)x3_hi_ptr(p);
    const bf16_t hi = f2bf(v);
    h[0] = hi;
    h[32] = f2bf(v - bf2f(hi));
  }
};
// 4 consecutive channels (one lane's accumulator tile row): two 8-byte accesses
__device__ __forceinline__ void x3_st4(x3_t* p, f32x4 v) {
  unsigned h0, l0, h1, l1;
  x3_split2(v[0], v[1], h0, l0);
  x3_split2(v[2], v[3], h1, l1);
  char* h = (char*)x3_hi_ptr(p);
  *(u32x2*)h = (u32x2){h0, h1};
  *(u32x2*)(h + 64) = (u32x2){l0, l1};
}
__device__ __forceinline__ f32x4 x3_ld4(const x3_t* p) {
  const char* h = (const char*)x3_hi_ptr(p);
  const u32x2 hi = *(const u32x2*)h, lo = *(const u32x2*)(h + 64);
  return (f32x4){x3_join_lo(hi.x, lo.x), x3_join_hi(hi.x, lo.x), x3_join_lo(hi.y, lo.y), x3_join_hi(hi.y, lo.y)};
}

// ---- one 16-byte unit of a storage type viewed as floats (the HBM-bound kernels move one unit per lane and access):
// 8 elements of a 16-bit type, 4 floats, or 8 x3 elements (16 bytes of hi + 16 bytes of lo)
template <typename T, int NTBIT = 0> struct V16 {   // the 16-bit storage types (bf16_t, f16_t)
  static constexpr int N = 8;
  static __device__ __forceinline__ void ld(const T* p, float (&v)[8]) {
    const u32x4 t = *(const u32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = cvt_lo<T>(t[i]);
      v[2 * i + 1] = cvt_hi<T>(t[i]);
    }
  }
  static __device__ __forceinline__ void st(T* p, const float (&v)[8]) {
    u32x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = pack2<T>(v[2 * i], v[2 * i + 1]);
    fsr_st<NTBIT>((u32x4*)p, t);
  }
};
template <int NTBIT> struct V16<float, NTBIT> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
    const f32x4 t = *(const f32x4*)p;
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) {
    *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]};
  }
};
template <int NTBIT> struct V16<x3_t, NTBIT> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void ld(const x3_t* p, float (&v)[8]) {
    const char* h = (const char*)x3_hi_ptr(p);
    const u32x4 hi = *(const u32x4*)h, lo = *(const u32x4*)(h + 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = x3_join_lo(hi[i], lo[i]);
      v[2 * i + 1] = x3_join_hi(hi[i], lo[i]);
    }
  }
  static __device__ __forceinline__ void st(x3_t* p, const float (&v)[8]) {
    u32x4 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned a, b;
      x3_split2(v[2 * i], v[2 * i + 1], a, b);
      hi[i] = a;
      lo[i] = b;
    }
    char* h = (char*)x3_hi_ptr(p);
    fsr_st<NTBIT>((u32x4*)h, hi);
    fsr_st<NTBIT>((u32x4*)(h + 64), lo);
  }
};

__device__ __forceinline__ f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// v > 0 ? v : v * slope (ReLU: slope 0, identity: slope 1).  NOT max(v, 0) + slope * min(v, 0): v_max_f32 / v_min_f32 return the
// non-NaN operand, so that form turns a NaN into 0 -- in a DATA-GRADIENT epilogue that silently zeroed the NaNs of an overflowed fp16
// backward on their way to the optimizer's non-finite check (round 6: the x3v mode's fp16 perceptual branch; found by
// tests/test_x3.py::test_x3v_overflow_in_the_fp16_perceptual_branch_skips_the_iteration).  Same result on every finite input.
__device__ __forceinline__ float act_slope(float v, float slope) { return v > 0.f ? v : v * slope; }

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
  switch (act) {
    case FSR_ACT_RELU: return v > 0.f ? v : 0.f;
    case FSR_ACT_LEAKY:
    case FSR_ACT_PRELU: return v > 0.f ? v : v * slope;
    case FSR_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// inference.py:53-56 on one head output y = tanh(z) in (-1,1): ((y + 1) / 2) * 255 in f32, then the C cast to uint8
// (truncation toward zero; the value lies in [0, 255))
__device__ __forceinline__ unsigned char image_u8(float y) {
  const float w = ((y + 1.f) / 2.f) * 255.f;
  return (unsigned char)w;
}

// sum over the 64 lanes of a wave; every lane gets the total
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N).  Keeps every index into
// register arrays (accumulators) a constant -- a runtime index sends the whole array to scratch.
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// XCD-aware bijective remap of a linear workgroup id: consecutive logical ids land on
// the same XCD (dispatcher places hardware block b on XCD b % 8), so neighbouring tiles
// that share halo rows / filter slices hit the same 4 MiB L2. Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
