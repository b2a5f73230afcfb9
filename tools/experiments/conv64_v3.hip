// conv64_v3.hip -- 64-input-channel 3x3 convolutions (stride 1, 16-bit), third form: the whole filter block resident in LDS like
// conv64_v2_kernel (conv64_persistent.hip), but on v_mfma_f32_32x32x16_{bf16,f16} with a 64-pixel x 64-channel wave tile and
// compile-time fragment addresses -- the operand mapping, swizzles and accumulator layout of conv_tall3_body.h.
//
// Replaces, for the 16-bit modes, the torch.nn.Conv2d(64, C, 3, padding=1) calls of the reference with C = 64, 128 or 256:
//   /root/reference/model.py:47-64, 86-93  ResidualBlock conv1 / conv2 and the bottleneck of the Generator (forwards with the
//                                          InstanceNorm sums, data gradients with the fused activation mask / skip addend),
//   /root/reference/model.py:30-37         the two 64 -> 256 up-sampling convolutions (PixelShuffle + PReLU epilogue),
//   /root/reference/model.py:154-159       Discriminator 64 -> 128,
//   torchvision vgg19.features 2 and 5 behind model.py:8 (conv1_2 with the fused 2x2 max-pool, conv2_1).
//
// Why a third form (round-5 verdict, item 4: conv64_v2 sits at 0.28-0.36 of the MFMA peak and of HBM -- bound by neither).
//   conv64_v2's wave owns 32 pixels x 64 channels of v_mfma_f32_16x16x32: 6 ds_read_b128 and 8 MFMAs of ~17 cycles per step, the
//   tap table decoded at run time in front of every step.  That is 0.75 reads and a dozen address instructions per 17-cycle MFMA: the
//   wave cannot issue them in the MFMAs' shadow (MI355X_MICROARCH.md: <= 5 single-issue instructions hide behind a 32-cycle
//   MFMA).  Here a wave owns 64 pixels x 64 channels of 32x32x16: 4 reads per 4 MFMAs of 32 cycles, every LDS offset an immediate.
//   And the channel blocks of one tile range (64 -> 128 / 256 outputs) run on the SAME XCD, so the input halo that conv64_v2
//   fetched from HBM once per block (FETCH 1.41x / 4x the input, profiles/r05_pmc_fetch_write_conv64.txt) is an L2 hit for all
//   but the first.
//
// Work decomposition
//   workgroup  8 waves, one per CU (LDS), persistent over a contiguous range of tiles of ONE 64-channel output block
//   tile       16 rows x 32 columns of output pixels; wave (rg = wave & 3, ch = wave >> 2) owns rows 4 rg .. 4 rg + 3, columns
//              16 ch .. 16 ch + 15: 2 pixel fragments (2 rows x 16 columns each) x 2 filter fragments (32 channels each)
//   K loop     2 chunks of 32 input channels x 9 taps x 2 halves of 16 channels = 36 substeps of 4 MFMAs; the fragment reads
//              of substep s + 1 are issued between the MFMAs of substep s
// LDS (bytes)  filter[2 chunks][9 taps][64 rows][64 B]   73,728   resident; row R keeps 16-byte unit u at u ^ ((R >> 2) & 3)
//              halo[2][40 KB]    18 x 34 pixels x 32 channels (64 B per pixel, unit u of column x at u ^ ((x >> 1) & 3)): chunk c of
//                                a tile lives in buffer c; both arrive by LDS-DMA (buffer_load ... lds, borders = range misses)
//              bias[64 floats]
// Synchronisation: ONE s_barrier per chunk (72 MFMAs per wave), placed before the chunk's last substep: it publishes the next
//   chunk's halo (every wave waits for its own five pieces first) and retires the reads of the buffer the DMA overwrites next.
// Epilogue: deferred into the next tile like conv64_v2's (the accumulators are copied; waves 0-3 store after substep 3 of the next
//   tile's first chunk, waves 4-7 -- their SIMD partners -- after substep 11), so one wave of a SIMD stores while the other owns
//   the matrix pipe.  Output mapping as conv_tall3: a lane's 16 accumulator registers are 16 CONSECUTIVE channels of one pixel.
// Results: the summation order differs from conv64_v2's (K is walked chunk by chunk), so the two agree to f32 rounding, not bit
//   for bit; the statistics are per (4-row x 16-column) wave patch, added in a fixed order by reduce.hip (no atomics).
#include "fsr_common.h"
#include "fsr_conv_args.h"
#include "fsr_host.h"

#include <stdlib.h>

// Ablation builds (-DFSR_ABLV3=<mask>; results WRONG on purpose, the product library is built with 0):
//   1 no output stores   2 no halo DMA after the prologue   4 no MFMAs   8 no epilogue at all   16 no vmcnt wait before the barriers
#ifndef FSR_ABLV3
#define FSR_ABLV3 0
#endif

namespace {

constexpr int V3_ABL = FSR_ABLV3;

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T> struct Mfma32v;
template <> struct Mfma32v<bf16_t> {
  static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
  }
};
template <> struct Mfma32v<f16_t> {
  static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_hw, a), __builtin_bit_cast(f16x8_hw, b), c, 0, 0, 0);
  }
};

constexpr int V3_TH = 16, V3_TW = 32;                  // output tile
constexpr int V3_P = V3_TW + 2, V3_HR = V3_TH + 2;     // halo columns / rows
constexpr int V3_ROWB = V3_P * 64;                     // bytes per halo row (64-byte pixels)
constexpr int V3_HUNITS = V3_HR * V3_P * 4;            // 16-byte units of one halo chunk (2448)
constexpr int V3_HPW = 5;                              // DMA pieces per chunk and wave (8 x 5 = 40 >= 39)
constexpr int V3_HALO_BYTES = 8 * V3_HPW * 1024;       // 40,960
constexpr int V3_W_BYTES = 2 * 9 * 64 * 64;            // 73,728
constexpr int V3_BIAS_OFF = V3_W_BYTES + 2 * V3_HALO_BYTES;
constexpr int V3_LDS = V3_BIAS_OFF + 256;
static_assert(V3_HUNITS <= 8 * V3_HPW * 64, "the halo fits the pieces");

__device__ __forceinline__ int v3_swz_row(int R) { return (R >> 2) & 3; }
__device__ __forceinline__ int v3_swz_col(int x) { return (x >> 1) & 3; }

template <typename V>
__device__ __forceinline__ V v3_lds_read(const char* smem, unsigned off) {
  return *FSR_LDS_PTR(const V, smem + off);
}

// VAR: 0 = plain family (bias, ReLU / LeakyReLU / PReLU / identity, PixelShuffle store, pre-activation copy, fused 2x2 max-pool),
//      1 = InstanceNorm statistics of the pre-activation, 2 = fused activation-gradient mask / skip addend (data gradients)
template <typename T, int VAR>
__global__ __launch_bounds__(512) void conv64_v3_kernel(const ConvKArgs a) {
  constexpr bool STATS = VAR == 1, MASK = VAR == 2;
  HIP_DYNAMIC_SHARED(char, smem)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave & 3, ch = wave >> 2;
  const int l31 = lane & 31, hi = lane >> 5, l15 = lane & 15, lrow = (lane >> 4) & 1;
  const fsr_lds_addr_t lds0 = FSR_LDS_ADDR(smem);

  // ---- which tiles: workgroup b -> XCD b & 7 (the dispatcher's round robin); the nblk channel blocks of tile range r get
  // consecutive slots of ONE XCD, so the halo a range streams is fetched from HBM by the first of them and hits L2 for the rest
  const fsr_buf_t in_buf = fsr_make_buf(a.in, (unsigned)((size_t)a.N * a.IH * a.IW * 64 * sizeof(T)));
  const fsr_buf_t w_buf = fsr_make_buf(a.wpk, (unsigned)((size_t)9 * a.CoutPad * 64 * sizeof(T)));

  const int nblk = a.Cout >> 6;
  const int q_ = (int)blockIdx.x >> 3;
  const int nb = q_ % nblk;
  const int range = (q_ / nblk) * 8 + ((int)blockIdx.x & 7);
  const int per = a.nblk_n;                                  // tiles per range (host)
  const int ntiles = a.t3_ntiles;
  const int tile_begin = range * per;
  if (tile_begin >= ntiles) return;
  const int tile_end = tile_begin + per < ntiles ? tile_begin + per : ntiles;
  const int tiles_per_img = a.tiles_x * a.tiles_y;

  const bool ps = VAR == 0 && a.ps != 0;
  const bool pool2 = VAR == 0 && a.pool2 != 0;
  float slope = (a.act == FSR_ACT_PRELU) ? a.prelu[0] : a.slope;
  if (a.act == FSR_ACT_NONE) slope = 1.f;
  if (a.act == FSR_ACT_RELU) slope = 0.f;

  // ---- the filter block: 72 pieces of 16 rows (chunk c, tap t, quarter g), nine per wave, once per workgroup.  LDS row i of a
  // 32-row fragment holds output channel 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3) (conv_tall3's mapping: a lane's 16
  // accumulator registers are then 16 consecutive channels)
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int q = wave + 8 * i;
    const int c = q / 36, rem = q - 36 * c, t = rem >> 2, g = rem & 3;
    const int R = g * 16 + (lane >> 2), ul = lane & 3, i5 = R & 31;
    const int co = (R & ~31) + 16 * ((i5 >> 2) & 1) + (i5 & 3) + 4 * (i5 >> 3);
    const unsigned vo = (unsigned)(((nb * 64 + co) * 64 + c * 32 + ((ul ^ v3_swz_row(R)) << 3)) * (int)sizeof(T));
    // (a run-time index into the kernel-argument block would send the WHOLE block to scratch: select among the nine scalars)
    unsigned wo = a.t3_woff[0];
    static_for<1, 9>([&](auto tc) {
      if (t == decltype(tc)::value) wo = a.t3_woff[decltype(tc)::value];
    });
    FSR_BLDS16(w_buf, vo, wo, lds0 + (fsr_lds_addr_t)(q * 1024));
  }
  if (tid < 64) {
    float b = 0.f;
    if (a.bias) b = ps ? a.bias[4 * tid + nb] : a.bias[nb * 64 + tid];      // PixelShuffle: torch order 4 * channel + quadrant
    *FSR_LDS_PTR(float, smem + V3_BIAS_OFF + 4 * tid) = b;
  }

  // ---- loop-invariant per-lane addresses
  unsigned aoff[2];      // filter fragment n of (chunk c, tap t), k half j: (c * 9 + t) * 4096 + n * 2048 + aoff[j]
#pragma unroll
  for (int j = 0; j < 2; ++j) aoff[j] = (unsigned)(l31 * 64 + (((2 * j + hi) ^ v3_swz_row(l31)) << 4));
  unsigned boff[3][2];   // pixel fragment m of tap (ky, kx), k half j, chunk buffer c: W + c * HALO + (2 m + ky) * ROWB + boff[kx][j]
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      boff[kx][j] = (unsigned)(V3_W_BYTES + ((rg * 4 + lrow) * V3_P + ch * 16 + l15 + kx) * 64 + (((2 * j + hi) ^ v3_swz_col(l15 + kx)) << 4));

  struct TileC { int img, gy0, gx0; };
  auto coords = [&](int tile) __attribute__((always_inline)) {
    TileC tc;
    tc.img = tile / tiles_per_img;
    const int rem = tile - tc.img * tiles_per_img;
    const int ty = rem / a.tiles_x;
    tc.gy0 = ty * V3_TH;
    tc.gx0 = (rem - ty * a.tiles_x) * V3_TW;
    return tc;
  };
  // DMA source offsets of this wave's five halo pieces (piece wave + 8 k): byte offset of the lane's 16 bytes of chunk 0
  auto halo_src = [&](const TileC& tc, unsigned (&hv)[V3_HPW]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < V3_HPW; ++k) {
      const int U = (wave + 8 * k) * 64 + lane;
      const int hp = U >> 2, ul = U & 3;
      const int hy = hp / V3_P, hx = hp - hy * V3_P;
      const int iy = tc.gy0 - 1 + hy, ix = tc.gx0 - 1 + hx;
      unsigned o = ~0u;                                          // beyond the buffer: the DMA writes zeros
      if (U < V3_HUNITS && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW)
        o = (unsigned)((((tc.img * a.IH + iy) * a.IW + ix) * 64 + ((ul ^ v3_swz_col(hx)) << 3)) * (int)sizeof(T));
      hv[k] = o;
    }
  };
  auto dma_halo = [&](const unsigned (&hv)[V3_HPW], int c, int k) __attribute__((always_inline)) {
    FSR_BLDS16(in_buf, hv[k], (unsigned)(c * 64), lds0 + (fsr_lds_addr_t)(V3_W_BYTES + c * V3_HALO_BYTES + (wave + 8 * k) * 1024));
  };

  f32x16 acc[2][2], accp[2][2];
  s16x8 fa[2][2], fb[2][2];
  auto acc_init = [&]() __attribute__((always_inline)) {       // the accumulators start at the bias
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      f32x16 b0;
      const unsigned bo = (unsigned)(V3_BIAS_OFF + (n * 32 + hi * 16) * 4);
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const f32x4 b = v3_lds_read<f32x4>(smem, bo + 16 * e4);
#pragma unroll
        for (int e = 0; e < 4; ++e) b0[4 * e4 + e] = b[e];
      }
      acc[n][0] = b0;
      acc[n][1] = b0;
    }
  };
  // fragment r (need order a0 b0 b1 a1) of substep s = 2 t + j of chunk c, into register set `buf`
  auto read_frag = [&](auto rc, auto bufc, auto cc, auto sc) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value, buf = decltype(bufc)::value, c = decltype(cc)::value, s = decltype(sc)::value;
    constexpr int t = s >> 1, j = s & 1, ky = t / 3, kx = t % 3;
    if constexpr (r == 0 || r == 3) {
      constexpr int n = r == 0 ? 0 : 1;
      fa[buf][n] = v3_lds_read<s16x8>(smem, aoff[j] + (unsigned)((c * 9 + t) * 4096 + n * 2048));
    } else {
      constexpr int m = r - 1;
      fb[buf][m] = v3_lds_read<s16x8>(smem, boff[kx][j] + (unsigned)(c * V3_HALO_BYTES + (2 * m + ky) * V3_ROWB));
    }
  };

  // ---- the deferred epilogue: tile `e_tc`, accumulators in accp
  // (ep_finish has three call sites: without always_inline hipcc emits it as a function, the closure -- every captured variable, the
  // kernel-argument block included -- moves to scratch, and the DMA's descriptor operands stop being SGPRs)
  TileC e_tc = {0, 0, 0};
  T* outp = (T*)a.out;
  T* prep = (T*)a.preact;
  const T* maskp = MASK ? (const T*)a.dmask : nullptr;
  // partc: -1 = the whole epilogue; 0 .. 3 = the statistics (part 0 only) and fragment (n, m) = (part >> 1, part & 1) alone
  // (-DFSR_V3_SPREAD: the four fragments are stored two substeps apart instead of in one burst)
  auto ep_part = [&](auto partc) __attribute__((always_inline)) {
    constexpr int PART = decltype(partc)::value;
    const int gx = e_tc.gx0 + ch * 16 + l15;
    if constexpr (STATS && PART <= 0) {
      // sums and sums of squares of the pre-activation over the wave's 4 x 16 patch, eight channels at a time (conv_tall3's
      // butterfly: the register set halves at the first three steps), one partial slot per patch
      auto butterfly = [&](float (&x)[8]) {
        static_for<0, 3>([&](auto sc) {
          constexpr int st = decltype(sc)::value, M = 16 >> st, C = 4 >> st;
          const bool up = (lane & M) != 0;
#pragma unroll
          for (int i = 0; i < C; ++i) {
            const float send = up ? x[i] : x[i + C];
            const float keep = up ? x[i + C] : x[i];
            x[i] = keep + __shfl_xor(send, M, 64);
          }
        });
        x[0] += __shfl_xor(x[0], 2, 64);
        x[0] += __shfl_xor(x[0], 1, 64);
      };
      const int py = (e_tc.gy0 >> 2) + rg, px = (e_tc.gx0 >> 4) + ch;
      const bool patch_ok = e_tc.gy0 + rg * 4 < a.GH && e_tc.gx0 + ch * 16 < a.GW;
      static_for<0, 4>([&](auto nc) {
        constexpr int n = decltype(nc)::value / 2, h = decltype(nc)::value % 2;
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
        static_for<0, 2>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          const int gy = e_tc.gy0 + rg * 4 + 2 * m + lrow;
          if (gy < a.GH && gx < a.GW) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float v = accp[n][m][h * 8 + e];
              s1[e] += v;
              s2[e] = fmaf(v, v, s2[e]);
            }
          }
        });
        butterfly(s1);
        butterfly(s2);
        if (patch_ok && !(lane & 3)) {
          const int e = ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
          const int co = nb * 64 + n * 32 + hi * 16 + h * 8 + e;
          const int slot = py * ((a.GW + 15) >> 4) + px;
          float* sp = a.stats + (((size_t)e_tc.img * a.stats_P + slot) * a.Cout + co) * 2;
          sp[0] = s1[0];
          sp[1] = s2[0];
        }
      });
    }
    static_for<0, 2>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      const int cb = n * 32 + hi * 16;                      // channel inside the 64-channel block
      static_for<0, 2>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        if constexpr (PART >= 0 && PART != 2 * n + m) return;
        const int gy = e_tc.gy0 + rg * 4 + 2 * m + lrow;
        const bool ok = gy < a.GH && gx < a.GW;
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = accp[n][m][e];
        auto store16 = [&](T* p) {
          u32x4 p0, p1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            p0[e] = pack2<T>(v[2 * e], v[2 * e + 1]);
            p1[e] = pack2<T>(v[8 + 2 * e], v[8 + 2 * e + 1]);
          }
          if constexpr (!(V3_ABL & 1)) {
            fsr_st<2>((u32x4*)p, (u32x4)(p0));
            fsr_st<2>((u32x4*)(p + 8), (u32x4)(p1));
          } else if (a.GW < 0) {          // (never true: keeps the values alive)
            fsr_st<2>((u32x4*)p, (u32x4)(p0));
            fsr_st<2>((u32x4*)(p + 8), (u32x4)(p1));
          }
        };
        if (pool2) {
          // MaxPool2d(2,2) fused: rows (gy, gy ^ 1) sit in lanes (l, l ^ 16), columns in (l, l ^ 1); the activation is monotonic
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float x = v[e];
            x = fmaxf(x, __shfl_xor(x, 16, 64));
            x = fmaxf(x, __shfl_xor(x, 1, 64));
            v[e] = fmaxf(x, 0.f) + slope * fminf(x, 0.f);
          }
          if (ok && lrow == 0 && !(l15 & 1))
            store16(outp + ((unsigned)((e_tc.img * (a.FOH >> 1) + (gy >> 1)) * (a.FOW >> 1) + (gx >> 1)) * (unsigned)a.Cout + (unsigned)(nb * 64 + cb)));
        } else if (ok) {
          // PixelShuffle(2): channel block nb is quadrant (nb >> 1, nb & 1) of the 2 x 2 block, 64 channels per output pixel
          const unsigned off = ps ? (unsigned)((e_tc.img * 2 * a.FOH + 2 * gy + (nb >> 1)) * (2 * a.FOW) + 2 * gx + (nb & 1)) * 64u + (unsigned)cb
                                  : (unsigned)((e_tc.img * a.FOH + gy) * a.FOW + gx) * (unsigned)a.Cout + (unsigned)(nb * 64 + cb);
          if constexpr (MASK) {      // fused activation backward of the producing layer (gate), or the gradient of a skip connection (addend)
            const u32x4 k0 = *(const u32x4*)(maskp + off), k1 = *(const u32x4*)(maskp + off + 8);
            const float ms = a.dmask_slope;
            if (a.dmask_add) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[2 * e] += cvt_lo<T>(k0[e]);
                v[2 * e + 1] += cvt_hi<T>(k0[e]);
                v[8 + 2 * e] += cvt_lo<T>(k1[e]);
                v[8 + 2 * e + 1] += cvt_hi<T>(k1[e]);
              }
            } else {
              // y > 0 on the raw 16-bit pattern: the element moved to the top of a signed word is positive (bf16 and f16 alike)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[2 * e] = (int)(k0[e] << 16) > 0 ? v[2 * e] : v[2 * e] * ms;
                v[2 * e + 1] = (int)(k0[e] & 0xffff0000u) > 0 ? v[2 * e + 1] : v[2 * e + 1] * ms;
                v[8 + 2 * e] = (int)(k1[e] << 16) > 0 ? v[8 + 2 * e] : v[8 + 2 * e] * ms;
                v[8 + 2 * e + 1] = (int)(k1[e] & 0xffff0000u) > 0 ? v[8 + 2 * e + 1] : v[8 + 2 * e + 1] * ms;
              }
            }
          }
          if constexpr (VAR == 0) {
            if (prep) store16(prep + off);               // the pre-activation a PReLU's backward needs (training)
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], 0.f) + slope * fminf(v[e], 0.f);
          store16(outp + off);
        }
      });
    });
  };

  auto ep_finish = [&]() __attribute__((always_inline)) { ep_part(std::integral_constant<int, -1>{}); };

  // ---- prologue: chunk 0 of the first tile, the filter, the bias
  TileC cur = coords(tile_begin), nxt = cur;
  unsigned hv_cur[V3_HPW], hv_nxt[V3_HPW];
  halo_src(cur, hv_cur);
#pragma unroll
  for (int k = 0; k < V3_HPW; ++k) dma_halo(hv_cur, 0, k);
  FSR_WAIT_VM(0);
  __syncthreads();
  acc_init();
  static_for<0, 4>([&](auto rc) {
    read_frag(rc, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  });

  int tile = tile_begin;
  bool have_prev = false;
  for (;;) {
    const bool has_nxt = tile + 1 < tile_end;
    static_for<0, 2>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if constexpr (c == 1) {
        if (has_nxt) {               // from here on the DMA feeds the next tile
          nxt = coords(tile + 1);
          halo_src(nxt, hv_nxt);
        }
      }
      static_for<0, 18>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int buf = s & 1;
        if constexpr (s == 17) {
          // publish the next chunk's halo: this wave's pieces have landed, then everybody's; the barrier also retires the reads of
          // this chunk's buffer (the last substep's fragments are in registers), which the DMA of the next chunk overwrites
          if constexpr (!(V3_ABL & 16)) FSR_WAIT_VM(0);
          FSR_BARRIER();
        }
        static_for<0, 4>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          constexpr int n = i >> 1, m = i & 1;
          if constexpr (!(V3_ABL & 4)) acc[n][m] = Mfma32v<T>::run(fa[buf][n], fb[buf][m], acc[n][m]);
          else if (a.GW < 0) acc[n][m] = Mfma32v<T>::run(fa[buf][n], fb[buf][m], acc[n][m]);
          // fragment i of the NEXT substep, into the other register set
          if constexpr (s + 1 < 18) {
            read_frag(ic, std::integral_constant<int, buf ^ 1>{}, cc, std::integral_constant<int, s + 1>{});
          } else if constexpr (c == 0) {
            read_frag(ic, std::integral_constant<int, buf ^ 1>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
          } else {
            if (has_nxt) read_frag(ic, std::integral_constant<int, buf ^ 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
          }
          // the halo pieces of the next chunk, one per substep (substeps 1 .. 5), after the substep's second MFMA
          if constexpr (i == 1 && s >= 1 && s <= V3_HPW && !(V3_ABL & 2)) {
            if constexpr (c == 0) dma_halo(hv_cur, 1, s - 1);
            else {
              if (has_nxt) dma_halo(hv_nxt, 0, s - 1);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        });
        // the previous tile's epilogue, inside this tile's first chunk: waves 0-3 early, their SIMD partners (waves 4-7) later
#ifdef FSR_V3_SPREAD
        if constexpr (c == 0 && s >= 2 && s <= 16 && (s & 1) == 0) {
          constexpr int part = ((s - 2) >> 1) & 3;
          if (have_prev && ch == (s <= 8 ? 0 : 1)) ep_part(std::integral_constant<int, part>{});
        }
#else
        if constexpr (c == 0 && (s == 3 || s == 11)) {
          if (have_prev && ch == (s == 3 ? 0 : 1) && (!(V3_ABL & 8) || a.GW < 0)) ep_finish();
        }
#endif
      });
    });
    // the tile is complete: hand the accumulators to the deferred epilogue
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < 2; ++m) accp[n][m] = acc[n][m];
    e_tc = cur;
    have_prev = true;
    if (!has_nxt) break;
    cur = nxt;
#pragma unroll
    for (int k = 0; k < V3_HPW; ++k) hv_cur[k] = hv_nxt[k];
    ++tile;
    acc_init();
  }
  ep_finish();
}

int v3_slots() {
  if (const char* e = getenv("FSR_PERSIST_CUS")) {
    const int v = atoi(e);
    if (v > 0) return v;
  }
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  return cus;
}

template <typename T, int VAR>
void v3_launch(const ConvKArgs& a, int grid, hipStream_t stream) {
  auto kern = conv64_v3_kernel<T, VAR>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), V3_LDS, stream, a);
}

}  // namespace

// 1 = launched, 0 = not this kernel's shape (the caller goes on to conv64_v2), < 0 = error.
int fsr_conv64_v3_try(int dtype, ConvKArgs& a, int S, hipStream_t stream) {
  if ((dtype != FSR_BF16 && dtype != FSR_F16) || S != 1 || a.Cin != 64 || a.ntaps != 9) return 0;
  if (const char* e = getenv("FSR_C64V3"))       // A/B switch: 0 = these layers stay on conv64_v2
    if (atoi(e) == 0) return 0;
  if (a.Cout % 64 != 0 || a.CoutPad != a.Cout || a.Cout > 256) return 0;
  if (a.in_ps || a.out_f32 || a.oscale || a.wlin || a.dmask_bits) return 0;
  if (a.ps && (a.Cout != 256 || a.stats || a.dmask)) return 0;      // PixelShuffle(2): one quadrant = one 64-row block
  if (a.preact && (a.dmask || a.stats || a.pool2)) return 0;
  if (a.pool2 && (a.ps || a.stats || a.dmask || (a.GH & 1) || (a.GW & 1))) return 0;
  if (a.stats && a.dmask) return 0;
  if (a.act != FSR_ACT_NONE && a.act != FSR_ACT_RELU && a.act != FSR_ACT_LEAKY && a.act != FSR_ACT_PRELU) return 0;
  if (a.act == FSR_ACT_PRELU && (a.stats || a.dmask)) return 0;
  if (a.osy != 1 || a.osx != 1 || a.ooy != 0 || a.oox != 0 || a.org_y != -1 || a.org_x != -1) return 0;
  if ((long long)a.N * a.IH * a.IW * 64 >= (1LL << 31) || (long long)a.N * a.FOH * a.FOW * a.Cout >= (1LL << 31)) return 0;
  // canonical tap order (ky, kx): which filter slice serves the tap that reads halo offset (ky, kx)
  int slice[9];
  for (int t = 0; t < 9; ++t) slice[t] = -1;
  for (int t = 0; t < 9; ++t) {
    if (a.tdy[t] < 0 || a.tdy[t] > 2 || a.tdx[t] < 0 || a.tdx[t] > 2) return 0;
    slice[a.tdy[t] * 3 + a.tdx[t]] = a.tw[t];
  }
  for (int t = 0; t < 9; ++t)
    if (slice[t] < 0) return 0;
  const int tiles_x = (a.GW + V3_TW - 1) / V3_TW, tiles_y = (a.GH + V3_TH - 1) / V3_TH;
  const long long ntiles = (long long)tiles_x * tiles_y * a.N;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) return 0;
  const int nblk = a.Cout / 64;
  int slots = v3_slots() / nblk;
  if (slots < 1) slots = 1;
  // Large tiles need enough of them: below two tiles per workgroup slot (batch-1 inference, the 96^2 maps of a small batch) the
  // 16 x 16 tiles of conv64_v2 fill the chip better.  FSR_C64V3=2 forces this kernel (tests).
  const char* force = getenv("FSR_C64V3");
  if (ntiles < 2LL * slots && !(force && atoi(force) == 2)) return 0;
  if (a.stats) {
    const long long P = (long long)((a.GH + 3) / 4) * ((a.GW + 15) / 16);
    if (P > a.stats_P_max) return 0;
    a.stats_P = (int)P;
    a.stats_tpi = a.stats_per = 0;
  }
  for (int t = 0; t < 9; ++t) a.t3_woff[t] = (unsigned)((size_t)slice[t] * a.CoutPad * 64 * 2);
  a.tiles_x = tiles_x;
  a.tiles_y = tiles_y;
  a.t3_ntiles = (int)ntiles;
  const int per = (int)((ntiles + slots - 1) / slots);       // contiguous tiles per workgroup
  a.nblk_n = per;
  const int ranges = (int)((ntiles + per - 1) / per);
  const int grid = ((ranges + 7) / 8) * 8 * nblk;
  const int var = a.stats ? 1 : (a.dmask ? 2 : 0);
  if (dtype == FSR_F16) {
    if (var == 1) v3_launch<f16_t, 1>(a, grid, stream);
    else if (var == 2) v3_launch<f16_t, 2>(a, grid, stream);
    else v3_launch<f16_t, 0>(a, grid, stream);
  } else {
    if (var == 1) v3_launch<bf16_t, 1>(a, grid, stream);
    else if (var == 2) v3_launch<bf16_t, 2>(a, grid, stream);
    else v3_launch<bf16_t, 0>(a, grid, stream);
  }
  fsr_note_kernel("conv64_v3_kernel<%s,%s>", dtype == FSR_F16 ? "f16" : "bf16", var == 1 ? "stats" : (var == 2 ? "mask" : "plain"));
  const int rc = fsr_check_launch("conv64_v3_kernel");
  return rc ? rc : 1;
}
