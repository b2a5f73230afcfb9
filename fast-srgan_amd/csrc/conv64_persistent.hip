// 64-input-channel 3x3 convolutions with the WHOLE filter resident in LDS: persistent kernels.
//
// The 64-channel layers -- ResidualBlock conv1/conv2 and the bottleneck of the Generator
// (/root/reference/model.py:47-64, 86-93: 17 layers forward + 17 data gradients per iteration), vgg19 conv1_2 / conv2_1
// (model.py:8), the discriminator's 64 -> 64 / 64 -> 128 blocks (model.py:148-159), the generator's two 64 -> 256
// up-sampling convolutions (model.py:30-37) and the 64 -> 3 ends (model.py:102-110) -- have only K = 9*64 = 576: in the
// generic implicit-GEMM kernel a workgroup spends more time fetching nine 8 KB filter slices from L2 (one barrier each) and
// in its prologue/epilogue than in its 288 MFMAs.  gfx950 has 160 KB of LDS per CU and the complete 16-bit filter block is
// 72 KB, so a persistent workgroup keeps it next to its halos and walks tiles:
//   conv64_v2_kernel       64-channel output blocks, stride 1: XOR-swizzled 128-byte rows, halo by LDS-DMA into two
//                          buffers, one barrier per tile, epilogue deferred into the next tile
//   conv64_thin_kernel     the 64 -> 3 ends (float / uint8 output): 9 x 16 filter rows, two workgroups per CU
//   conv64_s2fwd_kernel    64 -> 64 stride-2 forward, streaming (3.1d)
//   conv64_s2dgrad_kernel  64 -> 64 stride-2 data gradient, all four parity classes per tile (3.1c)
// Same operand mapping, tap table and epilogue semantics (bias, ReLU / LeakyReLU / identity, InstanceNorm statistics, fused
// activation-backward mask) -- and therefore the same results -- as conv_igemm_kernel.  16-bit only: the f32 filter
// (144 KB + pads) does not fit next to a halo, the parity mode stays on the generic kernel.
#include "fsr_common.h"
#include "fsr_conv_args.h"
#include "fsr_host.h"

#include <stdlib.h>

namespace {

constexpr int P64 = 80;          // LDS pitch in bf16 elements (160 B = 10 slots of 16 B: P = 2 mod 4)
constexpr int HT = 18;           // halo extent of a 16x16 tile, 3x3 footprint
constexpr int NTHR64 = 512;
constexpr int W_BYTES = 9 * 64 * P64 * 2;   // a whole 64 x 64 x 9 filter at the conflict-free 160-byte pitch (conv64_s2dgrad_kernel)
constexpr int H_BYTES = HT * HT * P64 * 2;
constexpr int SRED_BYTES = 8 * 128 * 4;   // per-wave InstanceNorm partial sums
constexpr int LDS64 = W_BYTES + H_BYTES + SRED_BYTES;
constexpr size_t LDS_THIN = 9 * 16 * P64 * 2 + H_BYTES + 16;   // conv64_thin_kernel: 9 x 16 filter rows + one halo

__device__ __forceinline__ unsigned tap64(const ConvKArgs& a, int t) {
  return (t < 8) ? (unsigned)((a.taps_lo >> (8 * t)) & 0xffull) : a.taps_hi;
}

// The 64 -> 3 ends (a float or uint8 output of at most 16 channels: the head with its tanh, model.py:102-110, and the image
// gradients with their per-channel scale): the filter is 9 x 16 rows, two workgroups fit a CU, and the kernel streams the
// 64-channel input at memory speed instead of living one tile per workgroup.  (This register-staged, two-barriers-per-tile
// structure was also the FIRST form of the 64-channel-block kernel; conv64_v2_kernel below replaced it in round 2 and the
// block instantiation was deleted in round 4.)
// X3 (round 6; T = bf16_t): the input is an x3 tensor -- per pixel and 32-channel group 64 bytes of hi, then 64 bytes of lo, i.e. 128
// bf16 elements per pixel of which group g owns [64 g, 64 g + 64) -- and the filter the x3 pack ([w_hi | w_lo] per 32 input channels).
// A tile is then two passes of this same kernel body, one per channel group: the group's 128 bytes per pixel ARE a 64-channel 16-bit
// pixel whose first half is x_hi and whose second half is x_lo, and the three products x_hi w_hi + x_lo w_hi + x_hi w_lo are three MFMAs
// on the four fragments the 16-bit form reads for its two.  Only eight filter rows are kept in LDS per (group, tap) (Cout <= 8: the
// lanes of rows 8 .. 15 read rows 0 .. 7 again and their accumulator rows are never stored), so the filter block is the 16-bit form's
// size and two workgroups still share a CU.  (Until round 6 these launches ran on conv_igemm<x3,8,16,...> at 2.3 TB/s.)
template <typename T, bool X3 = false>
__global__ __launch_bounds__(NTHR64, 2) void conv64_thin_kernel(const ConvKArgs a) {
  static_assert(!X3 || std::is_same<T, bf16_t>::value, "x3: bf16 planes");
  constexpr int NT = 1;
  constexpr int NG = X3 ? 2 : 1;                        // channel-group passes per tile
  constexpr int ROWS = X3 ? 8 : NT * 16;                // filter rows kept per (group, tap)
  constexpr int PIX = X3 ? 128 : 64;                    // T elements per input pixel
  constexpr int HUNITS = HT * HT * 8;                   // 16-byte units of one halo (one group: 128 bytes per pixel)
  constexpr int HPT = (HUNITS + NTHR64 - 1) / NTHR64;   // 6
  HIP_DYNAMIC_SHARED(char, smem)
  T* wl = (T*)smem;
  constexpr int WB = 9 * NT * 16 * P64 * 2;            // bytes of the resident filter block (x3: 2 groups x 9 taps x 8 rows)
  static_assert(NG * 9 * ROWS * P64 * 2 == WB, "the filter block of every form is 9 x 16 rows of the 160-byte pitch");
  T* halo = (T*)(smem + WB);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const T* in = (const T*)a.in;
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int ntiles = tiles_per_img * a.N;

  // ---- the filter rows, all nine taps (x3: of both channel groups): [g][9][ROWS][64] -> LDS, once
  {
    const T* wpk = (const T*)a.wpk;
    constexpr int WU = NG * 9 * ROWS * 8;               // 16-byte units of the resident filter block
#pragma unroll
    for (int i = 0; i < (WU + NTHR64 - 1) / NTHR64; ++i) {
      const int u = tid + i * NTHR64;                   // ((g*9 + slice)*ROWS + row)*8 + unit
      if (WU % NTHR64 == 0 || u < WU) {
        const int gs = u / (ROWS * 8), ru = u % (ROWS * 8);
        const int g = gs / 9, slice = gs - 9 * g, row = ru >> 3, unit = ru & 7;
        const u32x4 v = *(const u32x4*)(wpk + ((size_t)(slice * a.CoutPad + row) * PIX + (size_t)(g * 64 + unit * 8)));
        *(u32x4*)(wl + (u >> 3) * P64 + (u & 7) * 8) = v;
      }
    }
  }

  u32x4 hreg[HPT];
  // halo of step s = (tile s / NG, channel group s % NG)
  auto halo_issue = [&](int step) {
    const int tile = step / NG, g = step - tile * NG;
    const int img = tile / tiles_per_img;
    const int rem = tile - img * tiles_per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int iy0 = ty * 16 + a.org_y, ix0 = tx * 16 + a.org_x;
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
      const int u = tid + i * NTHR64;
      u32x4 v = (u32x4){0u, 0u, 0u, 0u};
      if (u < HUNITS) {
        const int unit = u & 7, p = u >> 3;
        const int hy = p / HT, hx = p - hy * HT;
        const int iy = iy0 + hy, ix = ix0 + hx;
        if (iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW)
          v = *(const u32x4*)(in + ((unsigned)((img * a.IH + iy) * a.IW + ix) * (unsigned)PIX + (unsigned)(g * 64 + unit * 8)));
      }
      hreg[i] = v;
    }
  };
  auto halo_commit = [&]() {
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
      const int u = tid + i * NTHR64;
      if (u < HUNITS) *(u32x4*)(halo + (u >> 3) * P64 + (u & 7) * 8) = hreg[i];
    }
  };

  float slope = (a.act == FSR_ACT_PRELU) ? a.prelu[0] : a.slope;
  if (a.act == FSR_ACT_NONE || a.act == FSR_ACT_TANH) slope = 1.f;
  if (a.act == FSR_ACT_RELU) slope = 0.f;

  int pixbase[2], wbase[NT];
#pragma unroll
  for (int m = 0; m < 2; ++m) pixbase[m] = ((wave * 2 + m) * HT + l15) * P64 + lg * 8;
#pragma unroll
  for (int n = 0; n < NT; ++n) wbase[n] = (n * 16 + (X3 ? (l15 & 7) : l15)) * P64 + lg * 8;
  f32x4 bias[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    bias[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bias[n][r] = a.bias[lg * 4 + r < a.Cout ? lg * 4 + r : 0];
    }
  }

  // a workgroup walks a CONTIGUOUS range of tiles
  const int tile_begin = (int)blockIdx.x * a.nblk_n;   // nblk_n = tiles per workgroup (set by the host)
  const int tile_end = (tile_begin + a.nblk_n < ntiles) ? tile_begin + a.nblk_n : ntiles;
  f32x4 oscale = (f32x4){1.f, 1.f, 1.f, 1.f};     // optional per-channel scale (VGG normalisation in the image gradient)
  if (a.oscale) {
#pragma unroll
    for (int r = 0; r < 4; ++r) oscale[r] = a.oscale[lg * 4 + r < a.Cout ? lg * 4 + r : 0];
  }
  const int step_begin = tile_begin * NG, step_end = tile_end * NG;
  int step = step_begin;
  if (step < step_end) halo_issue(step);
  __syncthreads();                 // filter visible
  if (step < step_end) halo_commit();
  __syncthreads();

  f32x4 acc[2][NT];
  for (; step < step_end; ++step) {
    const int next = step + 1;
    const int tile = step / NG, g = step - tile * NG;
    if (next < step_end) halo_issue(next);

    if (g == 0) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const unsigned tc = tap64(a, t);
      const int toff = ((int)(tc & 3u) * HT + (int)((tc >> 2) & 3u)) * P64;
      const T* wsl = wl + ((g * 9 + (int)(tc >> 4)) * ROWS) * P64;
      if constexpr (X3) {
        // the group's pixel = [x_hi 32 | x_lo 32], its filter row = [w_hi 32 | w_lo 32]: hi hi + lo hi + hi lo
        s16x8 wf[2], xf[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          wf[ks] = *(const s16x8*)(wsl + wbase[0] + ks * 32);
#pragma unroll
          for (int m = 0; m < 2; ++m) xf[m][ks] = *(const s16x8*)(halo + pixbase[m] + toff + ks * 32);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          acc[m][0] = mfma16<T>(wf[0], xf[m][0], acc[m][0]);
          acc[m][0] = mfma16<T>(wf[0], xf[m][1], acc[m][0]);
          acc[m][0] = mfma16<T>(wf[1], xf[m][0], acc[m][0]);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s16x8 wf[NT], xf[2];
#pragma unroll
          for (int n = 0; n < NT; ++n) wf[n] = *(const s16x8*)(wsl + wbase[n] + ks * 32);
#pragma unroll
          for (int m = 0; m < 2; ++m) xf[m] = *(const s16x8*)(halo + pixbase[m] + toff + ks * 32);
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = mfma16<T>(wf[n], xf[m], acc[m][n]);
        }
      }
    }

    // ---- epilogue of this tile (after its last channel group)
    FSR_WAIT_LOADS();   // the prefetched halo has landed; the stores below then drain under the next tile
    if (g == NG - 1) {
      const int img = tile / tiles_per_img;
      const int rem = tile - img * tiles_per_img;
      const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
      const int gx = tx * 16 + l15, gyb = ty * 16 + wave * 2;
      const bool col_ok = gx < a.GW;
      // float output, Cout <= 16 valid channels (lane group lg holds channels 4 lg .. 4 lg + 3): scale, bias, tanh / slope
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (col_ok && gyb + m < a.GH) {
          const size_t off = (((size_t)img * a.FOH + gyb + m) * a.FOW + gx) * a.Cout + lg * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (lg * 4 + r < a.Cout) {
              float v = acc[m][0][r] * oscale[r] + bias[0][r];
              v = (a.act == FSR_ACT_TANH) ? tanhf(v) : (v > 0.f ? v : v * slope);
              if (a.out_f32 == FSR_OUT_U8) ((unsigned char*)a.out)[off + r] = image_u8(v);   // HWC uint8 image (inference)
              else ((float*)a.out)[off + r] = v;
            }
        }
      }
    }
    __syncthreads();   // every wave is done with the halo
    if (next < step_end) halo_commit();
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The 64-channel-block kernel (second form; the thin kernel above keeps the structure of the first, whose small LDS
// footprint admits two workgroups per CU).  Measured on the first form at 96^2 (tools/conv_bench.py, batch 32 .. 256): a
// workgroup needs 6.5 us per tile, of which the 288 MFMAs per SIMD are 1.9 us -- prefetch, MFMAs, epilogue, barrier,
// LDS commit, barrier run one after the other and all eight waves walk them in lock step.  This form takes the phases
// apart:
//   * LDS rows are 128 B with an XOR swizzle (16-byte unit u of row r sits at u ^ (r & 7)) instead of a 160 B pitch:
//     filter 73,728 B + TWO halo buffers of 328 pixels (41,984 B each) + statistics 4 KB = 161,792 B.
//   * the halo of the next tile goes global -> LDS by DMA (global_load_lds_dwordx4, swizzle on the source address,
//     padding from a zero page) into the other buffer: no staging registers, no commit pass, ONE barrier per tile.
//   * the epilogue of tile i runs inside tile i + 1: waves 0-3 after the second tap, waves 4-7 (their SIMD partners)
//     after the sixth, so one wave of a SIMD stores while the other owns the MFMA pipe.  Waves 0-3 also issue the DMA
//     (after their epilogue: the mask loads of a fused activation gradient are then not queued behind it).
// Results are those of the first form (same operand mapping, same summation order).
#ifndef FSR_ABL64
#define FSR_ABL64 0     // ablation builds (tools/build_variant.sh): 1 no stores, 2 no halo DMA after the first tile, 3 both, 4 no MFMAs
#endif
constexpr int HPIX2 = 328;                    // 18 x 18 halo pixels, rounded up to whole 8-pixel DMA instructions
constexpr int NDMA2 = HPIX2 / 8;
constexpr int W2_BYTES = 9 * 64 * 64 * 2;
constexpr int H2_BYTES = HPIX2 * 64 * 2;
constexpr int LDS64V2 = W2_BYTES + 2 * H2_BYTES + SRED_BYTES;

__device__ unsigned conv64_zero_page[64];     // padding source of the halo DMA

// STATS / MASK: InstanceNorm statistics / fused activation-gradient mask compiled in (never both: see the launcher);
// the variants without them free the 32 / 16 registers those carry across the MFMA phase.
template <typename T, bool STATS, bool MASK>
__global__ __launch_bounds__(NTHR64) void conv64_v2_kernel(const ConvKArgs a) {
  constexpr int NT = 4;
  HIP_DYNAMIC_SHARED(char, smem)
  T* wl = (T*)smem;
  T* halo0 = (T*)(smem + W2_BYTES);
  float* sred = (float*)(smem + W2_BYTES + 2 * H2_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const int l8 = lane >> 3, s8 = lane & 7;
  const T* in = (const T*)a.in;
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int ntiles = tiles_per_img * a.N;
  const int nblk = a.CoutPad >> 6;
  const int nb = (int)blockIdx.x % nblk;
  const T* zero = (const T*)conv64_zero_page;
  const fsr_lds_addr_t lds_addr = FSR_LDS_ADDR(smem);      // LDS byte address of the filter; halos follow at W2_BYTES

  // ---- filter block nb, all nine taps: LDS row r = slice * 64 + row; 72 wave instructions of 8 rows
  {
    const T* wpk = (const T*)a.wpk + (size_t)nb * 64 * 64;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int k = wave + i * 8;
      const int r = k * 8 + l8;
      const int slice = r >> 6, j = r & 63;
      const int row = ((j & 12) << 2) + ((j >> 4) << 2) + (j & 3);   // MFMA row i of tile n <- channel (i >> 2) * 16 + n * 4 + (i & 3)
      FSR_GLDS16_AT(wpk + ((size_t)slice * a.CoutPad * 64 + (size_t)(row * 64 + ((s8 ^ (r & 7)) * 8))), lds_addr + (fsr_lds_addr_t)(k * 8 * 64) * sizeof(T));
    }
  }
  // halo of `tile` -> buffer `buf`: wave instruction k fills pixels 8k .. 8k+7 (this wave: k = wid, wid + nw, ...)
  auto halo_dma = [&](int tile, int buf, int wid, int nw) {
    const int img = tile / tiles_per_img;
    const int rem = tile - img * tiles_per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int iy0 = ty * 16 + a.org_y, ix0 = tx * 16 + a.org_x;
    const fsr_lds_addr_t hb = lds_addr + W2_BYTES + (fsr_lds_addr_t)buf * H2_BYTES;
    for (int k = wid; k < NDMA2; k += nw) {
      const int p = k * 8 + l8;
      const int hy = p / HT, hx = p - hy * HT;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = p < HT * HT && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
      const T* src = ok ? in + ((unsigned)((img * a.IH + iy) * a.IW + ix) * 64u + (unsigned)((s8 ^ (p & 7)) * 8)) : zero + s8 * 8;
      FSR_GLDS16_AT(src, hb + (fsr_lds_addr_t)(k * 8 * 64) * sizeof(T));
    }
  };

  float slope = (a.act == FSR_ACT_PRELU) ? a.prelu[0] : a.slope;
  if (a.act == FSR_ACT_NONE || a.act == FSR_ACT_TANH) slope = 1.f;
  if (a.act == FSR_ACT_RELU) slope = 0.f;
  constexpr bool want_stats = STATS;
  const bool ps = !STATS && !MASK && a.ps;          // PixelShuffle / fused pool: plain variant only (launcher)
  const bool pool2 = !STATS && !MASK && a.pool2;
  T* outp = (T*)a.out;
  T* prep = (T*)a.preact;
  const T* maskp = MASK ? (const T*)a.dmask : nullptr;

  int p0[2], wofs[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) p0[m] = (wave * 2 + m) * HT + l15;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) wofs[ks] = l15 * 64 + (((ks * 4 + lg) ^ (l15 & 7)) * 8);
  f32x4 bias[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    bias[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
      if (!ps) bias[n] = *(const f32x4*)(a.bias + nb * 64 + lg * 16 + n * 4);
      else
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[n][r] = a.bias[4 * (lg * 16 + n * 4 + r) + nb];   // torch order 4*cc + quadrant
    }
  }
  const int tile_begin = ((int)blockIdx.x / nblk) * a.nblk_n;
  const int tile_end = (tile_begin + a.nblk_n < ntiles) ? tile_begin + a.nblk_n : ntiles;
  f32x4 s1acc[STATS ? NT : 1], s2acc[STATS ? NT : 1];
#pragma unroll
  for (int n = 0; n < (STATS ? NT : 1); ++n) s1acc[n] = s2acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (tile_begin < tile_end) halo_dma(tile_begin, 0, wave, 8);
  FSR_WAIT_DMA();
  __syncthreads();

  f32x4 acc[2][NT], accp[2][NT];
  const T* hb = halo0;
  // One step = (tap, 32-channel half): 6 fragment reads, 8 MFMAs.  Inside a segment of taps the reads of step s + 1 are
  // issued before the MFMAs of step s (two fragment sets): a wave then keeps the matrix pipe busy on its own and the
  // DMA / epilogue issue time of its SIMD partner is covered instead of added.
  auto load_step = [&](auto sc, s16x8 (&wf)[NT], s16x8 (&xf)[2]) {
    constexpr int st = decltype(sc)::value;
    constexpr int t = st >> 1, ks = st & 1;
    const unsigned tc = tap64(a, t);
    const int shift = (int)(tc & 3u) * HT + (int)((tc >> 2) & 3u);
    const T* wsl = wl + (int)(tc >> 4) * (64 * 64);
#pragma unroll
    for (int n = 0; n < NT; ++n) wf[n] = *(const s16x8*)(wsl + wofs[ks] + n * (16 * 64));
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int p = p0[m] + shift;
      const int u = (lg + ks * 4) ^ (p & 7);
      xf[m] = *(const s16x8*)(hb + (p * 64 + u * 8));
    }
  };
  auto run_taps = [&](auto t0c, auto t1c) {
#if FSR_ABL64 == 4
    if (a.GW >= 0) return;
#endif
    constexpr int S0 = decltype(t0c)::value * 2, S1 = decltype(t1c)::value * 2;
    s16x8 wfa[NT], xfa[2], wfb[NT], xfb[2];
    load_step(std::integral_constant<int, S0>{}, wfa, xfa);
    static_for<S0, S1>([&](auto sc) {
      constexpr int st = decltype(sc)::value;
      constexpr bool even = ((st - S0) & 1) == 0;
      if constexpr (st + 1 < S1) {
        if constexpr (even) load_step(std::integral_constant<int, st + 1>{}, wfb, xfb);
        else load_step(std::integral_constant<int, st + 1>{}, wfa, xfa);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = even ? mfma16<T>(wfa[n], xfa[m], acc[m][n]) : mfma16<T>(wfb[n], xfb[m], acc[m][n]);
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- epilogue of the PREVIOUS tile, in two parts: coordinates + mask loads at the top of an iteration, the rest later
  int e_img = 0, e_gx = 0, e_gyb = 0;
  bool e_col_ok = false, e_flush = false;
  unsigned e_rstride = 0, e_base0 = 0;
  u32x4 mkv[MASK ? 2 : 1][2];     // the lane's 16 mask channels of a row: two 16-byte loads
  auto ep_prepare = [&](int tile) {
    e_img = tile / tiles_per_img;
    const int rem = tile - e_img * tiles_per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    e_gx = tx * 16 + l15;
    e_gyb = ty * 16 + wave * 2;
    e_col_ok = e_gx < a.GW;
    e_rstride = ps ? (unsigned)(4 * a.FOW * 64) : (unsigned)(a.FOW * a.Cout);
    e_base0 = ps ? (unsigned)((e_img * 2 * a.FOH + 2 * e_gyb + (nb >> 1)) * (2 * a.FOW) + 2 * e_gx + (nb & 1)) * 64u + (unsigned)(lg * 16)
                   : (unsigned)((e_img * a.FOH + e_gyb) * a.FOW + e_gx) * (unsigned)a.Cout + (unsigned)(nb * 64 + lg * 16);
    e_flush = want_stats && (tile + 1 >= tile_end || (tile + 1) / tiles_per_img != e_img);
  };
  // mask vectors of the CURRENT tile (plain layout only), loaded late in its MFMA phase and used by its epilogue in the
  // next iteration: they land before the iteration's closing s_waitcnt vmcnt(0), so the halo DMA issued at the top of the
  // next iteration is never waited for by an epilogue
  auto mask_issue = [&](int tile) {
    const int img = tile / tiles_per_img;
    const int rem = tile - img * tiles_per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int gx = tx * 16 + l15, gyb = ty * 16 + wave * 2;
    const unsigned rstride = (unsigned)(a.FOW * a.Cout);
    const unsigned base0 = (unsigned)((img * a.FOH + gyb) * a.FOW + gx) * (unsigned)a.Cout + (unsigned)(nb * 64 + lg * 16);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        mkv[m][h] = (gx < a.GW && gyb + m < a.GH) ? *(const u32x4*)(maskp + (base0 + m * rstride + h * 8)) : (u32x4){0u, 0u, 0u, 0u};
  };
  auto ep_finish = [&]() {
    const int img = e_img, gx = e_gx, gyb = e_gyb;
    const bool col_ok = e_col_ok;
    if (pool2) {
      // MaxPool2d(2,2) fused (no-grad vgg19 passes): the wave's two rows and the columns (l15, l15 ^ 1) of neighbouring
      // lanes; even lanes store pixel (gy / 2, gx / 2) of the [N][FOH/2][FOW/2][Cout] tensor
      u32x4 pk[2];
      static_for<0, NT>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        f32x4 v = accp[0][n] + bias[n], w = accp[1][n] + bias[n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = fmaxf(v[r], w[r]);
          x = fmaxf(x, __shfl_xor(x, 1, 64));
          v[r] = fmaxf(x, 0.f) + slope * fminf(x, 0.f);
        }
        pk[n >> 1][(n & 1) * 2] = pack2<T>(v[0], v[1]);
        pk[n >> 1][(n & 1) * 2 + 1] = pack2<T>(v[2], v[3]);
      });
      if (col_ok && gyb < a.GH && !(l15 & 1)) {
        const unsigned off = (unsigned)((img * (a.FOH >> 1) + (gyb >> 1)) * (a.FOW >> 1) + (gx >> 1)) * (unsigned)a.Cout + (unsigned)(nb * 64 + lg * 16);
        fsr_st<2>((u32x4*)(outp + off), (u32x4)(pk[0]));
        fsr_st<2>((u32x4*)(outp + off + 8), (u32x4)(pk[1]));
      }
      return;
    }
    // the lane holds channels lg * 16 .. lg * 16 + 15 of its pixel (tile n: + 4 n): 32 contiguous bytes per row, two
    // 16-byte stores (8-byte stores are issue-bound: ~7 B / clk / CU)
    static_for<0, 2>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      if (col_ok && gyb + m < a.GH) {
        const unsigned off = e_base0 + m * e_rstride;
        u32x4 pk[2], pp[2];
        static_for<0, NT>([&](auto nc) {
          constexpr int n = decltype(nc)::value;
          f32x4 v = accp[m][n] + bias[n];
          if constexpr (MASK) {
            const unsigned t0 = mkv[m][n >> 1][(n & 1) * 2], t1 = mkv[m][n >> 1][(n & 1) * 2 + 1];
            const float mk[4] = {cvt_lo<T>(t0), cvt_hi<T>(t0), cvt_lo<T>(t1), cvt_hi<T>(t1)};
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = a.dmask_add ? v[r] + mk[r] : (mk[r] > 0.f ? v[r] : v[r] * a.dmask_slope);
          }
          if constexpr (STATS) {
            s1acc[n] += v;
            s2acc[n] += v * v;
          }
          if (prep) {
            pp[n >> 1][(n & 1) * 2] = pack2<T>(v[0], v[1]);
            pp[n >> 1][(n & 1) * 2 + 1] = pack2<T>(v[2], v[3]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = act_slope(v[r], slope);       // (NaN-propagating: fsr_common.h)
          pk[n >> 1][(n & 1) * 2] = pack2<T>(v[0], v[1]);
          pk[n >> 1][(n & 1) * 2 + 1] = pack2<T>(v[2], v[3]);
        });
        if (prep) {
          fsr_st<2>((u32x4*)(prep + off), (u32x4)(pp[0]));
          fsr_st<2>((u32x4*)(prep + off + 8), (u32x4)(pp[1]));
        }
#if FSR_ABL64 == 1 || FSR_ABL64 == 3
        if (a.GW < 0)
#endif
        {
          fsr_st<2>((u32x4*)(outp + off), (u32x4)(pk[0]));
          fsr_st<2>((u32x4*)(outp + off + 8), (u32x4)(pk[1]));
        }
      }
    });
    if constexpr (STATS) if (e_flush) {
      static_for<0, NT>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x1 = s1acc[n][r], x2 = s2acc[n][r];
#pragma unroll
          for (int o = 8; o >= 1; o >>= 1) {
            x1 += __shfl_xor(x1, o, 64);
            x2 += __shfl_xor(x2, o, 64);
          }
          if (l15 == 0) {
            const int cl = lg * 16 + n * 4 + r;
            sred[(wave * 64 + cl) * 2] = x1;
            sred[(wave * 64 + cl) * 2 + 1] = x2;
          }
        }
        s1acc[n] = s2acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
      });
    }
  };
  // the eight waves' statistics of image `img`, added in order -> this workgroup's slot (after the barrier that follows
  // the flushing epilogue)
  int pend_img = -1;
  auto stats_store = [&]() {
    if (tid < 128) {
      float sum = sred[tid];
#pragma unroll
      for (int w = 1; w < 8; ++w) sum += sred[w * 128 + tid];
      const int slot = (int)blockIdx.x / nblk - (pend_img * tiles_per_img) / a.nblk_n;
      a.stats[(((size_t)pend_img * a.stats_P + slot) * a.Cout + nb * 64 + (tid >> 1)) * 2 + (tid & 1)] = sum;
    }
  };

  int buf = 0;
  for (int tile = tile_begin; tile <= tile_end && tile_begin < tile_end; ++tile) {
    const bool have = tile < tile_end, hasprev = tile > tile_begin;
    const int next = tile + 1;
    if (pend_img >= 0) {          // uniform: a flush happened in the previous iteration
      stats_store();
      __syncthreads();            // sred may be rewritten by this iteration's epilogue
      pend_img = -1;
    }
    // the whole iteration is the DMA's window: nothing this wave loads later is queued behind it (see mask_issue)
#if FSR_ABL64 == 2 || FSR_ABL64 == 3
    if (wave < 4 && have && next < tile_end && a.GW < 0) halo_dma(next, buf ^ 1, wave, 4);
#else
    if (wave < 4 && have && next < tile_end) halo_dma(next, buf ^ 1, wave, 4);
#endif
    if (hasprev) ep_prepare(tile - 1);
    hb = halo0 + buf * (HPIX2 * 64);
    if (have) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
      run_taps(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
    }
    if (wave < 4 && hasprev) ep_finish();
    if (have) run_taps(std::integral_constant<int, 2>{}, std::integral_constant<int, 6>{});
    if (wave >= 4 && hasprev) ep_finish();
    if constexpr (MASK) {
      if (have) mask_issue(tile);
    }
    if (have) {
      run_taps(std::integral_constant<int, 6>{}, std::integral_constant<int, 9>{});
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) accp[m][n] = acc[m][n];
    }
    if (hasprev && e_flush) pend_img = e_img;
    FSR_WAIT_DMA();
    if constexpr (MASK) {
      // the compiler's own wait for the mask loads goes here (first use), where it is already satisfied
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) FSR_TOUCH(mkv[m][h]);
    }
    __syncthreads();
    buf ^= 1;
  }
  if (pend_img >= 0) stats_store();
}

// ---------------------------------------------------------------------------------------------------------------
// FORWARD of the 64 -> 64 STRIDE-2 convolution (Discriminator block 0, /root/reference/model.py:148-152: 384^2 -> 192^2, the
// largest activation of the network).  In the generic kernel a workgroup lives for two 36 KB halo chunks and 144 MFMAs per wave:
// 9216 workgroups of ~17 us each move 755 MB in 308 us (2.45 TB/s) -- the layer is HBM-bound (87 GFLOP are 35 us of MFMA) and
// latency-bound per workgroup.  Here a persistent workgroup keeps the whole filter in LDS (as conv64_v2_kernel) and streams
// 8 x 16-pixel output tiles: the 17 x 33-pixel input halo arrives by LDS-DMA in two 32-channel chunks (64-byte pixels, unit
// swizzled by (pixel >> 2) & 3: the stride-2 fragment read is then 2-way conflicted, the minimum for 64-byte pixels), each
// chunk in its own buffer, the DMA of the next half step always in flight under the MFMAs and the epilogue of the current one.
// Wave w = output row w of the tile (16 pixels x 64 channels); epilogue and statistics as in conv64_v2_kernel.
constexpr int S2F_HH = 17, S2F_HW = 33;
constexpr int S2F_NDMA = (S2F_HH * S2F_HW + 15) / 16;          // wave instructions per chunk (16 pixels of 64 bytes each)
constexpr int S2F_CHUNK_BYTES = S2F_NDMA * 1024;
constexpr int LDS64S2F = W2_BYTES + 2 * S2F_CHUNK_BYTES + SRED_BYTES;

template <typename T, bool STATS>
__global__ __launch_bounds__(NTHR64) void conv64_s2fwd_kernel(const ConvKArgs a) {
  constexpr int NT = 4;
  HIP_DYNAMIC_SHARED(char, smem)
  T* wl = (T*)smem;
  T* chunk0 = (T*)(smem + W2_BYTES);
  float* sred = (float*)(smem + W2_BYTES + 2 * S2F_CHUNK_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const int l8 = lane >> 3, s8 = lane & 7;
  const int l4 = lane >> 2, s4 = lane & 3;
  const T* in = (const T*)a.in;
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int ntiles = tiles_per_img * a.N;
  const T* zero = (const T*)conv64_zero_page;
  const fsr_lds_addr_t lds_addr = FSR_LDS_ADDR(smem);

  // ---- the whole filter: LDS row r = slice * 64 + j, MFMA row permutation as in conv64_v2_kernel
  {
    const T* wpk = (const T*)a.wpk;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int k = wave + i * 8;
      const int r = k * 8 + l8;
      const int slice = r >> 6, j = r & 63;
      const int row = ((j & 12) << 2) + ((j >> 4) << 2) + (j & 3);
      FSR_GLDS16_AT(wpk + ((size_t)slice * 64 * 64 + (size_t)(row * 64 + ((s8 ^ (r & 7)) * 8))), lds_addr + (fsr_lds_addr_t)(k * 8 * 64) * sizeof(T));
    }
  }
  // 32-channel chunk c of the halo of `tile` -> buffer c: wave instruction k fills pixels 16k .. 16k+15 (4 units each)
  auto chunk_dma = [&](int tile, int c) {
    const int img = tile / tiles_per_img;
    const int rem = tile - img * tiles_per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int iy0 = ty * 16 + a.org_y, ix0 = tx * 32 + a.org_x;
    const fsr_lds_addr_t cb = lds_addr + W2_BYTES + (fsr_lds_addr_t)c * S2F_CHUNK_BYTES;
    for (int k = wave; k < S2F_NDMA; k += 8) {
      const int p = k * 16 + l4;
      const int hy = p / S2F_HW, hx = p - hy * S2F_HW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = p < S2F_HH * S2F_HW && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
      const T* src = ok ? in + ((unsigned)((img * a.IH + iy) * a.IW + ix) * 64u + (unsigned)(c * 32 + ((s4 ^ ((p >> 2) & 3)) * 8))) : zero + s4 * 8;
      FSR_GLDS16_AT(src, cb + (fsr_lds_addr_t)(k * 1024));
    }
  };

  float slope = (a.act == FSR_ACT_PRELU) ? a.prelu[0] : a.slope;
  if (a.act == FSR_ACT_NONE || a.act == FSR_ACT_TANH) slope = 1.f;
  if (a.act == FSR_ACT_RELU) slope = 0.f;
  T* outp = (T*)a.out;
  T* prep = (T*)a.preact;

  int wofs[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) wofs[ks] = l15 * 64 + (((ks * 4 + lg) ^ (l15 & 7)) * 8);
  f32x4 bias[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    bias[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias[n] = *(const f32x4*)(a.bias + lg * 16 + n * 4);
  }
  const int tile_begin = (int)blockIdx.x * a.nblk_n;
  const int tile_end = (tile_begin + a.nblk_n < ntiles) ? tile_begin + a.nblk_n : ntiles;
  f32x4 s1acc[STATS ? NT : 1], s2acc[STATS ? NT : 1];
#pragma unroll
  for (int n = 0; n < (STATS ? NT : 1); ++n) s1acc[n] = s2acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (tile_begin < tile_end) chunk_dma(tile_begin, 0);
  FSR_WAIT_DMA();
  __syncthreads();

  int pend_img = -1;
  auto stats_store = [&]() {
    if (tid < 128) {
      float sum = sred[tid];
#pragma unroll
      for (int w = 1; w < 8; ++w) sum += sred[w * 128 + tid];
      const int slot = (int)blockIdx.x - (pend_img * tiles_per_img) / a.nblk_n;
      a.stats[(((size_t)pend_img * a.stats_P + slot) * a.Cout + (tid >> 1)) * 2 + (tid & 1)] = sum;
    }
  };

  f32x4 acc[NT];
  for (int tile = tile_begin; tile < tile_end; ++tile) {
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      // the next half step's chunk: the other buffer, last read one half step (and one barrier) ago
      if (c == 0) chunk_dma(tile, 1);
      else if (tile + 1 < tile_end) chunk_dma(tile + 1, 0);
      if (c == 0 && pend_img >= 0) {       // uniform: the previous tile flushed statistics
        stats_store();
        __syncthreads();
        pend_img = -1;
      }
      const T* hb = chunk0 + c * (S2F_CHUNK_BYTES / (int)sizeof(T));
      static_for<0, 9>([&](auto tcn) {
        constexpr int t = decltype(tcn)::value;
        constexpr int ky = t / 3, kx = t % 3;
        const int p = (2 * wave + ky) * S2F_HW + kx + 2 * l15;
        const s16x8 xf = *(const s16x8*)(hb + (p * 32 + ((lg ^ ((p >> 2) & 3)) * 8)));
        const T* wsl = wl + t * (64 * 64);
        s16x8 wf[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) wf[n] = *(const s16x8*)(wsl + wofs[c] + n * (16 * 64));
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = mfma16<T>(wf[n], xf, acc[n]);
      });
      if (c == 1) {
        // ---- epilogue: the lane holds channels lg * 16 .. lg * 16 + 15 of pixel (row `wave`, column l15): two 16-byte stores
        const int img = tile / tiles_per_img;
        const int rem = tile - img * tiles_per_img;
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int gx = tx * 16 + l15, gy = ty * 8 + wave;
        const bool flush = STATS && (tile + 1 >= tile_end || (tile + 1) / tiles_per_img != img);
        if (gx < a.GW && gy < a.GH) {
          const unsigned off = (unsigned)((img * a.FOH + gy) * a.FOW + gx) * 64u + (unsigned)(lg * 16);
          u32x4 pk[2], pp[2];
          static_for<0, NT>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            f32x4 v = acc[n] + bias[n];
            if constexpr (STATS) {
              s1acc[n] += v;
              s2acc[n] += v * v;
            }
            if (prep) {
              pp[n >> 1][(n & 1) * 2] = pack2<T>(v[0], v[1]);
              pp[n >> 1][(n & 1) * 2 + 1] = pack2<T>(v[2], v[3]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f) + slope * fminf(v[r], 0.f);
            pk[n >> 1][(n & 1) * 2] = pack2<T>(v[0], v[1]);
            pk[n >> 1][(n & 1) * 2 + 1] = pack2<T>(v[2], v[3]);
          });
          if (prep) {
            fsr_st<2>((u32x4*)(prep + off), (u32x4)(pp[0]));
            fsr_st<2>((u32x4*)(prep + off + 8), (u32x4)(pp[1]));
          }
          fsr_st<2>((u32x4*)(outp + off), (u32x4)(pk[0]));
          fsr_st<2>((u32x4*)(outp + off + 8), (u32x4)(pk[1]));
        }
        if constexpr (STATS) {
          if (flush) {
            static_for<0, NT>([&](auto nc) {
              constexpr int n = decltype(nc)::value;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float x1 = s1acc[n][r], x2 = s2acc[n][r];
#pragma unroll
                for (int o = 8; o >= 1; o >>= 1) {
                  x1 += __shfl_xor(x1, o, 64);
                  x2 += __shfl_xor(x2, o, 64);
                }
                if (l15 == 0) {
                  const int cl = lg * 16 + n * 4 + r;
                  sred[(wave * 64 + cl) * 2] = x1;
                  sred[(wave * 64 + cl) * 2 + 1] = x2;
                }
              }
              s1acc[n] = s2acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            });
            pend_img = img;
          }
        }
      }
      FSR_WAIT_DMA();
      __syncthreads();
    }
  }
  if (pend_img >= 0) stats_store();
}

}  // namespace

// Workgroup slots of the persistent kernels: one per CU.  FSR_PERSIST_CUS overrides the device's CU count (tests: lets a
// small problem exercise tile ranges that straddle image borders on any device).
static int persistent_slots() {
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

// Stride-2 forward, 64 -> 64 channels: 1 = launched, 0 = not this kernel's shape, < 0 = error.
int fsr_conv64_s2fwd_try(int dtype, ConvKArgs& a, int S, hipStream_t stream) {
  static const bool off = getenv("FSR_CONV64_S2FWD") && atoi(getenv("FSR_CONV64_S2FWD")) == 0;   // A/B switch
  if (off || (dtype != FSR_BF16 && dtype != FSR_F16) || S != 2 || a.Cin != 64 || a.Cout != 64 || a.CoutPad != 64 || a.ntaps != 9) return 0;
  if (a.ps || a.in_ps || a.out_f32 || a.oscale || a.dmask || a.pool2) return 0;
  if (a.osy != 1 || a.osx != 1 || a.ooy != 0 || a.oox != 0 || a.org_y != -1 || a.org_x != -1) return 0;
  for (int t = 0; t < 9; ++t)
    if (a.tdy[t] != t / 3 || a.tdx[t] != t % 3 || a.tw[t] != t) return 0;     // the forward tap table
  if ((long long)a.N * a.IH * a.IW * 64 >= (1LL << 31)) return 0;
  a.tiles_x = (a.GW + 15) / 16;
  a.tiles_y = (a.GH + 7) / 8;
  const long long ntiles = (long long)a.tiles_x * a.tiles_y * a.N;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) return 0;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv64_s2fwd_kernel<bf16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64S2F);
    (void)hipFuncSetAttribute((const void*)conv64_s2fwd_kernel<bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64S2F);
    (void)hipFuncSetAttribute((const void*)conv64_s2fwd_kernel<f16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64S2F);
    (void)hipFuncSetAttribute((const void*)conv64_s2fwd_kernel<f16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64S2F);
    attr_set = true;
  }
  const int cus = persistent_slots();
  const int per = (int)((ntiles + cus - 1) / cus);
  a.nblk_n = per;
  a.stats_tpi = a.tiles_x * a.tiles_y;
  a.stats_per = per;
  a.stats_P = (a.stats_tpi + per - 1) / per + 1;
  if (a.stats && a.stats_P > a.stats_P_max) return fsr_fail(-3, "conv64: %d partial slots per image exceed the scratch buffer's %d", a.stats_P, a.stats_P_max);
  const int grid = (int)((ntiles + per - 1) / per);
  if (dtype == FSR_F16) {
    if (a.stats) hipLaunchKernelGGL((conv64_s2fwd_kernel<f16_t, true>), dim3(grid), dim3(NTHR64), LDS64S2F, stream, a);
    else hipLaunchKernelGGL((conv64_s2fwd_kernel<f16_t, false>), dim3(grid), dim3(NTHR64), LDS64S2F, stream, a);
  } else {
    if (a.stats) hipLaunchKernelGGL((conv64_s2fwd_kernel<bf16_t, true>), dim3(grid), dim3(NTHR64), LDS64S2F, stream, a);
    else hipLaunchKernelGGL((conv64_s2fwd_kernel<bf16_t, false>), dim3(grid), dim3(NTHR64), LDS64S2F, stream, a);
  }
  fsr_note_kernel("conv64_s2fwd_kernel");
  int rc = fsr_check_launch("conv64_s2fwd_kernel");
  return rc ? rc : 1;
}

// Returns 1 if the launch was taken, 0 if the shape is not this kernel's, < 0 on error.
int fsr_conv64_persistent_try(int dtype, ConvKArgs& a, int S, hipStream_t stream) {
  // x3 (a.Cin = 128 physical bf16 channels): the thin kernel only (the head and the image gradients; round 6)
  const bool x3 = dtype == FSR_X3;
  if ((dtype != FSR_BF16 && dtype != FSR_F16 && !x3) || S != 1 || a.Cin != (x3 ? 128 : 64) || a.ntaps != 9) return 0;
  // thin: float output of at most 16 channels (head conv, image gradients); otherwise 64-channel blocks
  const bool thin = a.CoutPad == 16 && a.out_f32 && !a.ps && !a.in_ps && !a.stats && !a.preact && !a.dmask && !a.pool2;
  if (x3 && (!thin || a.Cout > 8 || a.wlin)) return 0;      // (eight filter rows per tap in LDS: conv64_thin_kernel<bf16_t, true>)
  if (!thin) {
    if (a.Cout % 64 != 0 || a.CoutPad != a.Cout || a.Cout > 256) return 0;
    if (a.in_ps || a.out_f32 || a.oscale) return 0;
    if (a.ps && (a.Cout != 256 || a.stats || a.dmask)) return 0;   // PixelShuffle(2): one quadrant = one 64-row block
    if (a.preact && a.dmask) return 0;
    if (a.pool2 && (a.ps || a.stats || a.dmask || a.preact)) return 0;
    if (a.stats && a.dmask) return 0;   // InstanceNorm backward sums: generic kernel
  }
  if (a.osy != 1 || a.osx != 1 || a.ooy != 0 || a.oox != 0) return 0;
  // (the thin kernel indexes its input with unsigned 32-bit element offsets; x3: 128 elements per pixel -- 720p x 32 frames is 3.8e9)
  if ((long long)a.N * a.IH * a.IW * (x3 ? 128 : 64) >= (x3 ? (1LL << 32) : (1LL << 31)) || (long long)a.N * a.FOH * a.FOW * a.Cout >= (1LL << 31)) return 0;
  a.tiles_x = (a.GW + 15) / 16;
  a.tiles_y = (a.GH + 15) / 16;
  a.taps_lo = 0;
  a.taps_hi = 0;
  for (int t = 0; t < 9; ++t) {
    if (a.tdy[t] > 2 || a.tdx[t] > 2) return 0;
    const unsigned code = (unsigned)a.tdy[t] | ((unsigned)a.tdx[t] << 2) | ((unsigned)a.tw[t] << 4);
    if (t < 8) a.taps_lo |= (unsigned long long)code << (8 * t);
    else a.taps_hi = code;
  }
  const long long ntiles = (long long)a.tiles_x * a.tiles_y * a.N;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) return 0;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv64_thin_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_THIN);
    (void)hipFuncSetAttribute((const void*)conv64_thin_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_THIN);
    (void)hipFuncSetAttribute((const void*)conv64_thin_kernel<bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_THIN);
    (void)hipFuncSetAttribute((const void*)conv64_v2_kernel<bf16_t, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64V2);
    (void)hipFuncSetAttribute((const void*)conv64_v2_kernel<bf16_t, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64V2);
    (void)hipFuncSetAttribute((const void*)conv64_v2_kernel<bf16_t, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64V2);
    (void)hipFuncSetAttribute((const void*)conv64_v2_kernel<f16_t, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64V2);
    (void)hipFuncSetAttribute((const void*)conv64_v2_kernel<f16_t, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64V2);
    (void)hipFuncSetAttribute((const void*)conv64_v2_kernel<f16_t, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64V2);
    attr_set = true;
  }
  const int cus = persistent_slots();    // one persistent workgroup per CU (LDS admits exactly one)
  const int nblk = thin ? 1 : a.Cout / 64;                   // channel blocks: workgroup w -> block w % nblk
  int slots = thin ? 2 * cus : cus / nblk;                   // tile ranges (the thin kernel's LDS admits two workgroups per CU)
  if (slots < 1) slots = 1;
  const int per = (int)((ntiles + slots - 1) / slots);      // contiguous tiles per workgroup
  a.nblk_n = per;
  // statistics: one partial slot per tile range that intersects an image
  a.stats_tpi = a.tiles_x * a.tiles_y;
  a.stats_per = per;
  a.stats_P = (a.stats_tpi + per - 1) / per + 1;
  if (a.stats && a.stats_P > a.stats_P_max) return fsr_fail(-3, "conv64: %d partial slots per image exceed the scratch buffer's %d", a.stats_P, a.stats_P_max);
  const int grid = (int)((ntiles + per - 1) / per) * nblk;
  if (x3) {
    hipLaunchKernelGGL((conv64_thin_kernel<bf16_t, true>), dim3(grid), dim3(NTHR64), LDS_THIN, stream, a);
  } else if (dtype == FSR_F16) {
    if (thin) hipLaunchKernelGGL((conv64_thin_kernel<f16_t>), dim3(grid), dim3(NTHR64), LDS_THIN, stream, a);
    else if (a.stats) hipLaunchKernelGGL((conv64_v2_kernel<f16_t, true, false>), dim3(grid), dim3(NTHR64), LDS64V2, stream, a);
    else if (a.dmask) hipLaunchKernelGGL((conv64_v2_kernel<f16_t, false, true>), dim3(grid), dim3(NTHR64), LDS64V2, stream, a);
    else hipLaunchKernelGGL((conv64_v2_kernel<f16_t, false, false>), dim3(grid), dim3(NTHR64), LDS64V2, stream, a);
  } else {
    if (thin) hipLaunchKernelGGL((conv64_thin_kernel<bf16_t>), dim3(grid), dim3(NTHR64), LDS_THIN, stream, a);
    else if (a.stats) hipLaunchKernelGGL((conv64_v2_kernel<bf16_t, true, false>), dim3(grid), dim3(NTHR64), LDS64V2, stream, a);
    else if (a.dmask) hipLaunchKernelGGL((conv64_v2_kernel<bf16_t, false, true>), dim3(grid), dim3(NTHR64), LDS64V2, stream, a);
    else hipLaunchKernelGGL((conv64_v2_kernel<bf16_t, false, false>), dim3(grid), dim3(NTHR64), LDS64V2, stream, a);
  }
  fsr_note_kernel(x3 ? "conv64_thin_kernel<x3>" : (thin ? "conv64_thin_kernel" : "conv64_v2_kernel"));
  int rc = fsr_check_launch(thin ? "conv64_thin_kernel" : "conv64_v2_kernel");
  return rc ? rc : 1;
}
