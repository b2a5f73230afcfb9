// conv_tall3_body.h -- the kernel template of conv_tall3 (included by conv_tall3_{bf16,f16,x3n,x3w}.hip: one translation unit per
// family of instantiations, so that the 50-odd kernels compile in parallel; the dispatcher is conv_tall3.hip).
//
// 3x3 convolution for the 128..512-channel layers: persistent implicit GEMM on v_mfma_f32_32x32x16_{bf16,f16}
// with BOTH operands arriving by LDS-DMA.  (Round 3: replaces the 16x16x32 "tall" configuration of conv_igemm.hip on the
// layers that dominate the GAN iteration.  Round 4: the stride-2 forward of the same layers, template parameter S = 2.)
//
// Replaces the torch.nn.Conv2d(k=3, p=1, stride 1) calls of the reference with 128 or more input channels and a
// multiple of 128 output channels:
//   torchvision vgg19.features convs 7..32 behind /root/reference/model.py:8 (conv2_2, conv3_x, conv4_x: forward twice per
//   iteration, trainer.py:190-191, and their data gradients once, trainer.py:195),
//   /root/reference/model.py:160-177 (Discriminator 128->256 and 256->512 stride-1 blocks: forwards -- with the sums and
//   sums of squares of their InstanceNorm, the STATS instantiation -- and data gradients).
//   /root/reference/model.py:154-159 (Discriminator 64->128 block: its data gradient, 128 -> 64 channels, on 64-channel blocks).
//   /root/reference/model.py:160-183, strides 2 at :162, :172, :182 (Discriminator 128->128 @192^2, 256->256 @96^2,
//   512->512 @48^2: the stride-2 FORWARDS with their InstanceNorm sums, S = 2 below).
//
// Work decomposition
//   tile      16 x 16 output pixels of one image x BN output channels (BN = 256: 8 waves, one workgroup per CU;
//             BN = 128: 4 waves, two workgroups per CU); a workgroup walks tiles persistently (tile += gridDim)
//   wave      128 pixels (8 rows) x 64 channels = 2 filter fragments x 4 pixel fragments of v_mfma_f32_32x32x16:
//             6 ds_read_b128 per 8 MFMAs of 32 cycles (the 16x16x32 form of conv_igemm.hip: 12 per 32 MFMAs of 16)
//   K loop    chunk of 32 input channels x tap; a "stage" = G taps of one chunk, a "substep" = 16 channels of one tap.
//             Fragment reads run ONE substep ahead of the MFMAs that consume them, spread between those MFMAs.
// LDS (bytes)  halo[2][21 KB]   18 x 18 pixels x 32 channels of chunk c / c+1, 64 B per pixel, unpadded
//              ring[NSLOT][G][BN][64 B]   filter slices of the stages in flight
//   Every byte arrives by LDS-DMA (buffer_load_dwordx4 ... lds: 1 KB per wave instruction, written linearly): no staging
//   registers, no ds_write.  Conflict-free ds_read_b128 needs the 16-byte unit index XOR-swizzled: filter row R keeps
//   channel unit u at u ^ ((R >> 2) & 3), halo column x at u ^ ((x >> 1) & 3) (checked exhaustively for the four lane
//   groups, every tap and both K halves); the swizzle is applied to the SOURCE address of the DMA and to the read.
//   Image borders are buffer-range misses (voffset = ~0): the DMA writes zeros, there is no zero page and no branch.
// Synchronisation: ONE s_barrier per stage, placed before the stage's last substep: it publishes the pieces of the next
//   stage (each wave waits for its own pieces with a counted vmcnt first) and retires the reads of the slot that the DMA
//   issued after it overwrites.  With G = 3 that is one barrier per 48 MFMAs (1536 matrix-pipe cycles) per wave.
// Output mapping: LDS filter row i of a 32-row block holds output channel 16*((i>>2)&1) + (i&3) + 4*(i>>3), so a lane's
//   16 accumulator registers are 16 CONSECUTIVE channels of one pixel: two 16-byte stores per fragment pair.
//
// Stride 2 (S = 2): output pixel (oy, ox) reads input (2 oy + ky - 1, 2 ox + kx - 1).  Split by the PARITY of (ky, kx) the
//   input of a tile is four planes, and inside a plane tap (ky, kx) reads position (oy + (ky >> 1), ox + (kx >> 1)): a
//   stride-1 access with offsets 0 / 1 -- the fragment read and the swizzle of the stride-1 kernel, unchanged.  The LDS-DMA
//   GATHERS the planes (a lane's source pixel is 2 hx + px, 2 hy + py of the tile's 17 x 33 input window: no layout change in
//   HBM): plane (0,0) serves taps (0,0) (0,2) (2,0) (2,2), plane (0,1) taps (0,1) (2,1), plane (1,0) taps (1,0) (1,2),
//   plane (1,1) tap (1,1) -- the nine stages of a 32-channel chunk walk the planes in that order.  An 8 x 16-output tile
//   keeps each plane in its OWN buffer (9 x 18 pixels x 64 B = 11 pieces; 4 x 11 KB + the 32 KB filter ring = 79 KB: two
//   workgroups per CU), refilled for the next chunk as soon as its last tap has been read: the 44 pieces of a chunk are dealt
//   round-robin over the four waves (11 each, so every wave issues the same number of pieces per stage and the counted
//   vmcnt waits stay compile-time constants) and issued on a fixed schedule, one or two per stage, each at least three
//   stages before its plane's first use.  Four times the input pixels per output pixel make this form DMA-issue-heavy (3.2
//   pieces per 8 MFMAs and wave against 2.7 per 16 at stride 1); that, not LDS capacity, is what bounds it.
#pragma once
#include "fsr_common.h"
#include "fsr_conv_args.h"
#include "fsr_host.h"

#include <stdlib.h>

// Ablation builds (tools/build_variant.sh -DFSR_ABL3=<mask>; results WRONG on purpose, the product library is built with 0):
//   1 no stores   2 no DMA   4 no main loop   16 lgkmcnt(0) before every barrier   32 no barriers   64 no DMA waits
#ifndef FSR_ABL3
#define FSR_ABL3 0
#endif


int fsr_t3_cus();      // conv_tall3.hip: CUs the persistent walk is sized for

namespace {

constexpr int T3_ABL = FSR_ABL3;

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T> struct Mfma32;
template <> struct Mfma32<bf16_t> {
  static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
  }
};
template <> struct Mfma32<f16_t> {
  static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_hw, a), __builtin_bit_cast(f16x8_hw, b), c, 0, 0, 0);
  }
};

// stride 2: the tap served by position p of a chunk's nine stages (plane by plane: (0,0) x 4, (0,1) x 2, (1,0) x 2, (1,1))
constexpr int T3_S2_TAP[9] = {0, 2, 6, 8, 1, 7, 3, 5, 4};
template <int S> constexpr int t3_tap_at(int pos) { return S == 2 ? T3_S2_TAP[pos] : pos; }
constexpr int t3_plane_of_tap(int t) { return ((t / 3) & 1) * 2 + ((t % 3) & 1); }

constexpr int T3_P = 18;                           // halo columns (16 outputs + the 3 x 3 footprint)
constexpr int T3_PIXB = 64, T3_ROWB = T3_P * T3_PIXB;   // bytes per halo pixel / halo row
// A tile is TH = 4 * MB output rows x 16 columns (MB = pixel fragments of two rows per wave: 4, 3 or 2 -> 16, 12, 8 rows)
template <int MB> constexpr int t3_hunits() { return (4 * MB + 2) * T3_P * 4; }          // 16-byte units of one halo chunk
template <int MB> constexpr int t3_nhp() { return (t3_hunits<MB>() + 63) / 64; }         // DMA pieces per halo chunk
template <int MB> constexpr int t3_halo_bytes() { return t3_nhp<MB>() * 1024; }

// stride 2: one buffer per parity plane, (TH + 1) rows x 18 columns (17 used) of 64-byte pixels, whole DMA pieces
template <int MB> constexpr int t3_s2_plane_units() { return (4 * MB + 1) * T3_P * 4; }
template <int MB> constexpr int t3_s2_npp() { return (t3_s2_plane_units<MB>() + 63) / 64; }       // pieces per plane (11)
template <int MB, int S> constexpr int t3_halo_total() { return S == 2 ? 4 * t3_s2_npp<MB>() * 1024 : 2 * t3_halo_bytes<MB>(); }

template <int BN, int G, int NSLOT, int MB, int S = 1> constexpr int t3_lds_bytes() { return t3_halo_total<MB, S>() + NSLOT * G * BN * 64 + 2 * 1024; }   // + two bias pieces

__device__ __forceinline__ int t3_swz_row(int R) { return (R >> 2) & 3; }
__device__ __forceinline__ int t3_swz_col(int x) { return (x >> 1) & 3; }

template <typename V>
__device__ __forceinline__ V t3_lds_read(const char* smem, unsigned off) {
  return *FSR_LDS_PTR(const V, smem + off);
}

// X3 (FSR_X3, T = bf16_t): the input is an x3 tensor seen as a bf16 tensor of a.Cin = 2 x logical channels whose
// 32-channel chunks alternate hi / lo, the filter pack alternates w_hi / w_lo chunks the same way.  The kernel then walks
// THREE virtual chunks per logical 32-channel group through the unchanged pipeline: only the source offsets of the DMA pieces
// are mapped (t3_hmap / t3_fmap), and the epilogue stores / reads x3 elements (hi and lo 64 bytes apart, fsr_common.h).
// Order: stride 2 -- (x_hi, w_hi), (x_lo, w_hi), (x_hi, w_lo), every chunk fetching its halo; stride 1 (REUSE below) --
// (x_hi, w_hi), (x_hi, w_lo), (x_lo, w_hi), the x_hi halo fetched ONCE.
__device__ __forceinline__ int t3_div3(int j) { return (int)(((unsigned)j * 0xAAABu) >> 17); }     // j < 2^15
// REUSE (x3, stride 1; round 6): the virtual chunks of a group are walked (x_hi, w_hi), (x_hi, w_lo), (x_lo, w_hi), and the second one
// reads the x_hi halo the first one brought in -- no second DMA of it.  Halo chunks then no longer alternate buffers by chunk
// parity: chunks 0 and 1 of a group live in buffer 0, chunk 2 in buffer 1 (every tile has a multiple of three chunks).
template <bool X3, bool REUSE = false> __device__ __forceinline__ int t3_hmap(int j) {      // physical input chunk of virtual chunk j
  if constexpr (!X3) return j;
  const int g = t3_div3(j), r = j - 3 * g;
  return __builtin_amdgcn_readfirstlane(2 * g + (r == (REUSE ? 2 : 1) ? 1 : 0));      // (wave-uniform: the DMA's scalar offset operand)
}
template <bool X3, bool REUSE = false> __device__ __forceinline__ int t3_fmap(int j) {      // physical filter chunk of virtual chunk j
  if constexpr (!X3) return j;
  const int g = t3_div3(j), r = j - 3 * g;
  return __builtin_amdgcn_readfirstlane(2 * g + (r == (REUSE ? 1 : 2) ? 1 : 0));
}

// PSM (x3 only; round 6: a TEMPLATE parameter -- as run-time flags of the shared body these paths cost every plain instantiation
// registers, SGPR spills and two scratch reloads per tile, round-5 verdict item 2):
//   0  plain   1  depth-to-space INPUT (the data gradient of a PixelShuffle convolution)
//   2  the up-sampling epilogue family: PixelShuffle store (a.ps), PReLU with a trained slope, pre-activation copy (a.preact)
template <typename T, int BN, int NW, int G, int NSLOT, int MB, int NA = 2, bool STATS = false, int S = 1, bool X3 = false, int PSM = 0>
__global__ __launch_bounds__(NW * 64, 2) void conv_tall3_kernel(const ConvKArgs a) {
  static_assert(!X3 || std::is_same<T, bf16_t>::value, "x3: bf16 planes");
  static_assert(PSM == 0 || (X3 && S == 1 && !STATS && (NA == 2 || PSM == 1)), "depth-to-space forms: x3, stride 1, no statistics; the store form on the 128-channel block");
  constexpr bool IN_PS = PSM == 1, UP = PSM == 2;
  constexpr bool REUSE = X3 && S == 1;      // the x_hi halo of a group is fetched once (t3_hmap)
  typedef typename std::conditional<X3, x3_t, T>::type ST;    // storage type of the output-side tensors
  // a wave owns 32 * MB pixels x 32 * NA channels; two waves per SIMD (256 registers each), from two workgroups.  (The
  // 128 x 128 wave tile with 256 accumulators in AGPRs and ONE 512-register wave per SIMD was built and measured in round 3 --
  // 620..800 TFLOP/s against 1130..1200, profiles/r03_conv_tall3_one_wave_per_simd.txt -- and removed.)
  static_assert(BN % (NA * 32) == 0 && NW == 2 * (BN / (NA * 32)), "two pixel-row groups x BN / (32 NA) channel groups of waves");
  static_assert(NA == 1 || NA == 2, "1 (64-channel blocks) or 2 filter fragments per wave");
  static_assert(MB >= 2 && MB <= 4, "8, 12 or 16 tile rows");
  static_assert(S == 1 || (S == 2 && G == 1 && NSLOT == 4 && NW == 4 && MB == 2), "stride 2: the 8 x 16-output tile, one-tap stages, four waves");
  constexpr int TH = 4 * MB;                       // tile rows
  constexpr int T3_HUNITS = t3_hunits<MB>(), T3_NHP = t3_nhp<MB>(), T3_HALO_BYTES = t3_halo_bytes<MB>();
  constexpr int HALO_TOTAL = t3_halo_total<MB, S>();                                        // bytes of all halo buffers
  constexpr int S2_PUNITS = t3_s2_plane_units<MB>(), S2_NPP = t3_s2_npp<MB>();              // stride 2: units / pieces of one plane
  static_assert(S == 1 || (4 * S2_NPP) % NW == 0, "stride 2: the planes' pieces divide evenly over the waves");
  constexpr int NM = NA * MB;                      // MFMAs per substep (NA filter x MB pixel fragments)
  constexpr int NR = NA + MB;                      // fragment reads per substep
  static_assert(9 % G == 0 && (9 / G) % NSLOT == 1, "stage s lives in slot s % NSLOT == (chunk + stage in chunk) % NSLOT");
  constexpr int WCO = BN / (NA * 32);
  constexpr int SPC = 9 / G;                       // stages per chunk
  constexpr int NQ = 2 * G;                        // substeps per stage
  constexpr int D = NSLOT - 1;                     // the DMA runs D stages ahead
  constexpr int FP = BN / 16 / NW;                 // filter pieces per tap and wave (2)
  constexpr int HPW = S == 2 ? 4 * S2_NPP / NW : (T3_NHP + NW - 1) / NW;      // halo pieces per chunk and wave (stride 2: 11, dealt round-robin)
  constexpr int SLOT_BYTES = G * BN * 64;
  constexpr int HPS = (HPW + SPC - 1) / SPC > 1 ? (HPW + SPC - 1) / SPC : 1;   // halo pieces a wave issues per stage
  static_assert(S == 2 || (HPW + HPS - 1) / HPS + D <= SPC + 1, "halo pieces must land before their chunk starts");
  static_assert(FP >= 1 && FP <= NM && (S == 2 || HPS + 1 <= NM), "one DMA piece per MFMA slot at most");
  // (the 64-channel block -- x3: the discriminator's 64 -> 64 stride-2 forward -- has two MFMA slots per substep: its filter piece
  // shares slot 1 with a halo piece; every piece of a stage is still issued in its first substep, which is all the counted waits need)
  static_assert(S == 1 || (NM - FP >= (NA == 1 ? 1 : 2) && HPW == 11 && D == 3), "stride 2: MFMA slots 0 and 1 of a stage's first substep carry its halo pieces (schedule below)");

  HIP_DYNAMIC_SHARED(char, smem)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wpx = wave / WCO, wco = wave % WCO;
  const int l31 = lane & 31, hi = lane >> 5, l15 = lane & 15, lrow = (lane >> 4) & 1;

  const fsr_lds_addr_t halo_addr = FSR_LDS_ADDR(smem);
  const fsr_lds_addr_t ring_addr = halo_addr + HALO_TOTAL;
  float pslope = 0.f;
  if constexpr (UP) {
    if (a.act == FSR_ACT_PRELU) pslope = a.prelu[0];      // (a scalar load at kernel start: lgkmcnt, not the DMA's vmcnt)
  }
  const fsr_buf_t in_buf = fsr_make_buf(a.in, (unsigned)((size_t)a.N * a.IH * a.IW * a.Cin * sizeof(T)));
  const fsr_buf_t w_buf = fsr_make_buf(a.wpk, (unsigned)((size_t)9 * a.CoutPad * a.Cin * sizeof(T)));
  const int nchunks = X3 ? 3 * (a.Cin >> 6) : a.Cin >> 5;       // x3: three virtual chunks per (hi, lo) pair of physical ones
  const unsigned wcs = a.wlin ? (unsigned)(9 * BN * 64) : 64u;      // byte step of the filter source from chunk to chunk

  // ---- loop-invariant per-lane addresses -------------------------------------------------------------------------
  // filter fragment (n, k half j) of tap g of the stage in ring slot `sl`: ring + sl + g*BN*64 + n*2048 + aoff[j]
  unsigned aoff[2];
  {
    const int R = wco * (NA * 32) + l31;
#pragma unroll
    for (int j = 0; j < 2; ++j) aoff[j] = (unsigned)(HALO_TOTAL + R * 64 + (((2 * j + hi) ^ t3_swz_row(R)) << 4));
  }
  // pixel fragment m of tap (ky, kx), k half j, halo buffer hb: hb*HALO + (2m + ky)*ROWB + boff[kx][j]
  // (stride 2: plane*PLANE + (2m + (ky >> 1))*ROWB + boff[kx >> 1][j] -- the same per-lane bases)
  unsigned boff[3][2];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      boff[kx][j] = (unsigned)(((wpx * 2 * MB + lrow) * T3_P + l15 + kx) * T3_PIXB + (((2 * j + hi) ^ t3_swz_col(l15 + kx)) << 4));
  // DMA source of filter piece k (rows 16*(wave + k*NW) .. +15 of a slice block): byte offset of this lane's 16 bytes
  unsigned wvoff[FP];
#pragma unroll
  for (int k = 0; k < FP; ++k) {
    const int R = (wave + k * NW) * 16 + (lane >> 2), ul = lane & 3, i = R & 31;
    const int co = (R & ~31) + 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3);
    wvoff[k] = a.wlin ? (unsigned)((wave + k * NW) * 1024 + lane * 16)      // stage-contiguous pack: a piece = 1 KB of contiguous memory
                      : (unsigned)((co * a.Cin + ((ul ^ t3_swz_row(R)) << 3)) * (int)sizeof(T));
  }

  f32x16 acc[NA][MB];
  s16x8 fa[2][NA], fb[2][MB];

  // ---- per-tile state: the tile being accumulated (`cur`) and the tile the DMA stream moves on to once the current
  // tile's last stages have their pieces (`nxt`).  The pipeline NEVER drains between tiles: halo chunks keep alternating
  // buffers and stages keep rotating through the ring across the tile boundary (global chunk counter `gc`).
  struct TileC { int img, gy0, gx0, nb; };
  TileC cur = {0, 0, 0, 0}, nxt = {0, 0, 0, 0};
  unsigned hv_cur[HPW], hv_nxt[HPW];      // halo source offsets of this lane's pieces
  unsigned ws_cur = 0, ws_nxt = 0;        // byte offset of the tile's channel block inside a filter slice
  auto setup = [&](int tile, TileC& tc, unsigned (&hv)[HPW], unsigned& ws) {
    int L = tile;
    tc.nb = L % a.nblk_n; L /= a.nblk_n;
    const int tx = L % a.tiles_x; L /= a.tiles_x;
    const int ty = L % a.tiles_y;
    tc.img = L / a.tiles_y;
    tc.gy0 = ty * TH;
    tc.gx0 = tx * 16;
    ws = a.wlin ? (unsigned)(tc.nb * (X3 ? (a.Cin >> 5) : nchunks) * 9 * BN * 64) : (unsigned)(tc.nb * BN * a.Cin * (int)sizeof(T));
    if constexpr (S == 2) {
      // piece g = wave + 4 k of the chunk's 44: plane g / 11 = (py, px), piece g % 11 of that plane; halo pixel (hy, hx) of the
      // plane is input pixel (2 gy0 - 1 + 2 hy + py, 2 gx0 - 1 + 2 hx + px); rows beyond TH - py / columns beyond 16 - px are
      // never read (written as zeros like the image border)
#pragma unroll
      for (int k = 0; k < HPW; ++k) {
        const int g = wave + k * NW;
        const int plane = g / S2_NPP, py = plane >> 1, px = plane & 1;
        const int U = (g - plane * S2_NPP) * 64 + lane;
        const int hp = U >> 2, ul = U & 3;
        const int hy = hp / T3_P, hx = hp - hy * T3_P;
        const int iy = 2 * tc.gy0 - 1 + 2 * hy + py, ix = 2 * tc.gx0 - 1 + 2 * hx + px;
        unsigned o = ~0u;
        if (U < S2_PUNITS && hy <= TH - py && hx <= 16 - px && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW)
          o = (unsigned)((((tc.img * a.IH + iy) * a.IW + ix) * a.Cin + ((ul ^ t3_swz_col(hx)) << 3)) * (int)sizeof(T));
        hv[k] = o;
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
      const int U = (wave + k * NW) * 64 + lane;
      const int hp = U >> 2, ul = U & 3;
      const int hy = hp / T3_P, hx = hp - hy * T3_P;
      const int iy = tc.gy0 - 1 + hy, ix = tc.gx0 - 1 + hx;
      unsigned o = ~0u;                                          // beyond the buffer: the DMA writes zeros
      if (U < T3_HUNITS && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW) {
        if constexpr (!IN_PS) o = (unsigned)((((tc.img * a.IH + iy) * a.IW + ix) * a.Cin + ((ul ^ t3_swz_col(hx)) << 3)) * (int)sizeof(T));
        // depth-to-space input [N][2 IH][2 IW][Cin / 4]: quadrant (0, 0) of the pixel; the chunk's quadrant is a scalar offset (hsoff)
        else o = (unsigned)((((tc.img * 2 * a.IH + 2 * iy) * (2 * a.IW) + 2 * ix) * (a.Cin >> 2) + ((ul ^ t3_swz_col(hx)) << 3)) * (int)sizeof(T));
      }
      hv[k] = o;
    }
  };
  // byte offset (wave-uniform) of virtual chunk c inside a pixel's channels; depth-to-space input: the chunk lies in quadrant
  // q = chunk >> t3_ps_shift of the 2 x 2 block: + (q >> 1) rows and (q & 1) columns of Cin / 4 channels
  auto hsoff = [&](int c) -> unsigned {
    const int p = t3_hmap<X3, REUSE>(c);
    if constexpr (!IN_PS) return (unsigned)(p * 64);
    else {
      const int q = p >> a.t3_ps_shift, cq = p - (q << a.t3_ps_shift);
      return (unsigned)__builtin_amdgcn_readfirstlane((((q >> 1) * 2 * a.IW + (q & 1)) * (a.Cin >> 2) + cq * 32) * (int)sizeof(T));
    }
  };
  // DMA pieces.  Halo piece k of tile-chunk c -> halo buffer (global chunk parity); filter pieces of tap t of chunk c -> ring
  auto dma_halo = [&](const unsigned (&hv)[HPW], int c, int k, unsigned par) {
    if constexpr (!(T3_ABL & 2))
      if (wave + k * NW < T3_NHP)
        FSR_BLDS16(in_buf, hv[k], hsoff(c), halo_addr + (fsr_lds_addr_t)(par * T3_HALO_BYTES + (wave + k * NW) * 1024));
  };
  // stride 2: piece wave + 4 k of chunk c's four planes (the planes lie back to back: piece g goes to g * 1 KB)
  auto dma_halo2 = [&](const unsigned (&hv)[HPW], int c, int k) {
    if constexpr (!(T3_ABL & 2))
      FSR_BLDS16(in_buf, hv[k], (unsigned)(t3_hmap<X3>(c) * 64), halo_addr + (fsr_lds_addr_t)((wave + k * NW) * 1024));
  };
  // filter pieces of the tap at position `pos` of chunk c's stages (stride 1: the tap itself; stride 2: T3_S2_TAP[pos])
  auto dma_filter = [&](unsigned ws, int c, unsigned woff_tap, int k, unsigned slot_tap_off) {
    if constexpr (!(T3_ABL & 2))
      FSR_BLDS16(w_buf, wvoff[k], woff_tap + ws + (unsigned)(t3_fmap<X3, REUSE>(c)) * wcs, ring_addr + (fsr_lds_addr_t)(slot_tap_off + (wave + k * NW) * 1024));
  };
  // The bias of a tile's channel block comes by DMA as well (one piece of BN floats, wave 0, double buffered by tile parity):
  // an ordinary global load next to the epilogue's stores would make hipcc drain vmcnt -- the whole DMA pipeline -- per tile.
  constexpr int BIAS_BYTES = BN * 4;                       // <= 1024: one piece
  const fsr_lds_addr_t bias_addr = ring_addr + NSLOT * SLOT_BYTES;
  const fsr_buf_t bias_buf = fsr_make_buf(a.bias, a.bias ? (unsigned)(a.Cout * 4) : 0u);
  // x3, PixelShuffle epilogue (a.ps: rows packed [quadrant][channel], the bias in torch order 4 * channel + quadrant): the WHOLE
  // bias (Cout <= 256 floats: one piece) is fetched and acc_init gathers from it
  const bool ps_out = UP && a.ps != 0;
  auto dma_bias = [&](int nbk, unsigned par) {
    if (!(T3_ABL & 2) && wave == 0 && a.bias) {
      unsigned vo = lane * 16 < BIAS_BYTES ? (unsigned)(nbk * BIAS_BYTES + lane * 16) : ~0u;
      if constexpr (UP) {
        if (ps_out) vo = lane * 16 < a.Cout * 4 ? (unsigned)(lane * 16) : ~0u;
      }
      FSR_BLDS16(bias_buf, vo, 0u, bias_addr + (fsr_lds_addr_t)(par * 1024));
    }
  };
  auto acc_init = [&](unsigned par) {
    // the accumulators start at the bias (a lane's 16 registers of fragment row n are 16 consecutive channels): the epilogue
    // has no bias pass
#pragma unroll
    for (int n = 0; n < NA; ++n) {
      f32x16 b0;
#pragma unroll
      for (int e = 0; e < 16; ++e) b0[e] = 0.f;
      if (UP && ps_out && a.bias) {
        const int row = cur.nb * BN + wco * (NA * 32) + n * 32 + hi * 16;          // 16 rows of one quadrant (Cout / 4 % 16 == 0)
        const int q = row >> a.t3_ps_shift, cc = row - (q << a.t3_ps_shift);
        const unsigned bo = (unsigned)(HALO_TOTAL + NSLOT * SLOT_BYTES) + par * 1024 + (unsigned)((4 * cc + q) * 4);
#pragma unroll
        for (int e = 0; e < 16; ++e) b0[e] = t3_lds_read<float>(smem, bo + 16 * e);
      } else if (a.bias) {
        const unsigned bo = (unsigned)(HALO_TOTAL + NSLOT * SLOT_BYTES) + par * 1024 + (unsigned)((wco * (NA * 32) + n * 32 + hi * 16) * 4);
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const f32x4 b = t3_lds_read<f32x4>(smem, bo + 16 * e4);
#pragma unroll
          for (int e = 0; e < 4; ++e) b0[4 * e4 + e] = b[e];
        }
      }
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[n][m] = b0;
    }
  };
  auto slot_of = [&](int gchunk, int si) { return (unsigned)(((gchunk + si) & (NSLOT - 1)) * SLOT_BYTES); };

  // reads of substep (tap t = ky*3+kx in ring position g of its stage, k half j): fragment r in the MFMA's need order
  // a0 b0 .. b(MB-1) a1 .. a(NA-1)
  auto read_frag = [&](auto rc, auto bufc, auto tc, auto gc_, auto jc, unsigned sl, unsigned hb) {
    constexpr int r = decltype(rc)::value, buf = decltype(bufc)::value, t = t3_tap_at<S>(decltype(tc)::value), g = decltype(gc_)::value,
                  j = decltype(jc)::value;
    // stride 2: the tap's parity plane (its own buffer) and its offset inside the plane
    constexpr int ky = S == 2 ? (t / 3) >> 1 : t / 3, kx = S == 2 ? (t % 3) >> 1 : t % 3;
    constexpr unsigned plane_off = S == 2 ? (unsigned)(t3_plane_of_tap(t) * S2_NPP * 1024) : 0u;
    if constexpr (r == 0 || r > MB) {      // a0 first, a1 .. a(NA-1) after the pixel fragments
      constexpr int n = r == 0 ? 0 : r - MB;
      fa[buf][n] = t3_lds_read<s16x8>(smem, aoff[j] + sl + (unsigned)(g * BN * 64 + n * 2048));
    } else {
      constexpr int m = r - 1;
      fb[buf][m] = t3_lds_read<s16x8>(smem, boff[kx][j] + hb + plane_off + (unsigned)((2 * m + ky) * T3_ROWB));
    }
  };

  const int nround = (int)gridDim.x;
  int tile = (int)blockIdx.x;
  // XCD-aware order inside a round of gridDim tiles: the workgroups of one XCD take neighbouring tiles (shared halos and
  // channel blocks hit that XCD's L2)
  auto logical = [&](int t) {
    const int r0 = (t / nround) * nround;                      // first tile of this round; the last round may be partial
    const int cnt = a.t3_ntiles - r0 < nround ? a.t3_ntiles - r0 : nround;
    return r0 + xcd_remap(t - r0, cnt);
  };
  if (tile >= a.t3_ntiles) return;

  // ---- prologue of the workgroup's first tile: bias, halo chunk 0, the first D stages ------------------------------------
  setup(logical(tile), cur, hv_cur, ws_cur);
  int gc = 0;                      // global chunk counter of this workgroup (halo buffer = gc & 1, ring slot = (gc + si) % NSLOT)
  int r3 = 0;                      // REUSE: position of the chunk in its group of three (halo buffer = r3 == 2)
  // halo buffer of the chunk `d` (0 or 1) chunks after the current one
  auto hbuf = [&](int d) -> unsigned {
    if constexpr (REUSE) return (unsigned)((r3 + d == 2 || r3 + d == 5) ? 1 : 0);
    else return (unsigned)((gc + d) & 1);
  };
  unsigned tpar = 0;               // tile parity (bias buffer)
  dma_bias(cur.nb, tpar);
  if constexpr (S == 2) {
    // what the steady-state schedule issues before a chunk starts: its pieces 0..4 (plane (0,0) and the head of plane (0,1))
#pragma unroll
    for (int k = 0; k < 5; ++k) dma_halo2(hv_cur, 0, k);
  } else {
#pragma unroll
    for (int k = 0; k < HPW; ++k) dma_halo(hv_cur, 0, k, 0u);
  }
  static_for<0, D>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    static_assert(D < SPC, "the first D stages lie in chunk 0");
    static_for<0, G>([&](auto gc_) {
      constexpr int g = decltype(gc_)::value;
#pragma unroll
      for (int k = 0; k < FP; ++k) dma_filter(ws_cur, 0, a.t3_woff[t3_tap_at<S>(s * G + g)], k, slot_of(0, s) + g * BN * 64);
    });
  });
  int next = tile + nround;
  bool has_nxt = next < a.t3_ntiles;
  FSR_WAIT_VM(0);
  FSR_BARRIER();
  acc_init(tpar);
  static_for<0, NR>([&](auto rc) {   // fragments of substep 0
    read_frag(rc, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{},
              std::integral_constant<int, 0>{}, slot_of(0, 0), 0u);
  });
  int issued_prev = 1;             // did the preceding stage issue filter pieces (counted waits, D = 3)
  bool full_prev = true;           // stride 2: did the preceding stage issue its complete static set of pieces

  for (;;) {
    for (int c = 0; c < ((T3_ABL & 4) ? 0 : nchunks); ++c, ++gc, r3 = (r3 == 2 ? 0 : r3 + 1)) {
      const bool last = c + 1 == nchunks;
      if (last && has_nxt) {         // from here on the DMA feeds the next tile
        setup(logical(next), nxt, hv_nxt, ws_nxt);
        dma_bias(nxt.nb, tpar ^ 1u);
      }
      static_for<0, SPC>([&](auto sic) {
        constexpr int si = decltype(sic)::value;
        const unsigned sl = slot_of(gc, si);
        const unsigned hb = S == 2 ? 0u : hbuf(0) * (unsigned)T3_HALO_BYTES;
        // the stage whose pieces this stage issues: D stages ahead, possibly in the next tile
        constexpr int siD = (si + D) % SPC, dcD = (si + D) / SPC;
        const bool crossD = c + dcD >= nchunks;
        const int cD = crossD ? c + dcD - nchunks : c + dcD;
        const bool issue = crossD ? has_nxt : true;
        const unsigned wsD = crossD ? ws_nxt : ws_cur;
        const unsigned slD = slot_of(gc + dcD, siD);
        // the stage after this one (its first fragments are read in this stage's last substep)
        constexpr int siN = (si + 1) % SPC, dcN = (si + 1) / SPC;
        const bool has_next_stage = (c + dcN < nchunks) || has_nxt;
        const unsigned slN = slot_of(gc + dcN, siN);
        const unsigned hbN = S == 2 ? 0u : hbuf(dcN) * (unsigned)T3_HALO_BYTES;
        // stride 2: this stage's pieces are its FP filter pieces + one halo piece (+ a second one in stages 4, 5), all issued
        // in its first substep; the set is complete when the filter stage ahead exists and (stages 4..8) a next chunk does
        const bool full_this = S == 2 ? (issue && (si < 4 || !last || has_nxt)) : true;

        static_for<0, NQ>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          constexpr int buf = q & 1;
          if constexpr (q == NQ - 1) {
            // publish the next stage: this wave's pieces of stage s+1 have landed when at most the pieces of the D-1 younger
            // stages are outstanding.  Loads retire in order, so halo / bias pieces and the previous tile's stores in the
            // queue can only make the wait longer, never shorter.
            if constexpr ((T3_ABL & 64) != 0) {
              // EXPERIMENT (wrong results): no wait for the DMA
            } else if constexpr (S == 2) {
              // everything issued two or more stages ago has landed when at most the pieces of the previous and of this stage
              // are outstanding (a stage issues all its pieces in its first substep, i.e. before this wait): 3 per stage, 4 in
              // stages 4 and 5.  A stage that issued less (the last tile's end) makes the count meaningless: wait for all.
              constexpr int sp = (si + SPC - 1) % SPC;
              constexpr int allowed = FP * G + 1 + (sp == 4 || sp == 5 ? 1 : 0) + FP * G + 1 + (si == 4 || si == 5 ? 1 : 0);
              if (full_prev && full_this) FSR_WAIT_VM(allowed);
              else FSR_WAIT_VM(0);
            } else if constexpr (D == 1) {
              FSR_WAIT_VM(0);
            } else {
              static_assert(D <= 3, "add cases for deeper rings");
              const int young = (issue ? 1 : 0) + (D == 3 ? issued_prev : 0);
              if (young == 0) FSR_WAIT_VM(0);
              else if (young == 1) FSR_WAIT_VM(G * FP);
              else FSR_WAIT_VM(2 * G * FP);
            }
            if constexpr ((T3_ABL & 16) != 0) FSR_WAIT_LGKM0();   // (strict form; the reads of a slot are >= 400 cycles older than any DMA into it)
            if constexpr (!(T3_ABL & 32)) FSR_BARRIER();          // (EXPERIMENT bit 32: no barrier, wrong results)
          }
          static_for<0, NM>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int n = i / MB, m = i % MB;
            acc[n][m] = Mfma32<T>::run(fa[buf][n], fb[buf][m], acc[n][m]);
            auto next_frag = [&](auto rc) {       // fragment rc of the NEXT substep, into the other register set
              if constexpr (decltype(rc)::value < NR) {
                if constexpr (q + 1 < NQ) {
                  constexpr int q1 = q + 1;
                  read_frag(rc, std::integral_constant<int, buf ^ 1>{}, std::integral_constant<int, si * G + q1 / 2>{},
                            std::integral_constant<int, q1 / 2>{}, std::integral_constant<int, q1 % 2>{}, sl, hb);
                } else {
                  if (has_next_stage)
                    read_frag(rc, std::integral_constant<int, buf ^ 1>{}, std::integral_constant<int, siN * G>{},
                              std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, slN, hbN);
                }
              }
            };
            next_frag(ic);                                             // one read per MFMA slot ...
            next_frag(std::integral_constant<int, i + NM>{});           // ... two where a substep has more reads than MFMAs (NA = 1)
            // DMA pieces of stage s+D: the halo piece first (the likeliest HBM miss), then two filter pieces per substep
            if constexpr (S == 2) {
              // halo schedule of stride 2 (piece k of a chunk = piece wave + 4 k of its 44; every piece is issued >= 3 stages
              // before its plane's first read and after that plane's last read of the previous chunk):
              //   stage    0  1  2  3  4  5        this chunk's pieces 5 .. 10   (planes (0,1) tail, (1,0), (1,1))
              //   stage    4  5  6  7  8           the NEXT chunk's pieces 0 .. 4 (plane (0,0), head of (0,1))
              if constexpr (q == 0 && i == 1 && si <= 5) dma_halo2(hv_cur, c, si + 5);
              if constexpr (q == 0 && i == (si <= 5 ? 0 : 1) && si >= 4) {
                if (!last) dma_halo2(hv_cur, c + 1, si - 4);
                else if (has_nxt) dma_halo2(hv_nxt, 0, si - 4);
              }
            } else if constexpr (q == 0 && i >= 1 && i <= HPS && si * HPS + (i - 1) < HPW) {
              constexpr int hk = si * HPS + (i - 1);
              // (REUSE: the chunk after position 0 reads the halo already there)
              if (!last) {
                if (!REUSE || r3 != 0) dma_halo(hv_cur, c + 1, hk, hbuf(1));
              } else if (has_nxt) dma_halo(hv_nxt, 0, hk, hbuf(1));
            }
            if constexpr (q < G && i >= NM - FP) {
              if (issue) dma_filter(wsD, cD, a.t3_woff[t3_tap_at<S>(siD * G + q)], i - (NM - FP), slD + q * BN * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
          });
        });
        issued_prev = issue ? 1 : 0;
        full_prev = full_this;
      });
    }

    // ---- epilogue -----------------------------------------------------------------------------------------------------
    // activation as ONE instruction per element: ReLU = max(v, 0), LeakyReLU = max(v, slope * v) (0 <= slope <= 1: host
    // checked), identity = nothing.  The three forms are separate instantiations of the store loop (one wave-uniform branch).
    ST* outp = (ST*)a.out;
    const ST* maskp = (const ST*)a.dmask;
    // 16 consecutive channels of one pixel -> memory (two 16-byte units; x3: two for the hi parts, two for the lo parts)
    auto store16 = [&](ST* p, const float (&v)[16], auto&& fn) {
      if constexpr (X3) {
        char* hp = (char*)x3_hi_ptr(p);
#pragma unroll
        for (int h = 0; h < 2; ++h) {       // eight channels at a time: the split needs two temporaries per pair
          u32x4 hh, ll;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            unsigned a_, b_;
            x3_split2(fn(v[8 * h + 2 * e]), fn(v[8 * h + 2 * e + 1]), a_, b_);
            hh[e] = a_;
            ll[e] = b_;
          }
          fsr_st<1>((u32x4*)(hp + 16 * h), hh);
          fsr_st<1>((u32x4*)(hp + 64 + 16 * h), ll);
        }
      } else {
        u32x4 p0, p1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          p0[e] = pack2<T>(fn(v[2 * e]), fn(v[2 * e + 1]));
          p1[e] = pack2<T>(fn(v[8 + 2 * e]), fn(v[8 + 2 * e + 1]));
        }
        fsr_st<1>((u32x4*)p, (u32x4)(p0));
        fsr_st<1>((u32x4*)(p + 8), (u32x4)(p1));
      }
    };
    const int oimg = cur.img, ogy0 = cur.gy0, ogx0 = cur.gx0, onb = cur.nb;
    const int gx = ogx0 + l15;
    ST* prep = (ST*)a.preact;
    auto store_tile = [&](auto actc) {
      constexpr int ACT = decltype(actc)::value;
      const float slope = ACT == FSR_ACT_PRELU ? pslope : a.slope;
      auto activate = [&](float x) {
        if constexpr (ACT == FSR_ACT_RELU) return fmaxf(x, 0.f);
        else if constexpr (ACT == FSR_ACT_LEAKY) return fmaxf(x, x * slope);
        else if constexpr (ACT == FSR_ACT_PRELU) return x > 0.f ? x : x * slope;      // any slope (a trained PReLU weight)
        else return x;
      };
      static_for<0, NA>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        const int co = onb * BN + wco * (NA * 32) + n * 32 + hi * 16;
        static_for<0, MB>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          const int gy = ogy0 + wpx * 2 * MB + 2 * m + lrow;
          const bool ok = gy < a.GH && gx < a.GW;
          float v[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = acc[n][m][e];
          if (a.pool2) {
            // MaxPool2d(2,2) fused: rows (gy, gy^1) sit in lanes (l, l^16), columns in (l, l^1); the activation is monotonic
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              float x = v[e];
              x = fmaxf(x, __shfl_xor(x, 16, 64));
              x = fmaxf(x, __shfl_xor(x, 1, 64));
              v[e] = activate(x);
            }
            if (!(T3_ABL & 1) && ok && lrow == 0 && !(l15 & 1)) {
              const unsigned off = (unsigned)((oimg * (a.FOH >> 1) + (gy >> 1)) * (a.FOW >> 1) + (gx >> 1)) * (unsigned)a.Cout + (unsigned)co;
              store16(outp + off, v, [](float x) { return x; });
            }
          } else if (!(T3_ABL & 1) && ok) {
            unsigned off = (unsigned)((oimg * a.FOH + gy) * a.FOW + gx) * (unsigned)a.Cout + (unsigned)co;
            if constexpr (UP) {
              if (ps_out) {       // depth-to-space store: packed row co = quadrant q, channel cc -> pixel (2 gy + (q >> 1), 2 gx + (q & 1))
                const int q = co >> a.t3_ps_shift, cc = co - (q << a.t3_ps_shift);
                off = (unsigned)((oimg * 2 * a.FOH + 2 * gy + (q >> 1)) * (2 * a.FOW) + 2 * gx + (q & 1)) * (unsigned)(a.Cout >> 2) + (unsigned)cc;
              }
              if (prep) store16(prep + off, v, [](float x) { return x; });      // the pre-activation a PReLU's backward needs (training)
            }

            if (maskp) {   // fused activation backward of the producing layer: dz = dx * act'(y), y = the saved forward input
              // (x3: the hi parts carry the sign; an addend needs both parts)
              const char* mp = X3 ? (const char*)x3_hi_ptr(maskp + off) : (const char*)(maskp + off);
              const u32x4 k0 = *(const u32x4*)mp, k1 = *(const u32x4*)(mp + 16);
              const float ms = a.dmask_slope;
              if (a.dmask_add) {   // the tensor is an addend (gradient of a skip connection), not a gate
                if constexpr (X3) {
                  const u32x4 q0 = *(const u32x4*)(mp + 64), q1 = *(const u32x4*)(mp + 80);
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    v[2 * e] += x3_join_lo(k0[e], q0[e]);
                    v[2 * e + 1] += x3_join_hi(k0[e], q0[e]);
                    v[8 + 2 * e] += x3_join_lo(k1[e], q1[e]);
                    v[8 + 2 * e + 1] += x3_join_hi(k1[e], q1[e]);
                  }
                } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  v[2 * e] += cvt_lo<T>(k0[e]);
                  v[2 * e + 1] += cvt_hi<T>(k0[e]);
                  v[8 + 2 * e] += cvt_lo<T>(k1[e]);
                  v[8 + 2 * e + 1] += cvt_hi<T>(k1[e]);
                }
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
            if constexpr (X3) {
              store16(outp + off, v, activate);       // (the activation inside the split: no second copy of v[] -- 254 registers, no scratch)
            } else {
              // (spelled out rather than store16(..., activate): this form compiles to round 4's epilogue -- 249 registers, no
              // scratch, 77 vmcnt(0) sites against 143)
              u32x4 p0, p1;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                p0[e] = pack2<T>(activate(v[2 * e]), activate(v[2 * e + 1]));
                p1[e] = pack2<T>(activate(v[8 + 2 * e]), activate(v[8 + 2 * e + 1]));
              }
              fsr_st<1>((u32x4*)(outp + off), (u32x4)(p0));
              fsr_st<1>((u32x4*)(outp + off + 8), (u32x4)(p1));
            }
          }
        });
      });
    };
    if constexpr (STATS) {
      // InstanceNorm statistics of the pre-activation (bias included), from the f32 accumulators, in a pass of its own BEFORE
      // the store loop and eight channels at a time: beside the accumulators it keeps 16 registers alive (the kernel has
      // none to spare), not the store loop's temporaries as well.
      // Sum over the wave's 32 pixel lanes (lane bits 0..4; bit 5 separates the two channel halves) by a butterfly that
      // HALVES the register set at its first three steps -- a lane keeps the channels whose index bit matches its lane bit
      // and hands the others to its partner: 4 + 2 + 1 + 1 + 1 shuffles per 8 channels instead of 40, in a fixed order.
      // Afterwards the four lanes that differ in bits 0, 1 all hold channel 4*b4 + 2*b3 + b2 (b_i = lane bit i) of the 8.
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
      static_for<0, 2 * NA>([&](auto nc) {
        constexpr int n = decltype(nc)::value / 2, h = decltype(nc)::value % 2;
        const int co = onb * BN + wco * (NA * 32) + n * 32 + hi * 16 + h * 8;
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
        static_for<0, MB>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          const int gy = ogy0 + wpx * 2 * MB + 2 * m + lrow;
          if (gy < a.GH && gx < a.GW) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float v = acc[n][m][h * 8 + e];
              s1[e] += v;
              s2[e] = fmaf(v, v, s2[e]);
            }
          }
        });
        butterfly(s1);
        butterfly(s2);
        // one partial slot per (tile, pixel-row group of waves): [img][slot][Cout][2], added in slot order by reduce.hip
        if (!(lane & 3)) {
          const int e = ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
          const int slot = ((ogy0 / TH) * a.tiles_x + (ogx0 >> 4)) * (NW / WCO) + wpx;
          float* sp = a.stats + (((size_t)oimg * a.stats_P + slot) * a.Cout + co + e) * 2;
          sp[0] = s1[0];
          sp[1] = s2[0];
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    }
    if (a.act == FSR_ACT_RELU) store_tile(std::integral_constant<int, FSR_ACT_RELU>{});
    else if (a.act == FSR_ACT_LEAKY) store_tile(std::integral_constant<int, FSR_ACT_LEAKY>{});
    else if (UP && a.act == FSR_ACT_PRELU) store_tile(std::integral_constant<int, UP ? FSR_ACT_PRELU : FSR_ACT_NONE>{});
    else store_tile(std::integral_constant<int, FSR_ACT_NONE>{});
    if (!has_nxt) break;
    // the next tile becomes the current one; its first fragments are already in registers, its pieces in flight
    cur = nxt;
#pragma unroll
    for (int k = 0; k < HPW; ++k) hv_cur[k] = hv_nxt[k];
    ws_cur = ws_nxt;
    tile = next;
    next = tile + nround;
    has_nxt = next < a.t3_ntiles;
    tpar ^= 1u;
    acc_init(tpar);
  }
}

template <typename T, int BN, int NW, int G, int NSLOT, int MB, int NA = 2, bool STATS = false, int S = 1, bool X3 = false, int PSM = 0>
int t3_launch(ConvKArgs& a, int wg_per_cu, hipStream_t stream) {
  auto kern = conv_tall3_kernel<T, BN, NW, G, NSLOT, MB, NA, STATS, S, X3, PSM>;
  constexpr int lds = t3_lds_bytes<BN, G, NSLOT, MB, S>();
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  // nothing of `a` is written before the launch is certain (a refused shape falls through to conv_igemm.hip with `a` intact)
  const int nblk_n = a.Cout / BN;
  const long long ntiles = (long long)a.tiles_x * a.tiles_y * a.N * nblk_n;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) return 0;
  if (STATS) {   // one partial slot per tile and pixel-row group of waves; the caller's scratch holds stats_P_max of them
    const long long slots = (long long)a.tiles_x * a.tiles_y * (NW / (BN / (NA * 32)));
    if (slots > a.stats_P_max) return 0;
    if (!a.query) {
      a.stats_P = (int)slots;
      a.stats_tpi = a.stats_per = 0;
    }
  }
  if (a.query) {            // fsr_conv3x3_pack_block: this kernel would run (every refusal above was passed: the query cannot drift
    a.wlin_want = BN;       // from the dispatch), and it takes the stage-contiguous pack of its block size
    return 1;
  }
  a.nblk_n = nblk_n;
  a.t3_ntiles = (int)ntiles;
  long long grid = (long long)fsr_t3_cus() * wg_per_cu;     // persistent tile walk, wg_per_cu workgroups per CU
  if (grid > ntiles) grid = ntiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, stream, a);
  fsr_note_kernel(S == 2 ? (STATS ? "conv_tall3_kernel<%s,%d,%d,%d,%d,%d,%d,stats,s2>" : "conv_tall3_kernel<%s,%d,%d,%d,%d,%d,%d,s2>")
                         : (STATS ? "conv_tall3_kernel<%s,%d,%d,%d,%d,%d,%d,stats>"
                                  : (PSM == 1 ? "conv_tall3_kernel<%s,%d,%d,%d,%d,%d,%d,ps_in>" : PSM == 2 ? "conv_tall3_kernel<%s,%d,%d,%d,%d,%d,%d,up>" : "conv_tall3_kernel<%s,%d,%d,%d,%d,%d,%d>")),
                  X3 ? "x3" : (std::is_same<T, f16_t>::value ? "f16" : "bf16"), BN, NW, G, NSLOT, MB, NA);
  const int rc = fsr_check_launch("conv_tall3_kernel");
  return rc ? rc : 1;
}

}  // namespace
