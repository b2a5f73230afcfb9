// 3x3 stride-1 convolution for the 128..512-channel layers: persistent implicit GEMM on v_mfma_f32_32x32x16_{bf16,f16}
// with BOTH operands arriving by LDS-DMA.  (Round 3: replaces the 16x16x32 "tall" configuration of conv_igemm.hip on the
// layers that dominate the GAN iteration.)
//
// Replaces the torch.nn.Conv2d(k=3, p=1, stride 1) calls of the reference with 128 or more input channels and a
// multiple of 128 output channels:
//   torchvision vgg19.features convs 7..32 behind /root/reference/model.py:8 (conv2_2, conv3_x, conv4_x: forward twice per
//   iteration, trainer.py:190-191, and their data gradients once, trainer.py:195),
//   /root/reference/model.py:160-177 (Discriminator 128->256 and 256->512 stride-1 blocks: forwards -- with the sums and
//   sums of squares of their InstanceNorm, the STATS instantiation -- and data gradients).
//   /root/reference/model.py:154-159 (Discriminator 64->128 block: its data gradient, 128 -> 64 channels, on 64-channel blocks).
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
#include "fsr_common.h"
#include "fsr_conv_args.h"
#include "fsr_host.h"

#include <stdlib.h>

namespace {

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

constexpr int T3_P = 18;                           // halo columns (16 outputs + the 3 x 3 footprint)
constexpr int T3_PIXB = 64, T3_ROWB = T3_P * T3_PIXB;   // bytes per halo pixel / halo row
// A tile is TH = 4 * MB output rows x 16 columns (MB = pixel fragments of two rows per wave: 4, 3 or 2 -> 16, 12, 8 rows)
template <int MB> constexpr int t3_hunits() { return (4 * MB + 2) * T3_P * 4; }          // 16-byte units of one halo chunk
template <int MB> constexpr int t3_nhp() { return (t3_hunits<MB>() + 63) / 64; }         // DMA pieces per halo chunk
template <int MB> constexpr int t3_halo_bytes() { return t3_nhp<MB>() * 1024; }

template <int BN, int G, int NSLOT, int MB> constexpr int t3_lds_bytes() { return 2 * t3_halo_bytes<MB>() + NSLOT * G * BN * 64 + 2 * 1024; }   // + two bias pieces

__device__ __forceinline__ int t3_swz_row(int R) { return (R >> 2) & 3; }
__device__ __forceinline__ int t3_swz_col(int x) { return (x >> 1) & 3; }

template <typename V>
__device__ __forceinline__ V t3_lds_read(const char* smem, unsigned off) {
  return *FSR_LDS_PTR(const V, smem + off);
}

template <typename T, int BN, int NW, int G, int NSLOT, int MB, int NA = 2, bool STATS = false>
__global__ __launch_bounds__(NW * 64, NA == 4 ? 1 : 2) void conv_tall3_kernel(const ConvKArgs a) {
  // a wave owns 32 * MB pixels x 32 * NA channels.  NA = 2: two waves per SIMD (256 registers each); NA = 4 (with MB = 4: the
  // 128 x 128 wave tile, 256 accumulator registers in the AGPR half of the file): ONE 512-register wave per SIMD, 8 fragment
  // reads per 16 MFMAs instead of 6 per 8, the 256 x 256 workgroup tile from four waves
  static_assert(BN % (NA * 32) == 0 && NW == 2 * (BN / (NA * 32)), "two pixel-row groups x BN / (32 NA) channel groups of waves");
  static_assert(NA == 1 || NA == 2 || NA == 4, "1 (64-channel blocks), 2 or 4 filter fragments per wave");
  static_assert(MB >= 2 && MB <= 4, "8, 12 or 16 tile rows");
  constexpr int TH = 4 * MB;                       // tile rows
  constexpr int T3_HUNITS = t3_hunits<MB>(), T3_NHP = t3_nhp<MB>(), T3_HALO_BYTES = t3_halo_bytes<MB>();
  constexpr int NM = NA * MB;                      // MFMAs per substep (NA filter x MB pixel fragments)
  constexpr int NR = NA + MB;                      // fragment reads per substep
  static_assert(9 % G == 0 && (9 / G) % NSLOT == 1, "stage s lives in slot s % NSLOT == (chunk + stage in chunk) % NSLOT");
  constexpr int WCO = BN / (NA * 32);
  constexpr int SPC = 9 / G;                       // stages per chunk
  constexpr int NQ = 2 * G;                        // substeps per stage
  constexpr int D = NSLOT - 1;                     // the DMA runs D stages ahead
  constexpr int FP = BN / 16 / NW;                 // filter pieces per tap and wave (2)
  constexpr int HPW = (T3_NHP + NW - 1) / NW;      // halo pieces per chunk and wave
  constexpr int SLOT_BYTES = G * BN * 64;
  constexpr int HPS = (HPW + SPC - 1) / SPC > 1 ? (HPW + SPC - 1) / SPC : 1;   // halo pieces a wave issues per stage
  static_assert((HPW + HPS - 1) / HPS + D <= SPC + 1, "halo pieces must land before their chunk starts");
  static_assert(FP >= 1 && FP <= NM && HPS + 1 <= NM, "one DMA piece per MFMA slot at most");

  HIP_DYNAMIC_SHARED(char, smem)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wpx = wave / WCO, wco = wave % WCO;
  const int l31 = lane & 31, hi = lane >> 5, l15 = lane & 15, lrow = (lane >> 4) & 1;

  const fsr_lds_addr_t halo_addr = FSR_LDS_ADDR(smem);
  const fsr_lds_addr_t ring_addr = halo_addr + 2 * T3_HALO_BYTES;
  const fsr_buf_t in_buf = fsr_make_buf(a.in, (unsigned)((size_t)a.N * a.IH * a.IW * a.Cin * sizeof(T)));
  const fsr_buf_t w_buf = fsr_make_buf(a.wpk, (unsigned)((size_t)9 * a.CoutPad * a.Cin * sizeof(T)));
  const int nchunks = a.Cin >> 5;

  // ---- loop-invariant per-lane addresses -------------------------------------------------------------------------
  // filter fragment (n, k half j) of tap g of the stage in ring slot `sl`: ring + sl + g*BN*64 + n*2048 + aoff[j]
  unsigned aoff[2];
  {
    const int R = wco * (NA * 32) + l31;
#pragma unroll
    for (int j = 0; j < 2; ++j) aoff[j] = (unsigned)(2 * T3_HALO_BYTES + R * 64 + (((2 * j + hi) ^ t3_swz_row(R)) << 4));
  }
  // pixel fragment m of tap (ky, kx), k half j, halo buffer hb: hb*HALO + (2m + ky)*ROWB + boff[kx][j]
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
    wvoff[k] = (unsigned)((co * a.Cin + ((ul ^ t3_swz_row(R)) << 3)) * (int)sizeof(T));
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
    ws = (unsigned)(tc.nb * BN * a.Cin * (int)sizeof(T));
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
      const int U = (wave + k * NW) * 64 + lane;
      const int hp = U >> 2, ul = U & 3;
      const int hy = hp / T3_P, hx = hp - hy * T3_P;
      const int iy = tc.gy0 - 1 + hy, ix = tc.gx0 - 1 + hx;
      unsigned o = ~0u;                                          // beyond the buffer: the DMA writes zeros
      if (U < T3_HUNITS && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW)
        o = (unsigned)((((tc.img * a.IH + iy) * a.IW + ix) * a.Cin + ((ul ^ t3_swz_col(hx)) << 3)) * (int)sizeof(T));
      hv[k] = o;
    }
  };
  // DMA pieces.  Halo piece k of tile-chunk c -> halo buffer (global chunk parity); filter pieces of tap t of chunk c -> ring
  auto dma_halo = [&](const unsigned (&hv)[HPW], int c, int k, unsigned par) {
    if (wave + k * NW < T3_NHP && !(a.t3_dbg & 2))
      FSR_BLDS16(in_buf, hv[k], (unsigned)(c * 64), halo_addr + (fsr_lds_addr_t)(par * T3_HALO_BYTES + (wave + k * NW) * 1024));
  };
  auto dma_filter = [&](unsigned ws, int c, int t, int k, unsigned slot_tap_off) {
    if (!(a.t3_dbg & 2)) FSR_BLDS16(w_buf, wvoff[k], a.t3_woff[t] + ws + (unsigned)(c * 64),
               ring_addr + (fsr_lds_addr_t)(slot_tap_off + (wave + k * NW) * 1024));
  };
  // The bias of a tile's channel block comes by DMA as well (one piece of BN floats, wave 0, double buffered by tile parity):
  // an ordinary global load next to the epilogue's stores would make hipcc drain vmcnt -- the whole DMA pipeline -- per tile.
  constexpr int BIAS_BYTES = BN * 4;                       // <= 1024: one piece
  const fsr_lds_addr_t bias_addr = ring_addr + NSLOT * SLOT_BYTES;
  const fsr_buf_t bias_buf = fsr_make_buf(a.bias, a.bias ? (unsigned)(a.Cout * 4) : 0u);
  auto dma_bias = [&](int nbk, unsigned par) {
    if (wave == 0 && a.bias && !(a.t3_dbg & 2)) {
      const unsigned vo = lane * 16 < BIAS_BYTES ? (unsigned)(nbk * BIAS_BYTES + lane * 16) : ~0u;
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
      if (a.bias) {
        const unsigned bo = (unsigned)(2 * T3_HALO_BYTES + NSLOT * SLOT_BYTES) + par * 1024 + (unsigned)((wco * (NA * 32) + n * 32 + hi * 16) * 4);
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
    constexpr int r = decltype(rc)::value, buf = decltype(bufc)::value, t = decltype(tc)::value, g = decltype(gc_)::value,
                  j = decltype(jc)::value;
    constexpr int ky = t / 3, kx = t % 3;
    if constexpr (r == 0 || r > MB) {      // a0 first, a1 .. a(NA-1) after the pixel fragments
      constexpr int n = r == 0 ? 0 : r - MB;
      fa[buf][n] = t3_lds_read<s16x8>(smem, aoff[j] + sl + (unsigned)(g * BN * 64 + n * 2048));
    } else {
      constexpr int m = r - 1;
      fb[buf][m] = t3_lds_read<s16x8>(smem, boff[kx][j] + hb + (unsigned)((2 * m + ky) * T3_ROWB));
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
  unsigned tpar = 0;               // tile parity (bias buffer)
  dma_bias(cur.nb, tpar);
#pragma unroll
  for (int k = 0; k < HPW; ++k) dma_halo(hv_cur, 0, k, 0u);
  static_for<0, D>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    static_assert(D < SPC, "the first D stages lie in chunk 0");
    static_for<0, G>([&](auto gc_) {
      constexpr int g = decltype(gc_)::value;
#pragma unroll
      for (int k = 0; k < FP; ++k) dma_filter(ws_cur, 0, s * G + g, k, slot_of(0, s) + g * BN * 64);
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

  for (;;) {
    for (int c = 0; c < ((a.t3_dbg & 4) ? 0 : nchunks); ++c, ++gc) {
      const bool last = c + 1 == nchunks;
      if (last && has_nxt) {         // from here on the DMA feeds the next tile
        setup(logical(next), nxt, hv_nxt, ws_nxt);
        dma_bias(nxt.nb, tpar ^ 1u);
      }
      static_for<0, SPC>([&](auto sic) {
        constexpr int si = decltype(sic)::value;
        const unsigned sl = slot_of(gc, si);
        const unsigned hb = (unsigned)((gc & 1) * T3_HALO_BYTES);
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
        const unsigned hbN = (unsigned)(((gc + dcN) & 1) * T3_HALO_BYTES);

        static_for<0, NQ>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          constexpr int buf = q & 1;
          if constexpr (q == NQ - 1) {
            // publish the next stage: this wave's pieces of stage s+1 have landed when at most the pieces of the D-1 younger
            // stages are outstanding.  Loads retire in order, so halo / bias pieces and the previous tile's stores in the
            // queue can only make the wait longer, never shorter.
            if (a.t3_dbg & 64) {
              // EXPERIMENT (wrong results): no wait for the DMA
            } else if constexpr (D == 1) {
              FSR_WAIT_VM(0);
            } else {
              static_assert(D <= 3, "add cases for deeper rings");
              const int young = (issue ? 1 : 0) + (D == 3 ? issued_prev : 0);
              if (young == 0) FSR_WAIT_VM(0);
              else if (young == 1) FSR_WAIT_VM(G * FP);
              else FSR_WAIT_VM(2 * G * FP);
            }
            if (a.t3_dbg & 16) FSR_WAIT_LGKM0();   // (strict form; the reads of a slot are >= 400 cycles older than any DMA into it)
            if (!(a.t3_dbg & 32)) FSR_BARRIER();   // (EXPERIMENT bit 32: no barrier, wrong results)
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
            if constexpr (q == 0 && i >= 1 && i <= HPS && si * HPS + (i - 1) < HPW) {
              constexpr int hk = si * HPS + (i - 1);
              if (!last) dma_halo(hv_cur, c + 1, hk, (unsigned)((gc + 1) & 1));
              else if (has_nxt) dma_halo(hv_nxt, 0, hk, (unsigned)((gc + 1) & 1));
            }
            if constexpr (q < G && i >= NM - FP) {
              if (issue) dma_filter(wsD, cD, siD * G + q, i - (NM - FP), slD + q * BN * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
          });
        });
        issued_prev = issue ? 1 : 0;
      });
    }

    // ---- epilogue -----------------------------------------------------------------------------------------------------
    // activation as ONE instruction per element: ReLU = max(v, 0), LeakyReLU = max(v, slope * v) (0 <= slope <= 1: host
    // checked), identity = nothing.  The three forms are separate instantiations of the store loop (one wave-uniform branch).
    T* outp = (T*)a.out;
    const T* maskp = (const T*)a.dmask;
    const int oimg = cur.img, ogy0 = cur.gy0, ogx0 = cur.gx0, onb = cur.nb;
    const int gx = ogx0 + l15;
    auto store_tile = [&](auto actc) {
      constexpr int ACT = decltype(actc)::value;
      const float slope = a.slope;
      auto activate = [&](float x) {
        if constexpr (ACT == FSR_ACT_RELU) return fmaxf(x, 0.f);
        else if constexpr (ACT == FSR_ACT_LEAKY) return fmaxf(x, x * slope);
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
            if (ok && lrow == 0 && !(l15 & 1) && !(a.t3_dbg & 1)) {
              const unsigned off = (unsigned)((oimg * (a.FOH >> 1) + (gy >> 1)) * (a.FOW >> 1) + (gx >> 1)) * (unsigned)a.Cout + (unsigned)co;
              u32x4 p0, p1;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                p0[e] = pack2<T>(v[2 * e], v[2 * e + 1]);
                p1[e] = pack2<T>(v[8 + 2 * e], v[8 + 2 * e + 1]);
              }
              *(u32x4*)(outp + off) = p0;
              *(u32x4*)(outp + off + 8) = p1;
            }
          } else if (ok && !(a.t3_dbg & 1)) {
            const unsigned off = (unsigned)((oimg * a.FOH + gy) * a.FOW + gx) * (unsigned)a.Cout + (unsigned)co;

            if (maskp) {   // fused activation backward of the producing layer: dz = dx * act'(y), y = the saved forward input
              const u32x4 k0 = *(const u32x4*)(maskp + off), k1 = *(const u32x4*)(maskp + off + 8);
              const float ms = a.dmask_slope;
              if (a.dmask_add) {   // the tensor is an addend (gradient of a skip connection), not a gate
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
            u32x4 p0, p1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              p0[e] = pack2<T>(activate(v[2 * e]), activate(v[2 * e + 1]));
              p1[e] = pack2<T>(activate(v[8 + 2 * e]), activate(v[8 + 2 * e + 1]));
            }
            *(u32x4*)(outp + off) = p0;
            *(u32x4*)(outp + off + 8) = p1;
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

int t3_mode() {
  // FSR_TALL3: 0 = off (A/B against conv_igemm.hip); 1 (default) = on, 4-wave workgroups of 256 px x 128 channels, two per
  // CU; 3 = the 8-wave 256 px x 256 channel workgroup (one per CU, persistent) where Cout allows it.  Measured per layer:
  // the 4-wave form wins on every benched shape but one (profiles/r03_conv_tall3_ab.txt).  Bit 4: the one-wave-per-SIMD form;
  // bit 8: no 64-channel block (Cout = 64 launches stay on conv_igemm.hip).
  const char* e = getenv("FSR_TALL3");
  return e ? atoi(e) : 1;
}

int t3_cus() {
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

template <typename T, int BN, int NW, int G, int NSLOT, int MB, int NA = 2, bool STATS = false>
int t3_launch(ConvKArgs& a, int wg_per_cu, hipStream_t stream) {
  auto kern = conv_tall3_kernel<T, BN, NW, G, NSLOT, MB, NA, STATS>;
  if (STATS) {   // one partial slot per tile and pixel-row group of waves
    a.stats_P = a.tiles_x * a.tiles_y * (NW / (BN / (NA * 32)));
    a.stats_tpi = a.stats_per = 0;
  }
  constexpr int lds = t3_lds_bytes<BN, G, NSLOT, MB>();
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  a.nblk_n = a.Cout / BN;
  const long long ntiles = (long long)a.tiles_x * a.tiles_y * a.N * a.nblk_n;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) return 0;
  a.t3_ntiles = (int)ntiles;
  a.t3_dbg = getenv("FSR_T3_DBG") ? atoi(getenv("FSR_T3_DBG")) : 0;
  // persistent tile walk, wg_per_cu workgroups per CU; FSR_T3_PERSIST=0 launches one workgroup per tile instead (A/B: the
  // hardware dispatcher balances better, but every tile then pays a cold prologue)
  static const bool persist = !(getenv("FSR_T3_PERSIST") && atoi(getenv("FSR_T3_PERSIST")) == 0);
  long long grid = (long long)t3_cus() * wg_per_cu;
  if (grid > ntiles || !persist) grid = ntiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, stream, a);
  fsr_note_kernel(STATS ? "conv_tall3_kernel<%s,%d,%d,%d,%d,%d,%d,stats>" : "conv_tall3_kernel<%s,%d,%d,%d,%d,%d,%d>", std::is_same<T, f16_t>::value ? "f16" : "bf16", BN, NW, G, NSLOT, MB, NA);
  const int rc = fsr_check_launch("conv_tall3_kernel");
  return rc ? rc : 1;
}

}  // namespace

// 1 = launched, 0 = not this kernel's shape (the caller falls through to conv_igemm.hip), < 0 = error.
int fsr_conv_tall3_try(int dtype, ConvKArgs& a, int S, hipStream_t stream) {
  const int mode = t3_mode();
  if (mode == 0 || (dtype != FSR_BF16 && dtype != FSR_F16) || S != 1 || a.ntaps != 9) return 0;
  if (a.Cin < 128 || a.Cin % 32 != 0 || a.Cout % 64 != 0 || a.CoutPad != a.Cout) return 0;
  // Cout = 64 (the data gradient of a 64 -> 128 layer): 64-channel blocks, a wave = 32 MB pixels x 32 channels (one filter
  // fragment, 5 reads per 4 MFMAs); FSR_TALL3 & 8 keeps these launches on conv_igemm.hip (A/B)
  const bool narrow = a.Cout % 128 != 0;
  if (narrow && ((mode & 14) || a.stats || a.pool2)) return 0;
  if (a.preact || a.oscale || a.ps || a.in_ps || a.out_f32) return 0;
  if (a.stats && (a.pool2 || a.dmask || (mode & 6))) return 0;   // statistics: the shipped 4-wave form, forward launches
  if (a.act != FSR_ACT_NONE && a.act != FSR_ACT_RELU && a.act != FSR_ACT_LEAKY) return 0;
  if (a.act == FSR_ACT_LEAKY && !(a.slope >= 0.f && a.slope <= 1.f)) return 0;   // the epilogue's max(v, slope * v) form
  if (a.osy != 1 || a.osx != 1 || a.ooy != 0 || a.oox != 0 || a.org_y != -1 || a.org_x != -1) return 0;
  if (a.pool2 && (a.dmask || (a.GH & 1) || (a.GW & 1))) return 0;
  if ((long long)a.N * a.IH * a.IW * a.Cin >= (1LL << 31) || (long long)a.N * a.FOH * a.FOW * a.Cout >= (1LL << 31)) return 0;
  // canonical tap order (ky, kx): which filter slice serves the tap that reads halo offset (ky, kx)
  int slice[9];
  for (int t = 0; t < 9; ++t) slice[t] = -1;
  for (int t = 0; t < 9; ++t) {
    if (a.tdy[t] < 0 || a.tdy[t] > 2 || a.tdx[t] < 0 || a.tdx[t] > 2) return 0;
    slice[a.tdy[t] * 3 + a.tdx[t]] = a.tw[t];
  }
  for (int t = 0; t < 9; ++t) {
    if (slice[t] < 0) return 0;
    a.t3_woff[t] = (unsigned)((size_t)slice[t] * a.CoutPad * a.Cin * 2);
  }
  a.tiles_x = (a.GW + 15) / 16;
  const bool wide = a.Cout % 256 == 0 && (mode & 2);
  // Tile height (16, 12 or 8 rows): the tallest one that pads the map least -- 16 rows everywhere except the 24-row maps
  // (two 12-row tiles; 16-row tiles would compute 32 rows: measured 906 against 752 TFLOP/s on 512 -> 512 @ 24^2).  Whole
  // tile rounds do NOT decide: 12-row tiles give 512 -> 512 @ 48^2 exactly 3 rounds instead of 2.25 and still measured
  // 0..4 % slower -- a workgroup whose partner has finished runs faster alone, and the shorter tile pays more staging per
  // MFMA (profiles/r03_conv_tall3_ab.txt).  FSR_T3_ROWS forces a height (A/B, tests).
  int best_mb = 4, best_rows = 1 << 30;
  const int forced = getenv("FSR_T3_ROWS") ? atoi(getenv("FSR_T3_ROWS")) : 0;
  for (int mb = 4; mb >= 2; --mb) {
    const int th = 4 * mb;
    if (forced && forced != th) continue;
    const int rows = (a.GH + th - 1) / th * th;
    if (rows < best_rows) { best_rows = rows; best_mb = mb; }
  }
  a.tiles_y = (a.GH + 4 * best_mb - 1) / (4 * best_mb);
  // FSR_TALL3 & 4 (A/B): the one-wave-per-SIMD form, 128 x 128 per wave (16-row tiles only)
  if ((mode & 4) && a.Cout % 256 == 0 && best_mb == 4 && dtype == FSR_BF16) return t3_launch<bf16_t, 256, 4, 3, 2, 4, 4>(a, 1, stream);
#define T3_GO(TT, MBV)                                                                      \
  do {                                                                                      \
    if (narrow) return t3_launch<TT, 64, 4, 1, 4, MBV, 1>(a, 2, stream);                    \
    if (a.stats) return t3_launch<TT, 128, 4, 1, 4, MBV, 2, true>(a, 2, stream);            \
    if (wide) return t3_launch<TT, 256, 8, 3, 2, MBV>(a, 1, stream);                        \
    return t3_launch<TT, 128, 4, 1, 4, MBV>(a, 2, stream);                                  \
  } while (0)
  if (dtype == FSR_F16) {
    if (best_mb == 4) T3_GO(f16_t, 4);
    if (best_mb == 3) T3_GO(f16_t, 3);
    T3_GO(f16_t, 2);
  }
  if (best_mb == 4) T3_GO(bf16_t, 4);
  if (best_mb == 3) T3_GO(bf16_t, 3);
  T3_GO(bf16_t, 2);
#undef T3_GO
}
