// 3x3 convolution as an implicit GEMM on the gfx950 matrix cores (NHWC activations).
//
// Replaces the torch.nn.Conv2d(k=3, p=1) calls of the reference on the hot path:
//   /root/reference/model.py:47-64  (ResidualBlock conv1/conv2, 64->64, no bias)
//   /root/reference/model.py:86-93  (bottleneck conv)
//   /root/reference/model.py:30-40  (UpSamplingBlock conv 64->256 + PixelShuffle(2) [+PReLU])
//   /root/reference/model.py:102-109 (head conv 64->3 + tanh; Cout padded to one 16-wide tile)
//   /root/reference/model.py:124-131,148-183 (Discriminator SimpleBlock convs, stride 1 and 2)
//   torchvision vgg19.features[2:34] convs used by /root/reference/model.py:8
// and, with other tap tables, the data-gradients (dgrad) of all of the above.
//
// Work decomposition
//   workgroup (WM x WN waves: 4; 8 also compiles) -> TH x 16 output pixels of one image x BN output channels
//   wave                              -> MT pixel rows (16 px each) x NT 16-wide channel tiles
//   K loop                            -> (input-channel chunk of KC) x (tap).  Three main-loop forms, by template parameter:
//        G = 1, DMA = 0 : one tap per step, filter slices register-staged two steps ahead, one barrier per step;
//        G = 2 / 3      : G taps per stage, slices single-buffered in LDS and prefetched into registers, two barriers per
//                         stage (stride-2 forward, 64-wide configurations, the classes of a stride-2 data gradient);
//        DMA = 1        : slices by LDS-DMA (global_load_lds_dwordx4) into a double buffer, XOR-swizzled unpadded rows,
//                         one barrier per stage (the tall 128..512-channel configuration).
// LDS images
//   halo : [(TH-1)*S+3][15*S+3][KC] input pixels of the current chunk, loaded ONCE per chunk and re-read by every
//          tap (9x reuse out of LDS instead of L2).  The loads of chunk c+1 are issued into registers at the start
//          of chunk c (per-thread offsets are loop invariant) and committed to LDS after its last tap;
//   wl   : [2][BN][KC] filter slices of the current / next step.  Prefetch distance is two steps: the global loads
//          of step s+2 go into one of two register sets before the MFMAs of step s and reach LDS after the MFMAs of
//          step s+1;
//   pitches are conflict-free for ds_read_b128 (see PITCHW / PITCHX below).
// Two independent 4-wave workgroups per CU overlap each other's barriers and staging (8-wave workgroups and a
// single-launch stride-2 data gradient were measured and lost 5-10 %).
// MFMA mapping: D[row = output channel][col = pixel] = W[cout][k] * X[k][pixel], so each lane ends
// up with 4 consecutive output channels of one pixel and the NHWC store is a 8/16-byte vector.
// K index permutation inside a chunk is free as long as both operands agree; both operands are
// read as one 16-byte vector per lane: lane group g = lane>>4 owns channels [8g,8g+8) (bf16, one
// v_mfma_f32_16x16x32_bf16 per 32 channels) or [4g,4g+4) (f32, four v_mfma_f32_16x16x4_f32, MFMA j
// consuming channel 4g+j).
#include "fsr_common.h"
#include "fsr_conv_args.h"
#include "fsr_host.h"

#include <stdlib.h>
#include <utility>

#ifndef FSR_ABL
#define FSR_ABL 0    // ablation builds (tools/ablate.sh): 1 no filter DMA after the first stage, 2 no halo reloads, 3 neither (WRONG RESULTS)
#endif
#ifndef FSR_PIPE
#define FSR_PIPE 1   // software-pipelined fragment reads (0: A/B builds of the plain main loops, tools/ab_lib.sh)
#endif

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef s16x8 type; };
template <> struct Frag<f16_t> { typedef s16x8 type; };
template <> struct Frag<float> { typedef f32x4 type; };

// tap t of the launch: bits 0-1 halo row offset, bits 2-3 halo column offset, bits 4-7 filter slice
__device__ __forceinline__ unsigned tap_code(const ConvKArgs& a, int t) {
  return (t < 8) ? (unsigned)((a.taps_lo >> (8 * t)) & 0xffull) : a.taps_hi;
}

// ---------------------------------------------------------------------------- epilogue
// The epilogue runs once per workgroup over MT*NT accumulator tiles; with a few thousand MFMA cycles of
// main loop per workgroup its instruction count matters as much as the loop's.  Three compact paths, picked
// by a wave-uniform branch, replace one path with per-element flag tests:
//   plain : T output, every channel valid (Cout % 16 == 0), optional pre-activation copy   (all big layers)
//   ps    : as plain, stored depth-to-space (PixelShuffle(2) fused, model.py:36)
//   thin  : float output with fewer than 16 channels per tile (head conv + tanh, image gradients)
// ReLU / LeakyReLU / PReLU / identity are ONE formula: v > 0 ? v : v * slope with slope 0 / s / a / 1.
template <typename T>
__device__ __forceinline__ void store_vec4(T* p, f32x4 v) {
  if constexpr (std::is_same<T, x3_t>::value) {
    x3_st4(p, v);
  } else if constexpr (sizeof(T) == 4) {
    *(f32x4*)p = v;
  } else {
    u32x2 pk;
    pk.x = pack2<T>(v[0], v[1]);
    pk.y = pack2<T>(v[2], v[3]);
    *(u32x2*)p = pk;
  }
}

// X3 = 1 (FSR_X3, T = bf16_t, KC = 64): the input is an x3 tensor seen as a bf16 tensor of a.Cin = 2 x logical channels whose
// 32-channel chunks alternate hi / lo, the filter pack has the same K order ([w_hi | w_lo] per 64), so staging, LDS images and
// addressing are exactly the bf16 kernel's at twice the channels.  Only two things differ: a 64-channel chunk of a tap is
// THREE MFMAs per fragment pair -- w_hi x_hi + w_hi x_lo + w_lo x_hi (4 fragment reads per 3 MFMAs; the bf16 form reads 4 per 2)
// -- and the epilogue stores / reads x3 elements (hi and lo 64 bytes apart, fsr_common.h).
template <typename T, int TH, int BN, int WM, int WN, int KC, int S, int G = 1, int DMA = 0, int X3 = 0>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_igemm_kernel(const ConvKArgs a_in, const ConvKClasses cls) {
  static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per workgroup");
  static_assert(!X3 || (std::is_same<T, bf16_t>::value && KC == 64 && G == 1 && DMA == 0), "x3: bf16 planes, 64-channel (hi | lo) chunks");
  typedef typename std::conditional<X3 != 0, x3_t, T>::type ST;   // storage type of the output-side tensors
  constexpr int NTHR = WM * WN * 64;
  constexpr int MT = TH / WM;
  constexpr int NT = BN / 16 / WN;
  constexpr int EPB = 16 / (int)sizeof(T);  // elements per 16-byte unit
  // LDS pitches in elements.  ds_read_b128 serves a wave in four 16-lane groups over 16 slots of 16 bytes; a
  // fragment read puts lane (l&15, l>>4) at slot (l&15)*stride*P + (l>>4) (P = pitch in slots), which is conflict-free
  // for P = 2 mod 4 when consecutive lanes are consecutive rows (filter rows, stride-1 pixels) and for odd P when
  // they are every second pixel (stride 2).  A plain +16 B pad (P = 9 for 64 bf16 channels) is 2-way conflicted.
  constexpr int PITCHW = KC + 2 * EPB;
  constexpr int PITCHX = KC + (S == 2 ? 1 : 2) * EPB;
  constexpr int UNITS = KC / EPB;
  constexpr int KSTEP = (sizeof(T) == 2) ? 32 : 16;  // channels consumed per operand vector pair
  constexpr int WPT = (BN * UNITS + NTHR - 1) / NTHR;
  typedef typename Frag<T>::type frag_t;

  HIP_DYNAMIC_SHARED(char, smem)
  T* halo = (T*)smem;
  constexpr int HH = (TH - 1) * S + 3, HW = 15 * S + 3;  // halo extent for the full 3x3 footprint (fewer taps use less)
  T* wl = halo + HH * HW * PITCHX;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: addresses derived from it stay scalar
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, lg = lane >> 4;

  // class selection (wave-uniform scalar selects; a single-class launch uses the arguments as they are)
  ConvKArgs a = a_in;
  int bid0 = (int)blockIdx.x, nwg = (int)gridDim.x;
  if (cls.n > 1) {
    ConvKClass k = cls.c[0];
    int first = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (i < cls.n && bid0 >= cls.c[i - 1].wg_end) {
        k = cls.c[i];
        first = cls.c[i - 1].wg_end;
      }
    a.GH = k.GH; a.GW = k.GW; a.ntaps = k.ntaps; a.ooy = k.ooy; a.oox = k.oox;
    a.tiles_x = k.tiles_x; a.tiles_y = k.tiles_y; a.taps_lo = k.taps_lo;
    bid0 -= first;
    nwg = k.wg_end - first;
  }
  int bid = xcd_remap(bid0, nwg);
  const int nb = bid % a.nblk_n;
  bid /= a.nblk_n;
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int img = bid / a.tiles_y;

  const int gy0 = ty * TH, gx0 = tx * 16;
  const int iy0 = gy0 * S + a.org_y, ix0 = gx0 * S + a.org_x;
  const T* in = (const T*)a.in;
  const T* wpk = (const T*)a.wpk;
  const int nchunks = a.Cin / KC;

  // Halo staging is split (issue early / commit late): the global loads of chunk c+1 are issued into
  // registers at the start of chunk c and written to LDS after its last tap, so their latency hides
  // under a whole chunk of MFMAs instead of stalling every workgroup once per chunk.
  constexpr int halo_total = HH * HW * UNITS;
  constexpr int HPT = (halo_total + NTHR - 1) / NTHR;  // 16-byte units per thread
  u32x4 hreg[HPT];
  // Per-thread element offset of each of its halo units for chunk 0 (loop invariant: pixel, bounds test and
  // channel unit depend only on the thread), ~0u outside the image (zero padding).  A chunk adds a scalar.
  unsigned hoff[HPT];
#pragma unroll
  for (int i = 0; i < HPT; ++i) {
    const int u = tid + i * NTHR;
    unsigned o = ~0u;
    if (u < halo_total) {
      const int unit = u % UNITS;
      const int p = u / UNITS;
      const int hx = p % HW, hy = p / HW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      if (iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW) {
        if (!a.in_ps) o = (unsigned)((img * a.IH + iy) * a.IW + ix) * (unsigned)a.Cin + (unsigned)(unit * EPB);
        else o = (unsigned)((img * 2 * a.IH + 2 * iy) * (2 * a.IW) + 2 * ix) * (unsigned)(a.Cin >> 2) + (unsigned)(unit * EPB);
      }
    }
    hoff[i] = o;
  }
  auto halo_issue = [&](int c) {
    unsigned add = (unsigned)(c * KC);
    if (a.in_ps) {  // depth-to-space input: a chunk lies inside one quadrant (host checked: (Cin/4) % KC == 0)
      const int cps = a.Cin >> 2;
      const int q = (c * KC) / cps;
      add = (unsigned)(((q >> 1) * 2 * a.IW + (q & 1)) * cps + (c * KC - q * cps));
    }
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
      u32x4 v = (u32x4){0u, 0u, 0u, 0u};
      if (hoff[i] != ~0u) v = *(const u32x4*)(in + (hoff[i] + add));
      hreg[i] = v;
    }
  };
  auto halo_commit = [&]() {
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
      const int u = tid + i * NTHR;
      if (u < halo_total) *(u32x4*)(halo + (size_t)(u / UNITS) * PITCHX + (u % UNITS) * EPB) = hreg[i];
    }
  };

  // Filter-slice staging, prefetch distance TWO steps (a step = one tap of one channel chunk): the global
  // loads of step s+2 are issued into one of two register sets before the MFMAs of step s and written to the
  // LDS buffer of step s+2 one step later, after the MFMAs of step s+1 -- two MFMA blocks (>= 1000 cycles) of
  // cover for an L2 round trip, against one barrier per step.
  u32x4 wregA[WPT], wregB[WPT];
  // per-thread element offsets inside a filter slice (loop invariant), and the slice geometry in 32 bits
  unsigned woff[WPT];
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int u = tid + i * NTHR;
    woff[i] = (unsigned)((u / UNITS) * a.Cin + (u % UNITS) * EPB);
  }
  const unsigned slice_stride = (unsigned)(a.CoutPad * a.Cin);   // elements per tap slice (< 2^31: host checked)
  const T* wnb = wpk + (size_t)nb * BN * a.Cin;
  auto wload = [&](u32x4 (&wr)[WPT], int c, int t) {
    const T* base = wnb + ((tap_code(a, t) >> 4) * slice_stride + (unsigned)(c * KC));
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int u = tid + i * NTHR;
      u32x4 v = (u32x4){0u, 0u, 0u, 0u};
      if ((BN * UNITS) % NTHR == 0 || u < BN * UNITS) v = *(const u32x4*)(base + woff[i]);
      wr[i] = v;
    }
  };
  auto wstore = [&](const u32x4 (&wr)[WPT], int buf) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int u = tid + i * NTHR;
      if ((BN * UNITS) % NTHR == 0 || u < BN * UNITS)
        *(u32x4*)(wl + ((size_t)buf * BN + (u / UNITS)) * PITCHW + (u % UNITS) * EPB) = wr[i];
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int pixbase[MT], wbase[NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) pixbase[m] = (((wm * MT + m) * S) * HW + l15 * S) * PITCHX + lg * EPB;
#pragma unroll
  for (int n = 0; n < NT; ++n) wbase[n] = ((wn * NT + n) * 16 + l15) * (DMA ? KC : PITCHW) + (DMA ? 0 : lg * EPB);
  // DMA filter images are unpadded rows of UNITS 16-byte units, the unit index XOR-swizzled by the row so that the
  // four 16-lane groups of a ds_read_b128 fragment read hit 16 different slots (checked exhaustively for both widths)
  auto swz = [](int row) { return UNITS == 4 ? ((row & 8) ? 3 : 0) : ((row >> 1) & 7); };
  int wsw[KC / KSTEP];
#pragma unroll
  for (int ks = 0; ks < KC / KSTEP; ++ks) wsw[ks] = DMA ? ((ks * (KSTEP / EPB) + lg) ^ swz(l15)) * EPB : ks * KSTEP;

  // One step.  P = step parity: LDS buffer P holds this step's filter slice; register set `mine` receives the
  // loads of the step two ahead (same parity: chunk c2, tap t2), register set `other` holds the next step's slice,
  // loaded one step ago.  All step bookkeeping is incremental scalar arithmetic (no division in the loop).
  // the MFMAs of one tap: filter slice at `wcur` in LDS, halo window shifted by the tap's offset
  auto tap_mfma = [&](const T* wcur, int t) {
    const unsigned tc = tap_code(a, t);
    const int toff = ((int)(tc & 3u) * HW + (int)((tc >> 2) & 3u)) * PITCHX;
    if constexpr (X3) {
      frag_t xh[MT], xl[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        xh[m] = *(const frag_t*)(halo + pixbase[m] + toff);
        xl[m] = *(const frag_t*)(halo + pixbase[m] + toff + KSTEP);
      }
      constexpr int NH = NT > 4 ? 4 : NT;
#pragma unroll
      for (int n0 = 0; n0 < NT; n0 += NH) {
        frag_t wh[NH], wlo[NH];
#pragma unroll
        for (int n = 0; n < NH; ++n) {
          wh[n] = *(const frag_t*)(wcur + wbase[n0 + n] + wsw[0]);
          wlo[n] = *(const frag_t*)(wcur + wbase[n0 + n] + wsw[1]);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NH; ++n) {
            acc[m][n0 + n] = mfma16<T>(wh[n], xh[m], acc[m][n0 + n]);
            acc[m][n0 + n] = mfma16<T>(wh[n], xl[m], acc[m][n0 + n]);
            acc[m][n0 + n] = mfma16<T>(wlo[n], xh[m], acc[m][n0 + n]);
          }
        if constexpr (NH < NT) __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
#pragma unroll
    for (int ks = 0; ks < KC / KSTEP; ++ks) {
      // filter fragments in groups of NH tiles: with multi-tap stages the 8-tile waves would otherwise hold 12
      // fragments (48 registers) next to 128 accumulators and the staging registers, and spill
      constexpr int NH = (G > 1 && NT > 4) ? 4 : NT;
      frag_t xf[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) xf[m] = *(const frag_t*)(halo + pixbase[m] + toff + ks * KSTEP);
#pragma unroll
      for (int n0 = 0; n0 < NT; n0 += NH) {
        frag_t wf[NH];
#pragma unroll
        for (int n = 0; n < NH; ++n) wf[n] = *(const frag_t*)(wcur + wbase[n0 + n] + wsw[ks]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NH; ++n) {
            if constexpr (sizeof(T) == 2) {
              acc[m][n0 + n] = mfma16<T>(wf[n], xf[m], acc[m][n0 + n]);
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[m][n0 + n] = mfma_f32_16x16x4(wf[n][j], xf[m][j], acc[m][n0 + n]);
            }
          }
        if constexpr (NH < NT) __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // The three taps of one LDS-DMA stage with the fragment reads SOFTWARE-PIPELINED (bf16 tall configuration).  The plain
  // form above compiles to  {8 or 4 ds_read_b128 -> s_waitcnt lgkmcnt(0) -> 16 MFMAs}  six times per stage: every block of 16
  // MFMAs (256 matrix-pipe cycles) is preceded by a full LDS round trip during which the wave issues nothing.  Here a stage is
  // twelve groups of 8 MFMAs (one tap x two 16-channel tiles x four pixel rows); the three reads a group issues -- the next
  // group's two filter fragments and one of the next tap's four pixel fragments -- go out BEFORE its MFMAs and are waited
  // for (counted lgkmcnt) only one group later, so LDS latency hides under 128+ cycles of matrix work.  Fragment registers:
  // 2 x 4 pixel + 2 x 2 filter = 48 (the plain form holds 32).
  auto stage_mfma3 = [&](const T* wstage, int g) {
    if constexpr (FSR_PIPE && DMA && G == 3 && MT == 4 && NT == 8 && sizeof(T) == 2 && KC == KSTEP) {
      int toff[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const unsigned tc = tap_code(a, g * 3 + j);
        toff[j] = ((int)(tc & 3u) * HW + (int)((tc >> 2) & 3u)) * PITCHX;
      }
      frag_t xf[2][MT], wf[2][2];
#pragma unroll
      for (int m = 0; m < MT; ++m) xf[0][m] = *(const frag_t*)(halo + pixbase[m] + toff[0]);
#pragma unroll
      for (int n = 0; n < 2; ++n) wf[0][n] = *(const frag_t*)(wstage + wbase[n] + wsw[0]);
      static_for<0, 12>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int t = i / 4, h = i % 4;
        if constexpr (i + 1 < 12) {
          constexpr int tn = (i + 1) / 4, hn = (i + 1) % 4;
#pragma unroll
          for (int n = 0; n < 2; ++n) wf[(i + 1) & 1][n] = *(const frag_t*)(wstage + (size_t)tn * BN * KC + wbase[hn * 2 + n] + wsw[0]);
        }
        if constexpr (t + 1 < 3) xf[(t + 1) & 1][h] = *(const frag_t*)(halo + pixbase[h] + toff[t + 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][h * 2 + n] = mfma16<T>(wf[i & 1][n], xf[t & 1][m], acc[m][h * 2 + n]);
        __builtin_amdgcn_sched_barrier(0);
      });
    } else {
      static_for<0, G>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        tap_mfma(wstage + (size_t)(j * BN) * KC, g * G + j);
      });
    }
  };

  auto step_body = [&](auto parity, int c, int t, bool has1, bool has2, int c2, int t2, u32x4 (&mine)[WPT], u32x4 (&other)[WPT]) {
    constexpr int P = decltype(parity)::value;
    if (has2) wload(mine, c2, t2);
    if (t == 0 && c + 1 < nchunks) halo_issue(c + 1);
    tap_mfma(wl + (size_t)P * BN * PITCHW, t);
    if (has1) wstore(other, P ^ 1);
    __syncthreads();
    if (t + 1 == a.ntaps && c + 1 < nchunks) {
      halo_commit();
      __syncthreads();
    }
  };

  if constexpr (DMA) {
    // Filter slices by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass), double buffered by
    // stage (= G taps of one chunk): the DMA of stage s+1 runs under the MFMAs of stage s, ONE barrier per stage.
    // A wave instruction writes 1 KB = 64 / UNITS filter rows linearly; the swizzle goes on the SOURCE address.
    static_assert(UNITS == 4 || UNITS == 8, "DMA filter rows are 64 or 128 bytes");
    constexpr bool EXACT = (G == 3);
    constexpr int ROWS = 64 / UNITS;                       // filter rows per wave instruction
    constexpr int NBLK = BN / ROWS;                        // wave instructions per slice
    constexpr int NWV = WM * WN;
    const int rib = lane / UNITS, up = lane % UNITS;       // this lane's row inside a block / LDS unit
    const int spc = (a.ntaps + G - 1) / G;
    const int nstages = nchunks * spc;
    // source = wave-uniform slice base (SGPRs) + this lane's byte offset (loop invariant); destination = LDS byte address
    // (one address-space cast here, integer offsets in the loop): a piece costs a few scalar instructions
    constexpr int NPIECE = (NBLK + NWV - 1) / NWV;
    const fsr_lds_addr_t wl_addr = FSR_LDS_ADDR(wl);
    unsigned voff[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int row = (wave + i * NWV) * ROWS + rib;
      voff[i] = (unsigned)(row * a.Cin + ((up ^ swz(row & 15)) * EPB)) * (unsigned)sizeof(T);
    }
    auto dma_stage = [&](int c, int g, int buf) {
      static_for<0, G>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int t = g * G + j;
        if (EXACT || t < a.ntaps) {
          const T* src = wnb + ((tap_code(a, t) >> 4) * slice_stride + (unsigned)(c * KC));
          const fsr_lds_addr_t dst = wl_addr + (fsr_lds_addr_t)((buf * G + j) * BN * KC) * sizeof(T);
#pragma unroll
          for (int i = 0; i < NPIECE; ++i) {
            const int blk = wave + i * NWV;
            if (NBLK % NWV == 0 || blk < NBLK) FSR_GLDS16_SAT(src, voff[i], dst + (fsr_lds_addr_t)(blk * ROWS * KC) * sizeof(T));
          }
        }
      });
    };
    halo_issue(0);
    dma_stage(0, 0, 0);
    halo_commit();
    FSR_WAIT_DMA();
    __syncthreads();
    int c = 0, g = 0;
    for (int s = 0; s < nstages; ++s) {
      const int buf = s & 1;
      int cn = c, gn = g + 1;
      if (gn == spc) { gn = 0; ++cn; }
#if FSR_ABL != 1 && FSR_ABL != 3
      if (s + 1 < nstages) dma_stage(cn, gn, buf ^ 1);   // that buffer was last read in stage s-1, a barrier ago
#endif
#if FSR_ABL != 2 && FSR_ABL != 3
      if (g == 0 && c + 1 < nchunks) halo_issue(c + 1);
#endif
      if constexpr (EXACT) {
        stage_mfma3(wl + (size_t)(buf * G * BN) * KC, g);
      } else {
        static_for<0, G>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if (g * G + j < a.ntaps) tap_mfma(wl + (size_t)((buf * G + j) * BN) * KC, g * G + j);
        });
      }
      FSR_WAIT_DMA();                          // this wave's pieces of the next stage have landed (issued a stage ago)
      if (gn == 0 && cn < nchunks) {           // the next stage opens a chunk: replace the halo between two barriers
        __syncthreads();
        halo_commit();
      }
      __syncthreads();                         // every wave's pieces are visible
      c = cn;
      g = gn;
    }
  } else if constexpr (G == 1) {
    const int nsteps = nchunks * a.ntaps;
    halo_issue(0);
    halo_commit();
    wload(wregA, 0, 0);
    wstore(wregA, 0);
    if (nsteps > 1) wload(wregB, a.ntaps > 1 ? 0 : 1, a.ntaps > 1 ? 1 : 0);
    __syncthreads();

    {
      int c = 0, t = 0;            // current step
      int c2 = 0, t2 = 2;          // the step two ahead
      while (t2 >= a.ntaps) { t2 -= a.ntaps; ++c2; }
      for (int s = 0; s < nsteps; s += 2) {
        step_body(std::integral_constant<int, 0>{}, c, t, s + 1 < nsteps, s + 2 < nsteps, c2, t2, wregA, wregB);
        if (++t == a.ntaps) { t = 0; ++c; }
        if (++t2 == a.ntaps) { t2 = 0; ++c2; }
        if (s + 1 < nsteps) {
          step_body(std::integral_constant<int, 1>{}, c, t, s + 2 < nsteps, s + 3 < nsteps, c2, t2, wregB, wregA);
          if (++t == a.ntaps) { t = 0; ++c; }
          if (++t2 == a.ntaps) { t2 = 0; ++c2; }
        }
      }
    }

  } else {
    // Stage = up to G taps of one chunk.  The G filter slices of a stage sit in LDS together (single
    // buffered); the slices of the next stage are loaded into registers right after the barriers and written one stage
    // later: two barriers per G taps (G x MT x NT x KC/KSTEP MFMAs per wave) instead of one per tap, and the halo of
    // the next chunk is committed between the same two barriers.
    // G = 3 serves the 9-tap launches (host checked ntaps % 3 == 0): no per-tap guards, so the compiler schedules
    // across the taps of a stage.  G = 2 serves the 1/2/2/4-tap classes: the last stage may hold a single tap.
    constexpr bool EXACT = (G == 3);
    u32x4 wreg[G][WPT];
    const int spc = (a.ntaps + G - 1) / G;     // stages per chunk
    const int nstages = nchunks * spc;
    halo_issue(0);
    static_for<0, G>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (EXACT || j < a.ntaps) wload(wreg[j], 0, j);
    });
    int c = 0, g = 0;
    for (int s = 0; s < nstages; ++s) {
      if (s) __syncthreads();                  // every wave is done reading the previous stage's slices (and halo)
      static_for<0, G>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (EXACT || g * G + j < a.ntaps) wstore(wreg[j], j);
      });
      if (g == 0) halo_commit();
      __syncthreads();
      int cn = c, gn = g + 1;
      if (gn == spc) { gn = 0; ++cn; }
      if (s + 1 < nstages)
        static_for<0, G>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if (EXACT || gn * G + j < a.ntaps) wload(wreg[j], cn, gn * G + j);
        });
      if (g == 0 && c + 1 < nchunks) halo_issue(c + 1);
      static_for<0, G>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (EXACT || g * G + j < a.ntaps) tap_mfma(wl + (size_t)j * BN * PITCHW, g * G + j);
      });
      c = cn;
      g = gn;
    }
    if (a.stats) __syncthreads();              // the epilogue reuses the halo image for the statistics
  }

  // ---------------------------------------------------------------- epilogue
  float slope = (a.act == FSR_ACT_PRELU) ? a.prelu[0] : a.slope;
  if (a.act == FSR_ACT_NONE || a.act == FSR_ACT_TANH) slope = 1.f;
  if (a.act == FSR_ACT_RELU) slope = 0.f;
  // [WM][BN][2] statistics of this workgroup's pixel-row groups (reuses the halo image: the main loop ended on a barrier,
  // every wave is done with LDS).  Every slot is written exactly once, by the wave (wm, wn) that owns it.
  float* sred = (float*)smem;
  const int gx = gx0 + l15;
  const int gyb = gy0 + wm * MT;            // first grid row of this wave
  const int cob = nb * BN + wn * NT * 16 + lg * 4;  // first of this lane's channels (tile n adds 16 n)
  const bool col_ok = gx < a.GW;
  if (BN == 16 && ((a.out_f32 && sizeof(T) == 2) || a.Cout % 16 != 0 || a.out_f32 == FSR_OUT_U8)) {  // only the 16-wide configs carry this code
    // ---- thin / float path (memory-bound layers): per-element guards, optional scale, tanh
    static_for<0, NT>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      const int co = cob + n * 16;
      static_for<0, MT>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const int gy = gyb + m;
        if (col_ok && gy < a.GH) {
          const size_t off = (((size_t)img * a.FOH + gy * a.osy + a.ooy) * a.FOW + gx * a.osx + a.oox) * a.Cout + co;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (co + r < a.Cout) {
              float v = acc[m][n][r];
              if (a.oscale) v *= a.oscale[co + r];
              if (a.bias) v += a.bias[co + r];
              v = (a.act == FSR_ACT_TANH) ? tanhf(v) : (v > 0.f ? v : v * slope);
              if (a.out_f32 == FSR_OUT_U8) ((unsigned char*)a.out)[off + r] = image_u8(v);
              else if (a.out_f32 || sizeof(T) == 4) ((float*)a.out)[off + r] = v;
              else ElemIO<ST>::st((ST*)a.out + (off + r), v);
            }
        }
      });
    });
    return;
  }
  ST* outp = (ST*)a.out;
  ST* prep = (ST*)a.preact;
  const ST* maskp = (const ST*)a.dmask;
  const bool want_stats = a.stats != nullptr;
  // Masks (fused activation backward) are loaded for ALL tiles of the wave before the first store: a load issued after
  // a store may alias it as far as the compiler knows, and gfx9 counts loads and stores on one counter, so every
  // (tile row) mask load would otherwise wait for the previous row's store to reach memory -- MT x NT serialised
  // round trips per workgroup.
  constexpr bool PREMASK = sizeof(T) == 2 && !X3;
  u32x2 mkreg[PREMASK ? NT : 1][PREMASK ? MT : 1];
  const bool premask = PREMASK && maskp && !a.ps && a.premask;
  if (premask) {
    const unsigned base0 = (unsigned)((img * a.FOH + gyb * a.osy + a.ooy) * a.FOW + gx * a.osx + a.oox) * (unsigned)a.Cout + (unsigned)cob;
    const unsigned rs = (unsigned)(a.osy * a.FOW * a.Cout);
    static_for<0, NT>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      static_for<0, MT>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        u32x2 t = (u32x2){0u, 0u};
        if constexpr (PREMASK) {
          if (col_ok && gyb + m < a.GH) t = *(const u32x2*)(maskp + (base0 + n * 16 + m * rs));
          mkreg[n][m] = t;
        }
      });
    });
  }
  static_for<0, NT>([&](auto nc) {
    constexpr int n = decltype(nc)::value;
    const int co = cob + n * 16;
    f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned base, rstride;  // 32-bit element offsets (tensors hold < 2^31 elements): grid row gyb, and the row stride
    if (!a.ps) {
      if (a.bias) bv = *(const f32x4*)(a.bias + co);
      base = (unsigned)((img * a.FOH + gyb * a.osy + a.ooy) * a.FOW + gx * a.osx + a.oox) * (unsigned)a.Cout + (unsigned)co;
      rstride = (unsigned)(a.osy * a.FOW * a.Cout);
    } else {
      // filter rows were packed [quadrant q][channel cc]; the bias stays in torch order 4*cc + q
      const int cps = a.Cout >> 2;
      const int q = co / cps, cc = co - q * cps;
      if (a.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = a.bias[4 * (cc + r) + q];
      }
      base = (unsigned)((img * 2 * a.FOH + 2 * gyb + (q >> 1)) * (2 * a.FOW) + 2 * gx + (q & 1)) * (unsigned)cps + (unsigned)cc;
      rstride = (unsigned)(4 * a.FOW * cps);
    }
    f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f}, s2 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(T) == 2 && MT % 2 == 0) {
      if (a.pool2) {
        // MaxPool2d(2,2) fused: rows (m, m+1) of this wave (gyb is even) and columns (l15, l15 ^ 1) of neighbouring lanes;
        // even lanes store pixel (gy / 2, gx / 2) of the [N][FOH/2][FOW/2][Cout] tensor.  The activation is monotonic, so
        // it commutes with the maximum.
        static_for<0, MT / 2>([&](auto mc) {
          constexpr int m = 2 * decltype(mc)::value;
          f32x4 v = acc[m][n] + bv, w = acc[m + 1][n] + bv;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = fmaxf(v[r], w[r]);
            x = fmaxf(x, __shfl_xor(x, 1, 64));
            v[r] = fmaxf(x, 0.f) + slope * fminf(x, 0.f);
          }
          if (col_ok && gyb + m < a.GH && !(l15 & 1)) {
            const unsigned off = (unsigned)((img * (a.FOH >> 1) + ((gyb + m) >> 1)) * (a.FOW >> 1) + (gx >> 1)) * (unsigned)a.Cout + (unsigned)co;
            store_vec4<ST>(outp + off, v);
          }
        });
        return;
      }
    }
    static_for<0, MT>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      if (col_ok && gyb + m < a.GH) {
        f32x4 v = acc[m][n] + bv;
        if (maskp) {   // data-gradient launch fused with the producer's activation backward: dz = dx * act'(y)
          float mk[4];
          if constexpr (X3) {
            const f32x4 t = x3_ld4(maskp + (base + m * rstride));
            mk[0] = t[0]; mk[1] = t[1]; mk[2] = t[2]; mk[3] = t[3];
          } else if constexpr (sizeof(T) == 4) {
            const f32x4 t = *(const f32x4*)(maskp + (base + m * rstride));
            mk[0] = t[0]; mk[1] = t[1]; mk[2] = t[2]; mk[3] = t[3];
          } else {
            u32x2 t;
            if (premask) t = mkreg[PREMASK ? n : 0][PREMASK ? m : 0];
            else t = *(const u32x2*)(maskp + (base + m * rstride));
            mk[0] = cvt_lo<T>(t.x); mk[1] = cvt_hi<T>(t.x);
            mk[2] = cvt_lo<T>(t.y); mk[3] = cvt_hi<T>(t.y);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = a.dmask_add ? v[r] + mk[r] : (mk[r] > 0.f ? v[r] : v[r] * a.dmask_slope);
        }
        if (want_stats) {
          s1 += v;
          s2 += v * v;
        }
        if (prep) store_vec4<ST>(prep + (base + m * rstride), v);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = act_slope(v[r], slope);       // (NaN-propagating: fsr_common.h)
        store_vec4<ST>(outp + (base + m * rstride), v);
      }
    });
    if (want_stats) {
      // per-(image, channel) partial sums: xor-reduce the 16 pixel lanes of each lane group (a fixed butterfly), park the
      // wave's sums in its own LDS slot; the WM row groups are added in order below and the workgroup's vector goes to
      // its slot of the partial buffer (no atomics: the sums are bit-reproducible, reduce.hip adds the slots in order).
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x1 = s1[r], x2 = s2[r];
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) {
          x1 += __shfl_xor(x1, o, 64);
          x2 += __shfl_xor(x2, o, 64);
        }
        if (l15 == 0) {
          const int cl = (wn * NT + n) * 16 + lg * 4 + r;
          sred[(wm * BN + cl) * 2] = x1;
          sred[(wm * BN + cl) * 2 + 1] = x2;
        }
      }
    }
  });
  if (a.stats) {
    __syncthreads();
    const int slot = ty * a.tiles_x + tx;
    for (int i = tid; i < 2 * BN; i += NTHR) {
      float s = sred[i];
#pragma unroll
      for (int w = 1; w < WM; ++w) s += sred[w * 2 * BN + i];
      const int co = nb * BN + (i >> 1);
      if (co < a.Cout) a.stats[(((size_t)img * a.stats_P + slot) * a.Cout + co) * 2 + (i & 1)] = s;
    }
  }
}

// ---------------------------------------------------------------------------- host side
static int stage_mode();

static int pack_taps(ConvKArgs& a) {
  a.taps_lo = 0;
  a.taps_hi = 0;
  for (int t = 0; t < a.ntaps; ++t) {
    if (a.tdy[t] > 2 || a.tdx[t] > 2) return fsr_fail(-2, "conv3x3: tap offsets exceed the 3x3 footprint");
    const unsigned code = (unsigned)a.tdy[t] | ((unsigned)a.tdx[t] << 2) | ((unsigned)a.tw[t] << 4);
    if (t < 8) a.taps_lo |= (unsigned long long)code << (8 * t);
    else a.taps_hi = code;
  }
  return 0;
}

// `more` (optional): further classes of the same launch (see ConvKClasses); they share everything with `a` except
// the output grid, the tap table and the output offset.
template <typename T, int TH, int BN, int WM, int WN, int KC, int S, int G = 1, int DMA = 0, int X3 = 0>
static int launch_cfg(ConvKArgs& a, hipStream_t stream, ConvKArgs* more = nullptr, int nmore = 0) {
  constexpr int EPB = 16 / (int)sizeof(T);
  constexpr int PITCHW = KC + 2 * EPB, PITCHX = KC + (S == 2 ? 1 : 2) * EPB;
  if (a.Cin % KC != 0) return fsr_fail(-2, "conv3x3: Cin=%d is not a multiple of the chunk %d", a.Cin, KC);
  if (a.in_ps && (a.Cin / 4) % KC != 0)
    return fsr_fail(-2, "conv3x3: pixel-shuffled input needs (Cin/4)=%d to be a multiple of the chunk %d", a.Cin / 4, KC);
  if (a.CoutPad % BN != 0) return fsr_fail(-2, "conv3x3: padded Cout=%d is not a multiple of %d", a.CoutPad, BN);
  a.nblk_n = a.CoutPad / BN;
  a.premask = (stage_mode() & 65536) ? 0 : 1;
  a.HH = (TH - 1) * S + 3;
  a.HW = 15 * S + 3;
  // (x3: a.Cin counts physical bf16 channels, 2 x logical; the kernel's input offsets are unsigned element offsets, good to 2^32 --
  // a 64-channel 720p batch of 32 is 3.8e9 of them)
  if ((long long)a.N * a.IH * a.IW * a.Cin >= (X3 ? (1LL << 32) : (1LL << 31)) || (long long)a.N * a.FOH * a.FOW * a.Cout >= (1LL << 31))
    return fsr_fail(-2, "conv3x3: tensors with 2^31 or more elements are not supported");
  if (G == 3 && (a.ntaps % 3 != 0 || nmore > 0)) return fsr_fail(-2, "conv3x3: three-tap stages need a multiple of 3 taps");
  if (a.stats && nmore > 0) return fsr_fail(-2, "conv3x3: statistics are not available for multi-class launches");
  ConvKClasses cls = {};
  long long nwg = 0;
  for (int k = 0; k <= nmore; ++k) {
    ConvKArgs& b = k ? more[k - 1] : a;
    if (int rc = pack_taps(b)) return rc;
    if (k && b.ntaps > 8) return fsr_fail(-2, "conv3x3: a class of a multi-class launch has more than 8 taps");
    b.tiles_x = (b.GW + 15) / 16;
    b.tiles_y = (b.GH + TH - 1) / TH;
    nwg += (long long)b.tiles_x * b.tiles_y * a.N * a.nblk_n;
    if (nwg <= 0 || nwg > 0x7fffffffLL) return fsr_fail(-2, "conv3x3: bad grid");
    ConvKClass& c = cls.c[k];
    c.GH = b.GH; c.GW = b.GW; c.ntaps = b.ntaps; c.ooy = b.ooy; c.oox = b.oox;
    c.tiles_x = b.tiles_x; c.tiles_y = b.tiles_y; c.taps_lo = b.taps_lo;
    c.wg_end = (int)nwg;
  }
  cls.n = nmore + 1;
  a.stats_P = a.tiles_x * a.tiles_y;   // one partial slot per tile of an image (one-tile workgroups)
  if (a.stats && a.stats_P > a.stats_P_max) return fsr_fail(-3, "conv3x3: %d partial slots per image exceed the scratch buffer's %d", a.stats_P, a.stats_P_max);
  a.stats_tpi = a.stats_per = 0;
  const size_t lds = ((size_t)a.HH * a.HW * PITCHX + (DMA ? 2 * G * (size_t)BN * KC : (G == 1 ? 2 : G) * (size_t)BN * PITCHW)) * sizeof(T);
  auto kern = conv_igemm_kernel<T, TH, BN, WM, WN, KC, S, G, DMA, X3>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(WM * WN * 64), lds, stream, a, cls);
  fsr_note_kernel("conv_igemm_kernel<%s,%d,%d,%d,%d,%d,%d,%d,%d>", X3 ? "x3" : (sizeof(T) == 4 ? "f32" : (std::is_same<T, f16_t>::value ? "f16" : "bf16")), TH, BN, WM, WN, KC, S, G, DMA);
  return fsr_check_launch("conv_igemm_kernel");
}

// Tuning / test switch FSR_CONV_STAGE (bit mask, read per launch; default 1374 = every variant that measured faster):
//   2    three-tap stages for stride-2 forward launches            4  ... for the 8x128 configurations
//   8    ... for the 64-wide configurations                       16  stride-2 data gradient as ONE four-class launch
//   32   force 16-row tiles (tests: lets small shapes reach the tall configurations)
//   64   tall configuration: filter slices by LDS-DMA, double-buffered three-tap stages
//   256  8x128 configuration with 64-channel chunks: LDS-DMA filter slices (one tap per stage)
//   1024 stride-2 forward, 64 output channels: LDS-DMA three-tap stages
//   65536 DISABLE the up-front mask loads of the epilogue
static int stage_mode() {
  const char* e = getenv("FSR_CONV_STAGE");
  return e ? atoi(e) : 1374;
}
int fsr_conv_stage_mode() { return stage_mode(); }

#define FSR_GO(...) return launch_cfg<T, __VA_ARGS__>(a, stream, more, nmore)

template <typename T, int KCW, int KCN>
static int dispatch_T(ConvKArgs& a, int S, hipStream_t stream, ConvKArgs* more = nullptr, int nmore = 0) {
  // KCW: wide input-channel chunk (bf16: 64 channels = 128 B per pixel) used whenever Cin allows it at
  // stride 1 (half the barriers and halo passes per FLOP); KCN: narrow chunk (stride 2 halos are 4x larger)
  const int w16 = ((a.GH + 15) / 16) * 16 - a.GH, w8 = ((a.GH + 7) / 8) * 8 - a.GH;
  // 8-row tiles when 16-row tiles would waste half a tile, or when there are too few of them to give every CU
  // two workgroups (batch-1 inference: a 180x320 frame is only 240 tiles of 16x16 pixels)
  const long long tiles16 = (long long)a.N * ((a.GH + 15) / 16) * ((a.GW + 15) / 16) * (a.CoutPad / 64 > 0 ? a.CoutPad / 64 : 1) * (nmore + 1);
  const int sm = stage_mode();
  const bool th8 = !(sm & 32) && ((w16 - w8 >= 8) || tiles16 < 1024);
  const bool wide = (a.Cin % KCW == 0) && (!a.in_ps || (a.Cin / 4) % KCW == 0);
  // multi-tap stages (template parameter G): 3 for the 9-tap launches, 2 for the 1/2/2/4-tap classes of a stride-2
  // data gradient
  const bool t9 = a.ntaps == 9, t2 = nmore > 0 || a.ntaps == 2 || a.ntaps == 4;
  if (nmore > 0 && S != 1) return fsr_fail(-2, "conv3x3: multi-class launches are stride-1 launches");
  if (a.CoutPad % 128 == 0) {
    if (S == 2) {
      if (t9 && (sm & 2)) FSR_GO(8, 128, 2, 2, KCN, 2, 3);
      FSR_GO(8, 128, 2, 2, KCN, 2);
    }
    // (8-wave variants <16,128,4,2> / <32,64,8,1> halve the filter traffic per FLOP but measured 5-10 % slower:
    // one workgroup per CU means every wave waits at the same barriers; two independent 4-wave workgroups overlap)
    // tall tile (256 px x 128 co per workgroup, wave = 64 px x 128 co): 12 fragment reads per 32 MFMAs instead of 16 and
    // half the workgroups; measured +2..7 % on the 128/256-channel layers.  Narrow chunks keep two workgroups per CU.
    if (!th8 && a.Cin >= 128) {
      // (32-row tiles with 8 waves -- half the filter traffic per FLOP, one workgroup per CU -- measured again with the
      // pipelined stage: 128-channel layers +-0, 256-channel layers -5 %: not dispatched.  12-row tiles, which turn the 2.25 /
      // 4.5 workgroup rounds of the 48^2 / 96^2 layers into whole rounds: +-1 % at 48^2, -4 % at 96^2 -- the workgroups of a
      // partial last round run alone on their CUs and finish early, the rounds are not the quantum the arithmetic suggests)
      if (t9 && (sm & 64)) FSR_GO(16, 128, 4, 1, KCN, 1, 3, 1);
      if (t2 && (sm & 64)) FSR_GO(16, 128, 4, 1, KCN, 1, 2, 1);
      FSR_GO(16, 128, 4, 1, KCN, 1);
    }
    if (t9 && wide && (sm & 256)) FSR_GO(8, 128, 2, 2, KCW, 1, 1, 1);
    if (t9 && (sm & 4)) FSR_GO(8, 128, 2, 2, KCN, 1, 3);
    if (t2 && wide && (sm & 4)) FSR_GO(8, 128, 2, 2, KCW, 1, 2);
    if (wide) FSR_GO(8, 128, 2, 2, KCW, 1);
    FSR_GO(8, 128, 2, 2, KCN, 1);
  }
  if (a.CoutPad % 64 == 0) {
    if (S == 2) {
      if (t9 && (sm & 1024)) FSR_GO(8, 64, 2, 2, KCN, 2, 3, 1);
      if (t9 && (sm & 2)) FSR_GO(8, 64, 2, 2, KCN, 2, 3);
      FSR_GO(8, 64, 2, 2, KCN, 2);
    }
    if (th8) {
      if (t2 && wide && (sm & 8)) FSR_GO(8, 64, 2, 2, KCW, 1, 2);
      if (wide) FSR_GO(8, 64, 2, 2, KCW, 1);
      FSR_GO(8, 64, 2, 2, KCN, 1);
    }
    if (t9 && (sm & 8)) FSR_GO(16, 64, 4, 1, KCN, 1, 3);
    if (t2 && wide && (sm & 8)) FSR_GO(16, 64, 4, 1, KCW, 1, 2);
    if (wide) FSR_GO(16, 64, 4, 1, KCW, 1);
    FSR_GO(16, 64, 4, 1, KCN, 1);
  }
  if (a.CoutPad % 16 == 0) {  // thin outputs (head conv, image gradients) and small test networks
    if (S == 2) FSR_GO(8, 16, 4, 1, KCN, 2);
    // bandwidth/latency-bound: small 8x16 tiles keep 4-5 workgroups per CU in flight
    if (wide) FSR_GO(8, 16, 4, 1, KCW, 1);
    FSR_GO(8, 16, 4, 1, KCN, 1);
  }
  return fsr_fail(-2, "conv3x3: unsupported padded Cout=%d (need a multiple of 16)", a.CoutPad);
}

// FSR_X3 on this kernel: a.Cin counts PHYSICAL bf16 channels (2 x logical, set by the caller), 64-channel (hi | lo) chunks,
// one tap per step (G = 1, register-staged filter slices).
#define FSR_GO3(...) return launch_cfg<bf16_t, __VA_ARGS__, 1, 0, 1>(a, stream, more, nmore)
static int dispatch_X3(ConvKArgs& a, int S, hipStream_t stream, ConvKArgs* more = nullptr, int nmore = 0) {
  const int w16 = ((a.GH + 15) / 16) * 16 - a.GH, w8 = ((a.GH + 7) / 8) * 8 - a.GH;
  const long long tiles16 = (long long)a.N * ((a.GH + 15) / 16) * ((a.GW + 15) / 16) * (a.CoutPad / 64 > 0 ? a.CoutPad / 64 : 1) * (nmore + 1);
  const bool th8 = !(stage_mode() & 32) && ((w16 - w8 >= 8) || tiles16 < 1024);
  if (nmore > 0 && S != 1) return fsr_fail(-2, "conv3x3: multi-class launches are stride-1 launches");
  if (a.CoutPad % 128 == 0) {
    if (S == 2) FSR_GO3(8, 128, 2, 2, 64, 2);
    FSR_GO3(8, 128, 2, 2, 64, 1);
  }
  if (a.CoutPad % 64 == 0) {
    if (S == 2) FSR_GO3(8, 64, 2, 2, 64, 2);
    if (th8) FSR_GO3(8, 64, 2, 2, 64, 1);
    FSR_GO3(16, 64, 4, 1, 64, 1);
  }
  if (a.CoutPad % 16 == 0) {
    if (S == 2) FSR_GO3(8, 16, 4, 1, 64, 2);
    FSR_GO3(8, 16, 4, 1, 64, 1);
  }
  return fsr_fail(-2, "conv3x3: unsupported padded Cout=%d (need a multiple of 16)", a.CoutPad);
}
#undef FSR_GO3

int fsr_conv_igemm_dispatch_classes(int dtype, ConvKArgs* cls, int n, hipStream_t stream) {
  if (n < 1 || n > 4) return fsr_fail(-2, "conv3x3: %d classes in one launch", n);
  if (cls[0].query) return 0;         // (fsr_conv3x3_pack_block: these launches read the standard pack)
  if (cls[0].wlin) return fsr_fail(-2, "conv3x3: a stage-contiguous filter pack reached a kernel that reads the standard one");
  if (dtype == FSR_BF16) return dispatch_T<bf16_t, 64, 32>(cls[0], 1, stream, cls + 1, n - 1);
  if (dtype == FSR_F16) return dispatch_T<f16_t, 64, 32>(cls[0], 1, stream, cls + 1, n - 1);
  if (dtype == FSR_F32) return dispatch_T<float, 16, 16>(cls[0], 1, stream, cls + 1, n - 1);
  if (dtype == FSR_X3) return dispatch_X3(cls[0], 1, stream, cls + 1, n - 1);
  return fsr_fail(-2, "conv3x3: unknown dtype %d", dtype);
}

#ifdef FSR_EXPERIMENT_C64V3
int fsr_conv64_v3_try(int dtype, ConvKArgs& a, int S, hipStream_t stream);      // tools/experiments/conv64_v3.hip
#endif

int fsr_conv_igemm_dispatch(int dtype, ConvKArgs& a, int S, hipStream_t stream) {
  if (dtype == FSR_X3) {
    // 64 and more logical input channels, stride 1: the all-DMA 32x32x16 kernel on three virtual chunks per channel group (conv_tall3.hip)
    if (const int rc = fsr_conv_tall3_try(dtype, a, S, stream)) return rc < 0 ? rc : 0;
    if (a.query) return 0;            // (the standard x3 pack)
    if (a.wlin) return fsr_fail(-2, "conv3x3: a stage-contiguous filter pack reached a kernel that reads the standard one");
    // the 64 -> 3 ends (head, image gradients): the streaming thin kernel's x3 form (conv64_persistent.hip)
    if (const int rc = fsr_conv64_persistent_try(dtype, a, S, stream)) return rc < 0 ? rc : 0;
    return dispatch_X3(a, S, stream);
  }
  if (a.Cin == 64) {
    if (a.query) return 0;            // (the 64-input-channel kernels read the standard pack)
    if (a.wlin) return fsr_fail(-2, "conv3x3: a stage-contiguous filter pack reached a kernel that reads the standard one");
  }
  // (Round 6 measured two alternatives for the 64-input-channel stride-1 layers and kept conv64_v2: the all-DMA kernel on two
  // 32-channel chunks per tile -- VGG 64->64 457 us against 412 -- and a resident-filter kernel on 32x32x16 MFMAs with 16 x 32-pixel
  // tiles, tools/experiments/conv64_v3.hip -- 406 against 412, 64->128 236 against 214; profiles/r06_conv64_v2_v3_t3.txt,
  // profiles/r06_conv64_v3_ablation.txt.)
#ifdef FSR_EXPERIMENT_C64V3   // tools/experiments/build_v3_variants.sh only: never defined for the product library
  if (const int rc = fsr_conv64_v3_try(dtype, a, S, stream)) return rc < 0 ? rc : 0;
#endif
  // 64 -> 64 channel stride-1 layers: persistent kernel with the whole filter resident in LDS (conv64_persistent.hip)
  if (const int rc = fsr_conv64_persistent_try(dtype, a, S, stream)) return rc < 0 ? rc : 0;
  // 64 -> 64 channel stride-2 forward (Discriminator block 0): persistent streaming kernel (conv64_persistent.hip)
  if (const int rc = fsr_conv64_s2fwd_try(dtype, a, S, stream)) return rc < 0 ? rc : 0;
  // 128..512-channel layers, stride 1 and the stride-2 forward: 32x32x16 MFMA, both operands by LDS-DMA (conv_tall3.hip)
  if (const int rc = fsr_conv_tall3_try(dtype, a, S, stream)) return rc < 0 ? rc : 0;
  if (a.query) return 0;              // (fsr_conv3x3_pack_block: conv_igemm_kernel reads the standard pack)
  if (a.wlin) return fsr_fail(-2, "conv3x3: a stage-contiguous filter pack reached a kernel that reads the standard one");
  if (dtype == FSR_BF16) return dispatch_T<bf16_t, 64, 32>(a, S, stream);
  if (dtype == FSR_F16) return dispatch_T<f16_t, 64, 32>(a, S, stream);
  if (dtype == FSR_F32) return dispatch_T<float, 16, 16>(a, S, stream);
  return fsr_fail(-2, "conv3x3: unknown dtype %d", dtype);
}
