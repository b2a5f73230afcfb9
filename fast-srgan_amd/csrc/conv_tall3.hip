// 3x3 stride-1 convolution for the 128..512-channel layers: persistent implicit GEMM on v_mfma_f32_32x32x16_{bf16,f16}
// with BOTH operands arriving by LDS-DMA.  (Round 3: replaces the 16x16x32 "tall" configuration of conv_igemm.hip on the
// layers that dominate the GAN iteration.)
//
// Replaces the torch.nn.Conv2d(k=3, p=1, stride 1) calls of the reference with 128 or more input channels and a
// multiple of 128 output channels:
//   torchvision vgg19.features convs 7..32 behind /root/reference/model.py:8 (conv2_2, conv3_x, conv4_x: forward twice per
//   iteration, trainer.py:190-191, and their data gradients once, trainer.py:195),
//   /root/reference/model.py:160-177 (Discriminator 128->256 and 256->512 stride-1 blocks: data gradients; their forwards
//   carry InstanceNorm statistics and stay on conv_igemm.hip).
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

constexpr int T3_P = 18;                           // halo rows and columns (16 x 16 outputs + the 3 x 3 footprint)
constexpr int T3_HUNITS = T3_P * T3_P * 4;         // 16-byte units of one halo chunk
constexpr int T3_NHP = (T3_HUNITS + 63) / 64;      // DMA pieces per halo chunk (21)
constexpr int T3_HALO_BYTES = T3_NHP * 1024;
constexpr int T3_PIXB = 64, T3_ROWB = T3_P * T3_PIXB;   // bytes per halo pixel / halo row

template <int BN, int G, int NSLOT> constexpr int t3_lds_bytes() { return 2 * T3_HALO_BYTES + NSLOT * G * BN * 64; }

__device__ __forceinline__ int t3_swz_row(int R) { return (R >> 2) & 3; }
__device__ __forceinline__ int t3_swz_col(int x) { return (x >> 1) & 3; }

template <typename V>
__device__ __forceinline__ V t3_lds_read(const char* smem, unsigned off) {
  return *FSR_LDS_PTR(const V, smem + off);
}

template <typename T, int BN, int NW, int G, int NSLOT>
__global__ __launch_bounds__(NW * 64, 2) void conv_tall3_kernel(const ConvKArgs a) {
  static_assert(BN == NW * 32, "a wave owns 128 pixels x 64 channels");
  static_assert(9 % G == 0 && (9 / G) % NSLOT == 1, "stage s lives in slot s % NSLOT == (chunk + stage in chunk) % NSLOT");
  constexpr int WCO = NW / 2;
  constexpr int SPC = 9 / G;                       // stages per chunk
  constexpr int NQ = 2 * G;                        // substeps per stage
  constexpr int D = NSLOT - 1;                     // the DMA runs D stages ahead
  constexpr int FP = BN / 16 / NW;                 // filter pieces per tap and wave (2)
  constexpr int HPW = (T3_NHP + NW - 1) / NW;      // halo pieces per chunk and wave
  constexpr int SLOT_BYTES = G * BN * 64;
  static_assert(HPW + D <= SPC + 1, "halo pieces must land before their chunk starts");
  static_assert(FP == 2, "piece placement below assumes two filter pieces per tap and wave");

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
    const int R = wco * 64 + l31;
#pragma unroll
    for (int j = 0; j < 2; ++j) aoff[j] = (unsigned)(2 * T3_HALO_BYTES + R * 64 + (((2 * j + hi) ^ t3_swz_row(R)) << 4));
  }
  // pixel fragment m of tap (ky, kx), k half j, halo buffer hb: hb*HALO + (2m + ky)*ROWB + boff[kx][j]
  unsigned boff[3][2];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      boff[kx][j] = (unsigned)(((wpx * 8 + lrow) * T3_P + l15 + kx) * T3_PIXB + (((2 * j + hi) ^ t3_swz_col(l15 + kx)) << 4));
  // DMA source of filter piece k (rows 16*(wave + k*NW) .. +15 of a slice block): byte offset of this lane's 16 bytes
  unsigned wvoff[FP];
#pragma unroll
  for (int k = 0; k < FP; ++k) {
    const int R = (wave + k * NW) * 16 + (lane >> 2), ul = lane & 3, i = R & 31;
    const int co = (R & ~31) + 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3);
    wvoff[k] = (unsigned)((co * a.Cin + ((ul ^ t3_swz_row(R)) << 3)) * (int)sizeof(T));
  }

  f32x16 acc[2][4];
  s16x8 fa[2][2], fb[2][4];

  // ---- per-tile state ----------------------------------------------------------------------------------------------
  int img = 0, gy0 = 0, gx0 = 0, nb = 0;
  unsigned hvoff[HPW];
  unsigned wsoff = 0;             // byte offset of this tile's channel block inside a filter slice
  auto setup = [&](int tile) {
    int L = tile;
    nb = L % a.nblk_n; L /= a.nblk_n;
    const int tx = L % a.tiles_x; L /= a.tiles_x;
    const int ty = L % a.tiles_y;
    img = L / a.tiles_y;
    gy0 = ty * 16;
    gx0 = tx * 16;
    wsoff = (unsigned)(nb * BN * a.Cin * (int)sizeof(T));
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
      const int U = (wave + k * NW) * 64 + lane;
      const int hp = U >> 2, ul = U & 3;
      const int hy = hp / T3_P, hx = hp - hy * T3_P;
      const int iy = gy0 - 1 + hy, ix = gx0 - 1 + hx;
      unsigned o = ~0u;                                          // beyond the buffer: the DMA writes zeros
      if (U < T3_HUNITS && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW)
        o = (unsigned)((((img * a.IH + iy) * a.IW + ix) * a.Cin + ((ul ^ t3_swz_col(hx)) << 3)) * (int)sizeof(T));
      hvoff[k] = o;
    }
  };
  // DMA pieces: halo piece k of chunk c -> buffer c & 1; filter pieces of tap t (canonical ky*3+kx) of chunk c -> ring slot
  auto dma_halo = [&](int c, int k) {
    if (wave + k * NW < T3_NHP && !(a.t3_dbg & 2))
      FSR_BLDS16(in_buf, hvoff[k], (unsigned)(c * 64), halo_addr + (fsr_lds_addr_t)((c & 1) * T3_HALO_BYTES + (wave + k * NW) * 1024));
  };
  auto dma_filter = [&](int c, int t, int k, unsigned slot_tap_off) {
    if (!(a.t3_dbg & 2)) FSR_BLDS16(w_buf, wvoff[k], a.t3_woff[t] + wsoff + (unsigned)(c * 64),
               ring_addr + (fsr_lds_addr_t)(slot_tap_off + (wave + k * NW) * 1024));
  };
  auto slot_of = [&](int c, int si) { return (unsigned)(((c + si) & (NSLOT - 1)) * SLOT_BYTES); };

  // reads of substep (tap t = ky*3+kx in ring position g of its stage, k half j): fragment r in the MFMA's need order
  // a0 b0 b1 b2 b3 a1
  auto read_frag = [&](auto rc, auto bufc, auto tc, auto gc, auto jc, unsigned sl, unsigned hb) {
    constexpr int r = decltype(rc)::value, buf = decltype(bufc)::value, t = decltype(tc)::value, g = decltype(gc)::value,
                  j = decltype(jc)::value;
    constexpr int ky = t / 3, kx = t % 3;
    if constexpr (r == 0 || r == 5) {
      constexpr int n = r == 0 ? 0 : 1;
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

  if (tile < a.t3_ntiles) {
    setup(logical(tile));
    // prologue of the first tile: halo chunk 0 and the first D stages
#pragma unroll
    for (int k = 0; k < HPW; ++k) dma_halo(0, k);
    static_for<0, D>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      static_for<0, G>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
#pragma unroll
        for (int k = 0; k < FP; ++k) dma_filter(s / SPC, (s % SPC) * G + g, k, slot_of(s / SPC, s % SPC) + g * BN * 64);
      });
    });
  }

  while (tile < a.t3_ntiles) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][m][e] = 0.f;
    FSR_WAIT_VM(0);                 // the prologue pieces of this wave (and the previous tile's stores)
    FSR_BARRIER();
    // fragments of substep 0
    static_for<0, 6>([&](auto rc) {
      read_frag(rc, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{},
                std::integral_constant<int, 0>{}, slot_of(0, 0), 0u);
    });

    for (int c = 0; c < ((a.t3_dbg & 4) ? 0 : nchunks); ++c) {
      static_for<0, SPC>([&](auto sic) {
        constexpr int si = decltype(sic)::value;
        const unsigned sl = slot_of(c, si);
        const unsigned hb = (unsigned)((c & 1) * T3_HALO_BYTES);
        // the stage whose pieces this stage issues
        constexpr int siD = (si + D) % SPC;
        const int cD = c + (si + D) / SPC;
        const bool issue = cD < nchunks;
        const unsigned slD = slot_of(cD, siD);
        // the stage after this one (its first fragments are read in this stage's last substep)
        constexpr int siN = (si + 1) % SPC;
        const int cN = c + (si + 1) / SPC;
        const bool has_next = cN < nchunks;
        const unsigned slN = slot_of(cN, siN);
        const unsigned hbN = (unsigned)((cN & 1) * T3_HALO_BYTES);

        static_for<0, NQ>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          constexpr int g = q / 2, j = q % 2, t = si * G + g;
          constexpr int buf = q & 1;
          if constexpr (q == NQ - 1) {
            // publish the next stage: this wave's pieces of stage s+1 have landed when at most the pieces of the D-1 younger
            // stages are outstanding (halo pieces among them only make the wait longer, never shorter)
            if constexpr (D == 1) {
              FSR_WAIT_VM(0);
            } else {
              const int s = c * SPC + si, nst = nchunks * SPC;
              const int lo = s - D + 2 > 0 ? s - D + 2 : 0;
              const int young = (s < nst - 1 - D ? s : nst - 1 - D) - lo + 1;   // stages in [s-D+2, s] that issued pieces
              if (young <= 0) FSR_WAIT_VM(0);
              else if (young == 1) FSR_WAIT_VM(G * FP);
              else FSR_WAIT_VM(2 * G * FP);
              static_assert(D <= 3, "add cases for deeper rings");
            }
            FSR_WAIT_LGKM0();
            FSR_BARRIER();
          }
          static_for<0, 8>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int n = i / 4, m = i % 4;
            acc[n][m] = Mfma32<T>::run(fa[buf][n], fb[buf][m], acc[n][m]);
            if constexpr (i < 6) {
              if constexpr (q + 1 < NQ) {
                constexpr int q1 = q + 1;
                read_frag(ic, std::integral_constant<int, buf ^ 1>{}, std::integral_constant<int, si * G + q1 / 2>{},
                          std::integral_constant<int, q1 / 2>{}, std::integral_constant<int, q1 % 2>{}, sl, hb);
              } else {
                if (has_next)
                  read_frag(ic, std::integral_constant<int, buf ^ 1>{}, std::integral_constant<int, siN * G>{},
                            std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, slN, hbN);
              }
            }
            // DMA pieces of stage s+D: the halo piece first (the likeliest HBM miss), then two filter pieces per substep
            if constexpr (q == 0 && i == 1 && si < HPW) {
              if (c + 1 < nchunks) dma_halo(c + 1, si);
            }
            if constexpr (q < G && i >= 6) {
              if (issue) dma_filter(cD, siD * G + q, i - 6, slD + q * BN * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
          });
        });
      });
    }

    // ---- next tile's prologue goes out before this tile's stores ---------------------------------------------------
    const int oimg = img, ogy0 = gy0, ogx0 = gx0, onb = nb;
    const int next = tile + nround;
    FSR_BARRIER();                                  // every wave has left the last substep's LDS reads behind
    if (next < a.t3_ntiles) {
      setup(logical(next));
#pragma unroll
      for (int k = 0; k < HPW; ++k) dma_halo(0, k);
      static_for<0, D>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        static_for<0, G>([&](auto gc) {
          constexpr int g = decltype(gc)::value;
#pragma unroll
          for (int k = 0; k < FP; ++k) dma_filter(s / SPC, (s % SPC) * G + g, k, slot_of(s / SPC, s % SPC) + g * BN * 64);
        });
      });
    }

    // ---- epilogue -----------------------------------------------------------------------------------------------------
    float slope = a.slope;
    if (a.act == FSR_ACT_NONE) slope = 1.f;
    if (a.act == FSR_ACT_RELU) slope = 0.f;
    T* outp = (T*)a.out;
    const T* maskp = (const T*)a.dmask;
    const int gx = ogx0 + l15;
    static_for<0, 2>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      const int co = onb * BN + wco * 64 + n * 32 + hi * 16;
      float bv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) bv[e] = 0.f;
      if (a.bias) {
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const f32x4 b = *(const f32x4*)(a.bias + co + 4 * e4);
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[4 * e4 + e] = b[e];
        }
      }
      static_for<0, 4>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const int gy = ogy0 + wpx * 8 + 2 * m + lrow;
        const bool ok = gy < a.GH && gx < a.GW;
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = acc[n][m][e] + bv[e];
        if (a.pool2) {
          // MaxPool2d(2,2) fused: rows (gy, gy^1) sit in lanes (l, l^16), columns in (l, l^1); the activation is monotonic
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float x = v[e];
            x = fmaxf(x, __shfl_xor(x, 16, 64));
            x = fmaxf(x, __shfl_xor(x, 1, 64));
            v[e] = fmaxf(x, 0.f) + slope * fminf(x, 0.f);
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
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * e] = cvt_lo<T>(k0[e]) > 0.f ? v[2 * e] : v[2 * e] * a.dmask_slope;
              v[2 * e + 1] = cvt_hi<T>(k0[e]) > 0.f ? v[2 * e + 1] : v[2 * e + 1] * a.dmask_slope;
              v[8 + 2 * e] = cvt_lo<T>(k1[e]) > 0.f ? v[8 + 2 * e] : v[8 + 2 * e] * a.dmask_slope;
              v[8 + 2 * e + 1] = cvt_hi<T>(k1[e]) > 0.f ? v[8 + 2 * e + 1] : v[8 + 2 * e + 1] * a.dmask_slope;
            }
          }
          u32x4 p0, p1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x0 = v[2 * e], x1 = v[2 * e + 1], y0 = v[8 + 2 * e], y1 = v[8 + 2 * e + 1];
            p0[e] = pack2<T>(fmaxf(x0, 0.f) + slope * fminf(x0, 0.f), fmaxf(x1, 0.f) + slope * fminf(x1, 0.f));
            p1[e] = pack2<T>(fmaxf(y0, 0.f) + slope * fminf(y0, 0.f), fmaxf(y1, 0.f) + slope * fminf(y1, 0.f));
          }
          *(u32x4*)(outp + off) = p0;
          *(u32x4*)(outp + off + 8) = p1;
        }
      });
    });
    tile = next;
  }
}

int t3_mode() {
  const char* e = getenv("FSR_TALL3");   // 0: off (A/B against conv_igemm.hip); 1 (default): on; 2: prefer the 4-wave 128-channel tile
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

template <typename T, int BN, int NW, int G, int NSLOT>
int t3_launch(ConvKArgs& a, int wg_per_cu, hipStream_t stream) {
  auto kern = conv_tall3_kernel<T, BN, NW, G, NSLOT>;
  constexpr int lds = t3_lds_bytes<BN, G, NSLOT>();
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
  long long grid = (long long)t3_cus() * wg_per_cu;
  if (grid > ntiles) grid = ntiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, stream, a);
  fsr_note_kernel("conv_tall3_kernel<%s,%d,%d,%d,%d>", std::is_same<T, f16_t>::value ? "f16" : "bf16", BN, NW, G, NSLOT);
  const int rc = fsr_check_launch("conv_tall3_kernel");
  return rc ? rc : 1;
}

}  // namespace

// 1 = launched, 0 = not this kernel's shape (the caller falls through to conv_igemm.hip), < 0 = error.
int fsr_conv_tall3_try(int dtype, ConvKArgs& a, int S, hipStream_t stream) {
  const int mode = t3_mode();
  if (mode == 0 || (dtype != FSR_BF16 && dtype != FSR_F16) || S != 1 || a.ntaps != 9) return 0;
  if (a.Cin < 128 || a.Cin % 32 != 0 || a.Cout % 128 != 0 || a.CoutPad != a.Cout) return 0;
  if (a.stats || a.preact || a.oscale || a.ps || a.in_ps || a.out_f32) return 0;
  if (a.act != FSR_ACT_NONE && a.act != FSR_ACT_RELU && a.act != FSR_ACT_LEAKY) return 0;
  if (a.osy != 1 || a.osx != 1 || a.ooy != 0 || a.oox != 0 || a.org_y != -1 || a.org_x != -1) return 0;
  if (a.pool2 && (a.dmask || (a.GH & 1) || (a.GW & 1))) return 0;
  if ((a.GH & 15) > 0 && (a.GH & 15) <= 8 && !(mode & 4)) return 0;    // 24-row maps: half of every second 16-row tile is padding
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
  a.tiles_y = (a.GH + 15) / 16;
  const bool wide = a.Cout % 256 == 0 && !(mode & 2);
  if (dtype == FSR_F16) {
    if (wide) return t3_launch<f16_t, 256, 8, 3, 2>(a, 1, stream);
    return t3_launch<f16_t, 128, 4, 1, 4>(a, 2, stream);
  }
  if (wide) return t3_launch<bf16_t, 256, 8, 3, 2>(a, 1, stream);
  return t3_launch<bf16_t, 128, 4, 1, 4>(a, 2, stream);
}
