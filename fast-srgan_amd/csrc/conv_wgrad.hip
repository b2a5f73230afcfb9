// Weight gradient of the 3x3 convolutions as a split-K implicit GEMM on the matrix cores.
//
//   dW[tap][co][ci] = sum over (image, output pixel) of dy[pixel][co] * x[pixel*S + tap - 1][ci]
//
// i.e. autograd's weight gradient of every Conv2d(k=3,p=1) the reference trains:
//   /root/reference/model.py:30-35, 47-64, 75-76, 86-93, 103-108 (Generator)
//   /root/reference/model.py:124-131, 143-183 (Discriminator)
//
// GEMM view: M = Cout, N = Cin (x 9 taps), K = N*OH*OW output pixels.  The contraction index is
// the PIXEL, which is the strided dimension of NHWC tensors, so both MFMA operands have to be
// transposed on the way from LDS to registers:
//   bf16 : ds_read_b64_tr_b16 -- each 16-lane group reads a [4 pixels][16 channels] block and every
//          lane receives one channel's 4 pixels; two reads feed one v_mfma_f32_16x16x32_bf16;
//   f32  : v_mfma_f32_16x16x4_f32 takes ONE element per lane, lane (l&15, l>>4) = (channel, pixel):
//          a plain ds_read_b32 with the 16 channel lanes on consecutive banks.
// Work decomposition
//   workgroup -> BM output channels x BN input channels x all 9 taps, for a slab of pixel tiles
//                (TPH x 16 output pixels each); the dy tile and the x halo of a pixel tile are
//                staged in LDS once and reused by all 9 taps (x) / all taps and channel tiles (dy);
//   wave      -> 128x64 blocks (16-bit, Cout % 128 == 0; 8 waves): FOUR co tiles x one ci tile x 9 taps (144 accumulator
//                registers), 26 transposing reads per 36 MFMAs, x fetched Cout / 128 times;
//                64x64 blocks (8 waves, two per SIMD): TWO co tiles x one ci tile x 9 taps (72 accumulator registers), 22
//                reads per 18 MFMAs (one co tile x two ci tiles needs 38: measured 10-20 % slower);
//                other blocks (thin layers, f32 edge cases): one co tile x TPW ci tiles x 9 taps;
//   staging   -> 16-bit 8-wave blocks: LDS-DMA (buffer_load ... lds) into two LDS images, tile i+1 in flight while tile i is
//                multiplied, one barrier per tile, the tile's 36 (K step, tap) stages unrolled with a two-stage read
//                lookahead (stride 2: 4-row tiles so that two images fit); everything else: register-prefetched tiles, two
//                barriers per tile;
//   split-K   -> slabs write f32 partials [slab][9][cout_pad][cin_pad] to the workspace; a second
//                kernel sums the slabs and adds the result into the OIHW float gradient (no atomics: bit-reproducible).
// Measurements, ablations and what bounds the kernel now: profiles/r03_wgrad_ablation.txt, DESIGN.md 3.2.
#include "fsr_common.h"
#include "fsr_host.h"
#include <cstdlib>

constexpr int WGRAD_GROUP_MAX = 32;

#ifndef FSR_ABLW
#define FSR_ABLW 0   // ablation builds (tools/build_variant.sh; results are WRONG on purpose): 1 no staging after a slab's first
#endif               // tile, 2 no LDS -> MFMA phase (memory only), 3 MFMAs on the first fragments only (no transposing reads in the loop),
                     // 4 no DMA wait, 5 no barrier, 6 neither, 7 DMA pieces with every lane out of range (LDS writes without memory traffic)

struct WgradKArgs {
  const void* x;
  const void* dy;
  float* ws;
  int N, IH, IW, CinPad;
  int OH, OW, CoutPad;
  int dy_ps;
  int tiles_x, tiles_y, tiles_total, tiles_per_slab, nslab;
  int nbm, nbn;
  // grouped launch (group_n > 0): group_n layers of ONE shape share the launch; workgroup b serves layer b / nslab, slab
  // b % nslab of that layer (nbm = nbn = 1), operands gx[layer] / gdy[layer], partials at ws + layer * nslab * 9 * Cout * Cin
  int x3;   // FSR_X3 on the physical bf16 views: channels come in (hi 32 | lo 32) groups; the lo x lo product blocks are not computed
  int group_n;
  const void* gx[WGRAD_GROUP_MAX];
  const void* gdy[WGRAD_GROUP_MAX];
};

struct WgradGroupOut {
  float* dw[WGRAD_GROUP_MAX];
};

// Staging by LDS-DMA (16-bit, stride 1, 8-wave blocks): the dy tile and the x halo of tile i+1 go global -> LDS with
// buffer_load ... lds into the OTHER of two LDS images while tile i is multiplied -- no staging registers, no commit pass,
// one barrier per tile.  A wave instruction fills 1 KB of LDS linearly, i.e. 64 consecutive 16-byte units of the PADDED pixel
// rows: lane -> (pixel, unit) by one division, the pad units and the out-of-image pixels get an out-of-range offset (zeros).
template <typename T, int BM, int BN, int S, int TPH>
constexpr bool wgrad_dma() {   // stride 2: the halo of an 8-row tile is 90 KB, two images only fit with 4-row tiles
  return sizeof(T) == 2 && BM >= 64 && BN == 64 && (S == 1 || TPH == 4);
}
template <typename T, int BM, int BN, int S, int TPH>
constexpr int wgrad_dma_dy_instr() {
  return (TPH * 16 * ((BM + 16) / (16 / (int)sizeof(T))) + 63) / 64;
}
template <typename T, int BM, int BN, int S, int TPH>
constexpr int wgrad_dma_halo_instr() {
  return (((TPH - 1) * S + 3) * (15 * S + 3) * ((BN + 16) / (16 / (int)sizeof(T))) + 63) / 64;
}
template <typename T, int BM, int BN, int S, int TPH>
constexpr size_t wgrad_dma_lds_bytes() {   // two images of (dy tile, halo), each region rounded up to whole 1 KB DMA pieces
  return (size_t)2 * 1024 * (wgrad_dma_dy_instr<T, BM, BN, S, TPH>() + wgrad_dma_halo_instr<T, BM, BN, S, TPH>());
}

template <typename T, int BM, int BN, int S, int TPH, bool X3W = false>
__global__ __launch_bounds__((BM >= 64 && BN == 64) ? 512 : 256) void conv_wgrad_kernel(const WgradKArgs a) {
  static_assert(!X3W || (BM == 128 && BN == 64 && sizeof(T) == 2), "the x3 form of the 128 x 64 block");
  constexpr int NW = (BM >= 64 && BN == 64) ? 8 : 4;   // waves per workgroup
  static_assert(BM <= 64 || (BM == 128 && BN == 64 && sizeof(T) == 2), "128-row blocks: 16-bit, 64 input channels");
  constexpr int NTHR = NW * 64;
  constexpr int EPB = 16 / (int)sizeof(T);
  constexpr int PAD = 16;
  constexpr int PA = BM + PAD, PB = BN + PAD;  // LDS pixel pitches (elements)
  constexpr int HH = (TPH - 1) * S + 3, HW = 15 * S + 3;
  constexpr int NPAIR = (BM / 16) * (BN / 16);
  constexpr int TPW = (NPAIR + NW - 1) / NW;  // (co tile, ci tile) pairs per wave; all share one co tile
  constexpr int NBT = BN / 16;
  static_assert(TPW <= NBT, "a wave's pairs must share its co tile");
  constexpr int TPIX = TPH * 16;

  HIP_DYNAMIC_SHARED(char, smem)
  T* dyt = (T*)smem;              // [TPIX][PA]
  T* halo = dyt + TPIX * PA;      // [HH*HW][PB]   (LDS-DMA form: two such images, see below)
  constexpr bool DMA = wgrad_dma<T, BM, BN, S, TPH>();

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  // the nbm x nbn workgroups of one slab read the same dy / x tiles: consecutive logical ids -> one XCD -> one L2
  int bid = a.group_n > 0 ? (int)blockIdx.x : xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int layer = a.group_n > 0 ? bid / a.nslab : 0;
  if (a.group_n > 0) bid -= layer * a.nslab;
  const int bn = bid % a.nbn;
  bid /= a.nbn;
  const int bm = bid % a.nbm;
  const int slab = bid / a.nbm;

  // 64x64 blocks (8 waves): wave -> (pair of co tiles, one ci tile): a transposed x fragment feeds TWO MFMAs, 22 instead
  // of 38 transposing reads per 18 MFMAs.  128x64 blocks: wave -> (FOUR co tiles, one ci tile), 26 reads per 36 MFMAs and
  // 144 accumulator registers.  Other blocks: wave -> (one co tile, TPW ci tiles).
  constexpr bool COPAIR = (BM >= 64 && BN == 64);
  const bool active = wave * TPW < NPAIR;
  const int co_t = COPAIR ? (wave / NBT) * TPW : (wave * TPW) / NBT;
  // x3, 128 x 64 blocks: a wave's four co tiles are (hi, hi, lo, lo) of one channel group and ci tiles 0, 1 / 2, 3 the hi / lo part
  // of the input group.  A wave on a lo ci tile skips its two lo co tiles (the dropped lo x lo block); the ci tiles are dealt so
  // that the two waves of a SIMD (w, w + 4) are one hi and one lo wave: 6 instead of 8 MFMAs per stage and SIMD.
  constexpr bool x3blk = X3W;
  const int ci_t0 = COPAIR ? (x3blk ? (wave + 2 * (wave / NBT)) % NBT : wave % NBT) : (wave * TPW) % NBT;
  const bool skip_lo = x3blk && ci_t0 >= 2;

  f32x4 acc[9][TPW];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // (a select chain over constant indices: indexing the by-value tables with `layer` would copy them to scratch)
  const void* xsel = a.x;
  const void* dysel = a.dy;
  if (a.group_n > 0) {
#pragma unroll
    for (int l = 0; l < WGRAD_GROUP_MAX; ++l)
      if (l == layer) {
        xsel = a.gx[l];
        dysel = a.gdy[l];
      }
  }
  const T* xg = (const T*)xsel;
  const T* dyg = (const T*)dysel;
  const int tile0 = slab * a.tiles_per_slab;
  const int tile1 = (tile0 + a.tiles_per_slab < a.tiles_total) ? tile0 + a.tiles_per_slab : a.tiles_total;

  // Staging is split (issue early / commit late): the global loads of tile i+1 are issued into registers
  // before the MFMAs of tile i and written to LDS after them, so HBM/L2 latency hides under a tile of math.
  constexpr int DUNITS = TPIX * (BM / EPB), HUNITS = HH * HW * (BN / EPB);
  constexpr int DPT = (DUNITS + NTHR - 1) / NTHR, HPT = (HUNITS + NTHR - 1) / NTHR;
  u32x4 dreg[DPT], hreg[HPT];
  auto stage_issue = [&](int tile) {
    const int tx = tile % a.tiles_x;
    const int ty = (tile / a.tiles_x) % a.tiles_y;
    const int img = tile / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * TPH, ox0 = tx * 16;
#pragma unroll
    for (int i = 0; i < DPT; ++i) {   // dy tile: [TPIX][BM]
      const int u = tid + i * NTHR;
      u32x4 v = (u32x4){0u, 0u, 0u, 0u};
      if (DUNITS % NTHR == 0 || u < DUNITS) {
        const int unit = u % (BM / EPB), p = u / (BM / EPB);
        const int oy = oy0 + p / 16, ox = ox0 + (p & 15);
        if (oy < a.OH && ox < a.OW) {
          const int ch = bm * BM + unit * EPB;
          unsigned eoff;
          if (!a.dy_ps) {
            eoff = (unsigned)((img * a.OH + oy) * a.OW + ox) * (unsigned)a.CoutPad + (unsigned)ch;
          } else {
            const int cps = a.CoutPad >> 2;
            const int q = ch / cps, cc = ch - q * cps;
            eoff = (unsigned)((img * 2 * a.OH + 2 * oy + (q >> 1)) * (2 * a.OW) + 2 * ox + (q & 1)) * (unsigned)cps + (unsigned)cc;
          }
          v = *(const u32x4*)(dyg + eoff);
        }
      }
      dreg[i] = v;
    }
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
#pragma unroll
    for (int i = 0; i < HPT; ++i) {   // x halo: [HH][HW][BN]
      const int u = tid + i * NTHR;
      u32x4 v = (u32x4){0u, 0u, 0u, 0u};
      if (HUNITS % NTHR == 0 || u < HUNITS) {
        const int unit = u % (BN / EPB), p = u / (BN / EPB);
        const int iy = iy0 + p / HW, ix = ix0 + p % HW;
        if (iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW)
          v = *(const u32x4*)(xg + (unsigned)((img * a.IH + iy) * a.IW + ix) * (unsigned)a.CinPad + (unsigned)(bn * BN + unit * EPB));
      }
      hreg[i] = v;
    }
  };
  auto stage_commit = [&]() {
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
      const int u = tid + i * NTHR;
      if (DUNITS % NTHR == 0 || u < DUNITS) *(u32x4*)(dyt + (u / (BM / EPB)) * PA + (u % (BM / EPB)) * EPB) = dreg[i];
    }
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
      const int u = tid + i * NTHR;
      if (HUNITS % NTHR == 0 || u < HUNITS) *(u32x4*)(halo + (u / (BN / EPB)) * PB + (u % (BN / EPB)) * EPB) = hreg[i];
    }
  };

  // ---- LDS-DMA staging: per wave IPW pieces of 1 KB; piece g = j * NW + wave of the image [dy region | halo region]
  constexpr int UPA = PA / EPB, UPB = PB / EPB;
  constexpr int DYI = wgrad_dma_dy_instr<T, BM, BN, S, TPH>(), HAI = wgrad_dma_halo_instr<T, BM, BN, S, TPH>();
  constexpr int NPIECE = DYI + HAI, IPW = (NPIECE + NW - 1) / NW;
  constexpr unsigned IMG_BYTES = NPIECE * 1024u;
  unsigned rel[DMA ? IPW : 1], pk[DMA ? IPW : 1];   // per piece and lane: element offset relative to the tile origin; py | px << 8 | invalid << 31
  if constexpr (DMA) {
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      const int g = j * NW + wave;
      unsigned r_ = 0, k_ = 0x80000000u;
      if (g < DYI) {
        const int slot = g * 64 + lane, p_ = slot / UPA, u_ = slot - p_ * UPA;
        const int py = p_ >> 4, px = p_ & 15;
        const int ch = bm * BM + u_ * EPB;
        if (!a.dy_ps) {
          r_ = (unsigned)((py * a.OW + px) * a.CoutPad + ch);
        } else {
          const int cps = a.CoutPad >> 2;
          const int q = ch / cps, cc = ch - q * cps;
          r_ = (unsigned)(((2 * py + (q >> 1)) * (2 * a.OW) + 2 * px + (q & 1)) * cps + cc);
        }
        k_ = (unsigned)py | ((unsigned)px << 8) | ((u_ >= BM / EPB || p_ >= TPIX) ? 0x80000000u : 0u);
      } else if (g < NPIECE) {
        const int slot = (g - DYI) * 64 + lane, p_ = slot / UPB, u_ = slot - p_ * UPB;
        const int py = p_ / HW, px = p_ - py * HW;
        r_ = (unsigned)((py * a.IW + px) * a.CinPad + bn * BN + u_ * EPB);
        k_ = (unsigned)py | ((unsigned)px << 8) | ((u_ >= BN / EPB || p_ >= HH * HW) ? 0x80000000u : 0u);
      }
      rel[j] = r_;
      pk[j] = k_;
    }
  }
  const fsr_buf_t dy_buf = fsr_make_buf(dyg, DMA ? (unsigned)((size_t)a.N * a.OH * a.OW * a.CoutPad * sizeof(T)) : 0u);
  const fsr_buf_t x_buf = fsr_make_buf(xg, DMA ? (unsigned)((size_t)a.N * a.IH * a.IW * a.CinPad * sizeof(T)) : 0u);
  const fsr_lds_addr_t lds0 = FSR_LDS_ADDR(smem);
  struct DmaTile {   // wave-uniform description of the tile being fetched
    int oy0, ox0, iy0, ix0;
    unsigned dy_base, x_base;
    fsr_lds_addr_t lds;
    bool on;
  };
  auto dma_tile = [&](int tile, int image, bool on) {
    DmaTile d;
    const int tx = tile % a.tiles_x;
    const int ty = (tile / a.tiles_x) % a.tiles_y;
    const int img = tile / (a.tiles_x * a.tiles_y);
    d.oy0 = ty * TPH;
    d.ox0 = tx * 16;
    d.dy_base = !a.dy_ps ? (unsigned)((img * a.OH + d.oy0) * a.OW + d.ox0) * (unsigned)a.CoutPad
                         : (unsigned)((img * 2 * a.OH + 2 * d.oy0) * (2 * a.OW) + 2 * d.ox0) * (unsigned)(a.CoutPad >> 2);
    d.iy0 = d.oy0 * S - 1;
    d.ix0 = d.ox0 * S - 1;
    d.x_base = (unsigned)((img * a.IH + d.iy0) * a.IW + d.ix0) * (unsigned)a.CinPad;   // may wrap; base + rel of a valid pixel does not
    d.lds = lds0 + (fsr_lds_addr_t)(image * IMG_BYTES);
    d.on = on;
    return d;
  };
  auto dma_piece = [&](const DmaTile& d, int j) {   // j: compile-time after unrolling
    const int g = j * NW + wave;
    if (!d.on || (NPIECE % NW != 0 && g >= NPIECE)) return;
    const int py = (int)(pk[j] & 0xffu), px = (int)((pk[j] >> 8) & 0xffu);
#if FSR_ABLW == 7
    const bool keep = false;             // every lane out of range: the pieces still write (zeros) to LDS, no memory traffic
#else
    const bool keep = !(pk[j] >> 31);
#endif
    if (g < DYI) {
      const bool ok = keep && d.oy0 + py < a.OH && d.ox0 + px < a.OW;
      const unsigned voff = ok ? (d.dy_base + rel[j]) * (unsigned)sizeof(T) : 0xffffffffu;
      FSR_BLDS16(dy_buf, voff, 0u, d.lds + (fsr_lds_addr_t)(g * 1024));
    } else {
      const bool ok = keep && (unsigned)(d.iy0 + py) < (unsigned)a.IH && (unsigned)(d.ix0 + px) < (unsigned)a.IW;
      const unsigned voff = ok ? (d.x_base + rel[j]) * (unsigned)sizeof(T) : 0xffffffffu;
      FSR_BLDS16(x_buf, voff, 0u, d.lds + (fsr_lds_addr_t)(g * 1024));
    }
  };
  auto dma_issue = [&](int tile, int image) {
    const DmaTile d = dma_tile(tile, image, true);
#pragma unroll
    for (int j = 0; j < IPW; ++j) dma_piece(d, j);
  };

  int image = 0;
  if constexpr (DMA) {
    if (tile0 < tile1) dma_issue(tile0, 0);
  } else {
    if (tile0 < tile1) stage_issue(tile0);
  }
  for (int tile = tile0; tile < tile1; ++tile) {
    if constexpr (DMA) {
#if FSR_ABLW != 4 && FSR_ABLW != 6
      FSR_WAIT_VM(0);     // this wave's pieces of the tile have landed ...
#endif
#if FSR_ABLW != 5 && FSR_ABLW != 6
      FSR_BARRIER();      // ... and everybody's; all waves are also done with the other image
#endif
      // the next tile's pieces, all at once (spreading them over the tile's stages measured +-0: what the traffic costs is
      // shader clock and memory-system time, not issue slots or LDS write cycles -- profiles/r03_wgrad_ablation.txt)
#if FSR_ABLW != 1
      if (tile + 1 < tile1) dma_issue(tile + 1, image ^ 1);
#endif
      dyt = (T*)(smem + image * IMG_BYTES);
      halo = (T*)(smem + image * IMG_BYTES + DYI * 1024);
      image ^= 1;
    } else {
      __syncthreads();  // the previous tile's LDS images are dead
#if FSR_ABLW == 1
      if (tile == tile0) stage_commit();
      __syncthreads();
#else
      stage_commit();
      __syncthreads();
      if (tile + 1 < tile1) stage_issue(tile + 1);
#endif
    }
    if (!active) continue;
#if FSR_ABLW == 2
    continue;
#endif

    if constexpr (sizeof(T) == 2 && COPAIR) {
      // The 8-wave blocks: the tile's NS * 9 (K step, tap) stages unrolled and software-pipelined.  A stage = one transposed
      // x fragment (2 reads) against the wave's TPW dy fragments (TPW MFMAs = TPW * 16 matrix-pipe cycles): the x fragment of
      // stage i + 2 is requested before the MFMAs of stage i (ring of three), the dy fragments of K step s + 1 in the middle
      // of step s (two sets), so a transposing read has two stages -- 130 / 260 cycles -- to come back.  K step = 32 pixels =
      // tile rows 2s, 2s+1; lane group g owns pixels k = 8g..8g+7: row 2s + (g>>1), columns 8(g&1) .. +7; read h (0/1)
      // fetches pixels 4h..4h+3: group-lane q supplies pixel 4h + (q>>2), channel chunk q&3.
      constexpr int NS = TPH / 2, NSTAGE = NS * 9;
      const int qrow = l15 >> 2, qch = (l15 & 3) * 4, c0 = 8 * (lg & 1);
      const T* pa_base = dyt + ((lg >> 1) * 16 + c0 + qrow) * PA + co_t * 16 + qch;
      const T* pb_base = halo + ((lg >> 1) * S * HW + (c0 + qrow) * S) * PB + ci_t0 * 16 + qch;
      auto frag = [&](const T* p0, int second) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(FSR_LDS_PTR(s16x4, p0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(FSR_LDS_PTR(s16x4, p0 + second));
        return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      };
      auto xfrag = [&](int i) {   // stage i = (K step i / 9, tap i % 9)
        const int ks = i / 9, t = i % 9;
        return frag(pb_base + ((ks * 2 * S + t / 3) * HW + t % 3) * PB, 4 * S * PB);
      };
      // (x3: a wave on a lo ci tile leaves out its two lo co tiles -- ONE code path with a wave-uniform skip around their reads and
      // MFMAs; two specialised copies of the unrolled stage loop made hipcc spill 220 registers)
      constexpr int NWH = (BM == 128) ? 2 : TPW;       // co tiles every wave multiplies
      s16x8 af[2][TPW], bf[3];
#pragma unroll
      for (int w = 0; w < TPW; ++w) af[0][w] = frag(pa_base + w * 16, 4 * PA);
      bf[0] = xfrag(0);
      bf[1] = xfrag(1);
#pragma unroll
      for (int i = 0; i < NSTAGE; ++i) {
        const int ks = i / 9, t = i % 9;
        if (i + 2 < NSTAGE) bf[(i + 2) % 3] = xfrag(i + 2);
        if (t == 4 && ks + 1 < NS) {
#pragma unroll
          for (int w = 0; w < NWH; ++w) af[(ks + 1) & 1][w] = frag(pa_base + (ks + 1) * 32 * PA + w * 16, 4 * PA);
          if (!skip_lo) {
#pragma unroll
            for (int w = NWH; w < TPW; ++w) af[(ks + 1) & 1][w] = frag(pa_base + (ks + 1) * 32 * PA + w * 16, 4 * PA);
          }
        }
#pragma unroll
        for (int w = 0; w < NWH; ++w) acc[t][w] = mfma16<T>(af[ks & 1][w], bf[i % 3], acc[t][w]);
        if (!skip_lo) {
#pragma unroll
          for (int w = NWH; w < TPW; ++w) acc[t][w] = mfma16<T>(af[ks & 1][w], bf[i % 3], acc[t][w]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (sizeof(T) == 2) {
      // K step = 32 pixels = tile rows 2s, 2s+1; lane group g owns pixels k = 8g..8g+7:
      // row 2s + (g>>1), columns 8(g&1) .. +7.  A transposing read h (0/1) fetches pixels 4h..4h+3:
      // group-lane q supplies pixel 4h + (q>>2), channel chunk q&3.
      const int qrow = l15 >> 2, qch = (l15 & 3) * 4;
#pragma unroll 1
      for (int s = 0; s < TPH / 2; ++s) {
        const int r = 2 * s + (lg >> 1);
        const int c0 = 8 * (lg & 1);
        constexpr int NA = COPAIR ? TPW : 1;       // co tiles per wave
        s16x8 af[NA];
#pragma unroll
        for (int w = 0; w < NA; ++w) {
          const T* pa = dyt + (r * 16 + c0 + qrow) * PA + (co_t + w) * 16 + qch;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(FSR_LDS_PTR(s16x4, pa));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(FSR_LDS_PTR(s16x4, pa + 4 * PA));
          af[w] = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int ky = t / 3, kx = t % 3;
          const T* pb0 = halo + ((r * S + ky) * HW + (c0 + qrow) * S + kx) * PB + ci_t0 * 16 + qch;
          if constexpr (COPAIR) {
#if FSR_ABLW == 3
            const s16x8 bf = af[(t + 1) & 1];
#else
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(FSR_LDS_PTR(s16x4, pb0));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(FSR_LDS_PTR(s16x4, pb0 + 4 * S * PB));
            const s16x8 bf = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#endif
#pragma unroll
            for (int w = 0; w < TPW; ++w) acc[t][w] = mfma16<T>(af[w], bf, acc[t][w]);
          } else {
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
              const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(FSR_LDS_PTR(s16x4, pb0 + j * 16));
              const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(FSR_LDS_PTR(s16x4, pb0 + j * 16 + 4 * S * PB));
              const s16x8 bf = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
              acc[t][j] = mfma16<T>(af[0], bf, acc[t][j]);
            }
          }
        }
      }
    } else {
      // K step = 4 pixels of one row: pixel k = lg -> (row s>>2, column 4(s&3) + lg)
#pragma unroll 1
      for (int s = 0; s < TPH * 4; ++s) {
        const int r = s >> 2, c = 4 * (s & 3) + lg;
        const float av = dyt[(r * 16 + c) * PA + co_t * 16 + l15];
        const float av1 = COPAIR ? dyt[(r * 16 + c) * PA + (co_t + 1) * 16 + l15] : 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int ky = t / 3, kx = t % 3;
          const T* pb = halo + ((r * S + ky) * HW + c * S + kx) * PB + ci_t0 * 16 + l15;
          if constexpr (COPAIR) {
            const float bv = pb[0];
            acc[t][0] = mfma_f32_16x16x4(av, bv, acc[t][0]);
            acc[t][1] = mfma_f32_16x16x4(av1, bv, acc[t][1]);
          } else {
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[t][j] = mfma_f32_16x16x4(av, pb[j * 16], acc[t][j]);
          }
        }
      }
    }
  }

  if (!active) return;
  // partial tile -> workspace [slab][tap][cout_pad][cin_pad]; D layout: lane column = ci, rows 4*lg+r = co
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int cot = COPAIR ? co_t + j : co_t, cit = COPAIR ? ci_t0 : ci_t0 + j;
      float* o = a.ws + (((size_t)(layer * a.nslab + slab) * 9 + t) * a.CoutPad + bm * BM + cot * 16 + lg * 4) * a.CinPad + bn * BN + cit * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[(size_t)r * a.CinPad] = acc[t][j][r];
    }
}

// dw[co][ci][tap] += sum_slab ws[slab][tap][row(co)][ci]; row(co) undoes the pixel-shuffle row order.
// Block = 32 consecutive (tap, co, ci) elements (coalesced along ci) x 8 slab lanes: lane j adds slabs j, j+8, ... in
// order, the eight lane sums are added in order and ONE thread adds the result to dw -- no atomics, so the weight
// gradient is bit-reproducible (the split-K partials themselves are plain stores of the weight-gradient kernel).
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nslab,
                                                                int cout, int cin, int cout_pad, int cin_pad, int ps) {
  __shared__ float red[8][32];
  const int total = 9 * cout * cin;
  const size_t sstride = (size_t)9 * cout_pad * cin_pad;
  const int oi = threadIdx.x & 31, pl = threadIdx.x >> 5;
  for (int i0 = blockIdx.x * 32; i0 < total; i0 += gridDim.x * 32) {
    const int i = i0 + oi;
    float s = 0.f;
    int co = 0, ci = 0, t = 0;
    if (i < total) {
      ci = i % cin;
      co = (i / cin) % cout;
      t = i / (cin * cout);
      const int row = ps ? (co & 3) * (cout_pad >> 2) + (co >> 2) : co;
      const float* p = ws + ((size_t)t * cout_pad + row) * cin_pad + ci;
      int k = pl;
      for (; k + 24 < nslab; k += 32) {   // four independent loads in flight per lane, added in order
        const float v0 = p[(size_t)k * sstride], v1 = p[(size_t)(k + 8) * sstride];
        const float v2 = p[(size_t)(k + 16) * sstride], v3 = p[(size_t)(k + 24) * sstride];
        s = ((s + v0) + v1) + v2 + v3;
      }
      for (; k < nslab; k += 8) s += p[(size_t)k * sstride];
    }
    red[pl][oi] = s;
    __syncthreads();
    if (pl == 0 && i < total) {
      float r = red[0][oi];
#pragma unroll
      for (int j = 1; j < 8; ++j) r += red[j][oi];
      dw[((size_t)co * cin + ci) * 9 + t] += r;
    }
    __syncthreads();
  }
}

// FSR_X3: the weight-gradient kernel ran on the bf16 views of x and dy (2 x the channels, hi / lo chunks of 32), so the workspace
// holds every (dy part, x part) product block: dW[co][ci] = P[hi(co)][hi(ci)] + P[hi(co)][lo(ci)] + P[lo(co)][hi(ci)], with
// hi(c) = (c >> 5) * 64 + (c & 31) and lo(c) = hi(c) + 32 (the lo x lo block, <= 2^-16 of the result, is computed and dropped:
// the blocks of a workgroup share its staging, skipping its MFMAs would only unbalance the waves).  Slab order and the order of
// the three terms are fixed: bit-reproducible.  cout_pad / cin_pad are the PHYSICAL workspace dims, cout / cin logical.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_x3_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nslab,
                                                                   int cout, int cin, int cout_pad, int cin_pad, int ps) {
  __shared__ float red[8][32];
  const int total = 9 * cout * cin;
  const size_t sstride = (size_t)9 * cout_pad * cin_pad;
  const int oi = threadIdx.x & 31, pl = threadIdx.x >> 5;
  for (int i0 = blockIdx.x * 32; i0 < total; i0 += gridDim.x * 32) {
    const int i = i0 + oi;
    float s = 0.f;
    int co = 0, ci = 0, t = 0;
    if (i < total) {
      ci = i % cin;
      co = (i / cin) % cout;
      t = i / (cin * cout);
      const int row = ps ? (co & 3) * (cout_pad >> 3) + (co >> 2) : co;     // logical row (cout_pad / 2 logical rows, a quarter per quadrant)
      const int rh = (row >> 5) * 64 + (row & 31), ch = (ci >> 5) * 64 + (ci & 31);
      const float* p = ws + ((size_t)t * cout_pad + rh) * cin_pad + ch;
      const size_t rlo = (size_t)32 * cin_pad;
      float hh = 0.f, hl = 0.f, lh = 0.f;
      for (int k = pl; k < nslab; k += 8) {
        const float* q = p + (size_t)k * sstride;
        hh += q[0];
        hl += q[32];
        lh += q[rlo];
      }
      s = hh + (hl + lh);
    }
    red[pl][oi] = s;
    __syncthreads();
    if (pl == 0 && i < total) {
      float r = red[0][oi];
#pragma unroll
      for (int j = 1; j < 8; ++j) r += red[j][oi];
      dw[((size_t)co * cin + ci) * 9 + t] += r;
    }
    __syncthreads();
  }
}

namespace {

// FSR_X3 -> the bf16 launch on the physical views (see conv_wgrad_reduce_x3_kernel)
inline fsr_wgrad_desc physical_desc(const fsr_wgrad_desc* d) {
  fsr_wgrad_desc q = *d;
  if (d->dtype == FSR_X3) {
    q.dtype = FSR_BF16;
    q.cin_pad = 2 * d->cin_pad;
    q.cout_pad = 2 * d->cout_pad;
    q.cin = q.cin_pad;
    q.cout = q.cout_pad;
  }
  return q;
}

struct WgradPlan {
  int BM, BN, TPH, S;
  int tiles_x, tiles_y, tiles_total, tiles_per_slab, nslab, nbm, nbn;
  size_t lds;
};

int make_plan(const fsr_wgrad_desc* d_in, WgradPlan& p) {
  if (!d_in) return fsr_fail(-1, "conv3x3_wgrad: null descriptor");
  if (d_in->dtype == FSR_X3 && (d_in->cin_pad % 32 || d_in->cout_pad % 32 || d_in->cin > d_in->cin_pad || d_in->cout > d_in->cout_pad ||
                                (d_in->dy_pixel_shuffled && (d_in->cout_pad / 4) % 32)))
    return fsr_fail(-2, "conv3x3_wgrad: x3 tensors have a multiple of 32 channels");
  const fsr_wgrad_desc dphys = physical_desc(d_in);
  const fsr_wgrad_desc* d = &dphys;
  if (d->dtype != FSR_F32 && d->dtype != FSR_BF16 && d->dtype != FSR_F16) return fsr_fail(-2, "conv3x3_wgrad: unknown dtype %d", d->dtype);
  const int cpad = d->dtype != FSR_F32 ? 32 : 16;
  if (d->stride != 1 && d->stride != 2) return fsr_fail(-2, "conv3x3_wgrad: stride must be 1 or 2");
  if (d->cin_pad % cpad || d->cin_pad <= 0) return fsr_fail(-2, "conv3x3_wgrad: cin_pad %d is not a multiple of %d", d->cin_pad, cpad);
  if (d->cout_pad % 16 || d->cout_pad <= 0) return fsr_fail(-2, "conv3x3_wgrad: cout_pad %d is not a multiple of 16", d->cout_pad);
  if (d->dtype != FSR_F32 && d->cout_pad % 8) return fsr_fail(-2, "conv3x3_wgrad: bad cout_pad");
  if (d->cin > d->cin_pad || d->cout > d->cout_pad || d->cin <= 0 || d->cout <= 0) return fsr_fail(-2, "conv3x3_wgrad: bad channel counts");
  if (d->oh != (d->ih - 1) / d->stride + 1 || d->ow != (d->iw - 1) / d->stride + 1)
    return fsr_fail(-2, "conv3x3_wgrad: output dims do not match k=3,p=1,stride=%d", d->stride);
  if (d->dy_pixel_shuffled && (d->cout_pad % 64 || d->cout != d->cout_pad))
    return fsr_fail(-2, "conv3x3_wgrad: pixel-shuffled dy needs cout %% 64 == 0");
  if ((long long)d->n * d->ih * d->iw * d->cin_pad >= (1LL << 31) || (long long)d->n * d->oh * d->ow * d->cout_pad >= (1LL << 31))
    return fsr_fail(-2, "conv3x3_wgrad: tensors with 2^31 or more elements are not supported");
  p.S = d->stride;
  p.BM = (d->cout_pad % 64 == 0) ? 64 : 16;
  p.BN = (d->cin_pad % 64 == 0) ? 64 : (d->cin_pad % 32 == 0 ? 32 : 16);
  // 128 output channels per workgroup where the layer has them: x is fetched Cout / BM times and dy Cin / BN times, and the
  // 64 x 64 block spends 60 % of its time on that traffic (FSR_WGRAD_BM=64 keeps the 64-row block for an A/B)
  const char* ebm = getenv("FSR_WGRAD_BM");
  const bool bm64 = ebm && atoi(ebm) == 64;
  if (d->dtype != FSR_F32 && p.BN == 64 && d->cout_pad % 128 == 0 && !bm64) p.BM = 128;
  // pixel-shuffled dy: the staging resolves the quadrant per 16-byte unit, so a block may span quadrant slices as long as
  // every 64-row part of it lies inside one
  if (d->dy_pixel_shuffled && (d->cout_pad / 4) % (p.BM == 128 ? 64 : p.BM)) p.BM = 16;
  p.TPH = d->dtype != FSR_F32 ? 8 : 4;
  {
    const char* e2 = getenv("FSR_WGRAD_S2");
    if (d->dtype != FSR_F32 && p.S == 2 && p.BM >= 64 && p.BN == 64 && !(e2 && atoi(e2) == 8)) p.TPH = 4;   // LDS-DMA form, 4-row tiles
    if (p.S == 2 && p.TPH == 8 && p.BM == 128) p.BM = 64;   // the register-staged 8-row form has no 128-row block
  }
  p.tiles_x = (d->ow + 15) / 16;
  p.tiles_y = (d->oh + p.TPH - 1) / p.TPH;
  p.tiles_total = p.tiles_x * p.tiles_y * d->n;
  p.nbm = d->cout_pad / p.BM;
  p.nbn = d->cin_pad / p.BN;
  // about one workgroup per CU (the partials cost HBM traffic); thin layers (3-channel side) are
  // bandwidth-bound with tiny partials, so they get four per CU to keep more loads in flight
  const int target = (p.BM == 16 || p.BN < 64) ? 1024 : 256;
  int want = (target + p.nbm * p.nbn - 1) / (p.nbm * p.nbn);
  if (const char* e = getenv("FSR_WGRAD_SLABS")) {   // tests: few slabs, so every workgroup walks several tiles
    const int v = atoi(e);
    if (v > 0) want = v;
  }
  if (want > p.tiles_total) want = p.tiles_total;
  if (want < 1) want = 1;
  p.tiles_per_slab = (p.tiles_total + want - 1) / want;
  p.nslab = (p.tiles_total + p.tiles_per_slab - 1) / p.tiles_per_slab;
  const int es = d->dtype != FSR_F32 ? 2 : 4;
  const int HH = (p.TPH - 1) * p.S + 3, HW = 15 * p.S + 3;
  p.lds = ((size_t)p.TPH * 16 * (p.BM + 16) + (size_t)HH * HW * (p.BN + 16)) * es;
  return 0;
}

template <typename T, int BM, int BN, int S, int TPH, bool X3W = false>
int launch_wgrad(const WgradKArgs& a, size_t lds, hipStream_t stream) {
  auto kern = conv_wgrad_kernel<T, BM, BN, S, TPH, X3W>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if constexpr (wgrad_dma<T, BM, BN, S, TPH>()) lds = wgrad_dma_lds_bytes<T, BM, BN, S, TPH>();
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.nslab * a.nbm * a.nbn * (a.group_n > 0 ? a.group_n : 1))), dim3((BM >= 64 && BN == 64) ? 512 : 256), lds, stream, a);
  return fsr_check_launch("conv_wgrad_kernel");
}

template <typename T, int TPH>
int dispatch_wgrad(const WgradPlan& p, const WgradKArgs& a, hipStream_t stream) {
  if constexpr (sizeof(T) == 2 && TPH == 8) {
    if (p.TPH == 4) {   // 16-bit stride 2, 64 input channels per block: LDS-DMA staging with 4-row tiles
      if (p.S != 2 || p.BN != 64) return fsr_fail(-2, "conv3x3_wgrad: 4-row tiles are a stride-2 configuration");
      if constexpr (std::is_same<T, bf16_t>::value) {
        if (p.BM == 128 && a.x3) return launch_wgrad<T, 128, 64, 2, 4, true>(a, p.lds, stream);
      }
      if (p.BM == 128) return launch_wgrad<T, 128, 64, 2, 4>(a, p.lds, stream);
      if (p.BM == 64) return launch_wgrad<T, 64, 64, 2, 4>(a, p.lds, stream);
      return fsr_fail(-2, "conv3x3_wgrad: no 4-row kernel for block %dx%d", p.BM, p.BN);
    }
  }
#define FSR_WG_CASE(bm, bn)                                                           \
  if (p.BM == bm && p.BN == bn)                                                       \
    return p.S == 1 ? launch_wgrad<T, bm, bn, 1, TPH>(a, p.lds, stream) : launch_wgrad<T, bm, bn, 2, TPH>(a, p.lds, stream);
  if constexpr (std::is_same<T, bf16_t>::value) {
    if (p.BM == 128 && p.BN == 64 && p.S == 1 && a.x3) return launch_wgrad<T, 128, 64, 1, TPH, true>(a, p.lds, stream);
  }
  if constexpr (sizeof(T) == 2) {
    if (p.BM == 128 && p.BN == 64 && p.S == 1) return launch_wgrad<T, 128, 64, 1, TPH>(a, p.lds, stream);
  }
  FSR_WG_CASE(64, 64)
  FSR_WG_CASE(64, 32)
  FSR_WG_CASE(16, 64)
  FSR_WG_CASE(16, 32)
  if constexpr (sizeof(T) == 4) {
    FSR_WG_CASE(64, 16)
    FSR_WG_CASE(16, 16)
  }
#undef FSR_WG_CASE
  return fsr_fail(-2, "conv3x3_wgrad: no kernel for block %dx%d", p.BM, p.BN);
}

}  // namespace

// Grouped form of the reduce: blockIdx.y = layer; layer l adds its nslab partial blocks into out.dw[l].
__global__ __launch_bounds__(256) void conv_wgrad_reduce_grouped_kernel(const float* __restrict__ ws, const WgradGroupOut out, int nslab,
                                                                        int cout, int cin, int cout_pad, int cin_pad) {
  __shared__ float red[8][32];
  const int total = 9 * cout * cin;
  const size_t sstride = (size_t)9 * cout_pad * cin_pad;
  const int oi = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const float* wl = ws + (size_t)blockIdx.y * nslab * sstride;
  float* dw = out.dw[0];
#pragma unroll
  for (int l = 1; l < WGRAD_GROUP_MAX; ++l)
    if (l == (int)blockIdx.y) dw = out.dw[l];
  for (int i0 = blockIdx.x * 32; i0 < total; i0 += gridDim.x * 32) {
    const int i = i0 + oi;
    float s = 0.f;
    int co = 0, ci = 0, t = 0;
    if (i < total) {
      ci = i % cin;
      co = (i / cin) % cout;
      t = i / (cin * cout);
      const float* p = wl + ((size_t)t * cout_pad + co) * cin_pad + ci;
      for (int k = pl; k < nslab; k += 8) s += p[(size_t)k * sstride];
    }
    red[pl][oi] = s;
    __syncthreads();
    if (pl == 0 && i < total) {
      float r = red[0][oi];
#pragma unroll
      for (int j = 1; j < 8; ++j) r += red[j][oi];
      dw[((size_t)co * cin + ci) * 9 + t] += r;
    }
    __syncthreads();
  }
}

// slabs per layer of a grouped launch: the layers share the ~256 workgroups of one ungrouped launch
static int group_slabs(const WgradPlan& p, int nlayers) {
  int wpl = 256 / nlayers;
  if (wpl < 1) wpl = 1;
  if (wpl > p.tiles_total) wpl = p.tiles_total;
  return wpl;
}

extern "C" size_t fsr_conv3x3_wgrad_grouped_workspace(const fsr_wgrad_desc* d, int nlayers) {
  WgradPlan p;
  if (nlayers < 1 || nlayers > WGRAD_GROUP_MAX || make_plan(d, p) != 0) return 0;
  return (size_t)nlayers * group_slabs(p, nlayers) * 9 * d->cout_pad * d->cin_pad * sizeof(float);
}

extern "C" int fsr_conv3x3_wgrad_grouped(const fsr_wgrad_desc* d, int nlayers, const void* const* x, const void* const* dy,
                                         float* const* dw_oihw, void* workspace, fsr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  WgradPlan p;
  if (int rc = make_plan(d, p)) return rc;
  if (nlayers < 1 || nlayers > WGRAD_GROUP_MAX) return fsr_fail(-2, "fsr_conv3x3_wgrad_grouped: 1..%d layers per launch", WGRAD_GROUP_MAX);
  if (!x || !dy || !dw_oihw || !workspace) return fsr_fail(-1, "fsr_conv3x3_wgrad_grouped: null argument");
  if (d->dtype == FSR_X3) return fsr_fail(-2, "fsr_conv3x3_wgrad_grouped: not available for x3 tensors (launch the layers one by one)");
  if (p.nbm * p.nbn != 1 || d->dy_pixel_shuffled || d->cout != d->cout_pad || d->cin != d->cin_pad)
    return fsr_fail(-2, "fsr_conv3x3_wgrad_grouped: layers of one 64 x 64 block (cout = cin = 64, unpadded, no pixel shuffle)");
  WgradKArgs a = {};
  WgradGroupOut out = {};
  for (int l = 0; l < nlayers; ++l) {
    if (!x[l] || !dy[l] || !dw_oihw[l]) return fsr_fail(-1, "fsr_conv3x3_wgrad_grouped: null pointer for layer %d", l);
    a.gx[l] = x[l];
    a.gdy[l] = dy[l];
    out.dw[l] = dw_oihw[l];
  }
  a.group_n = nlayers;
  a.ws = (float*)workspace;
  a.N = d->n; a.IH = d->ih; a.IW = d->iw; a.CinPad = d->cin_pad;
  a.OH = d->oh; a.OW = d->ow; a.CoutPad = d->cout_pad;
  a.dy_ps = 0;
  a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.tiles_total = p.tiles_total;
  a.nslab = group_slabs(p, nlayers);
  a.tiles_per_slab = (p.tiles_total + a.nslab - 1) / a.nslab;
  a.nslab = (p.tiles_total + a.tiles_per_slab - 1) / a.tiles_per_slab;
  a.nbm = a.nbn = 1;
  const int rc = d->dtype == FSR_BF16 ? dispatch_wgrad<bf16_t, 8>(p, a, stream)
                 : (d->dtype == FSR_F16 ? dispatch_wgrad<f16_t, 8>(p, a, stream) : dispatch_wgrad<float, 4>(p, a, stream));
  if (rc) return rc;
  const int total = 9 * d->cout * d->cin;
  int blocks = (total + 31) / 32;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(conv_wgrad_reduce_grouped_kernel, dim3(blocks, nlayers), dim3(256), 0, stream, (const float*)workspace, out, a.nslab,
                     d->cout, d->cin, d->cout_pad, d->cin_pad);
  return fsr_check_launch("conv_wgrad_reduce_grouped_kernel");
}

extern "C" size_t fsr_conv3x3_wgrad_workspace(const fsr_wgrad_desc* d) {
  WgradPlan p;
  if (make_plan(d, p) != 0) return 0;
  const int f = d->dtype == FSR_X3 ? 4 : 1;     // x3: all four (dy part, x part) product blocks
  return (size_t)p.nslab * 9 * d->cout_pad * d->cin_pad * f * sizeof(float);
}

extern "C" int fsr_conv3x3_wgrad(const fsr_wgrad_desc* d_in, const void* x, const void* dy, float* dw_oihw, void* workspace,
                                 fsr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  WgradPlan p;
  if (int rc = make_plan(d_in, p)) return rc;
  if (!x || !dy || !dw_oihw || !workspace) return fsr_fail(-1, "fsr_conv3x3_wgrad: null argument");
  const bool x3 = d_in->dtype == FSR_X3;
  if (x3 && ((((size_t)x | (size_t)dy) & 127) != 0)) return fsr_fail(-2, "fsr_conv3x3_wgrad: x3 tensors must be 128-byte aligned");
  const fsr_wgrad_desc dphys = physical_desc(d_in);
  const fsr_wgrad_desc* d = &dphys;
  WgradKArgs a = {};
  a.x = x;
  a.dy = dy;
  a.ws = (float*)workspace;
  a.N = d->n;
  a.IH = d->ih;
  a.IW = d->iw;
  a.CinPad = d->cin_pad;
  a.OH = d->oh;
  a.OW = d->ow;
  a.CoutPad = d->cout_pad;
  a.dy_ps = d->dy_pixel_shuffled;
  a.tiles_x = p.tiles_x;
  a.tiles_y = p.tiles_y;
  a.tiles_total = p.tiles_total;
  a.tiles_per_slab = p.tiles_per_slab;
  a.nslab = p.nslab;
  a.nbm = p.nbm;
  a.nbn = p.nbn;
  a.x3 = x3 ? 1 : 0;
  int rc = d->dtype == FSR_BF16 ? dispatch_wgrad<bf16_t, 8>(p, a, stream)
           : (d->dtype == FSR_F16 ? dispatch_wgrad<f16_t, 8>(p, a, stream) : dispatch_wgrad<float, 4>(p, a, stream));
  if (rc) return rc;
  const int total = 9 * d->cout * d->cin;
  int blocks = (total + 31) / 32;
  if (blocks > 8192) blocks = 8192;
  if (x3) {
    const int total3 = 9 * d_in->cout * d_in->cin;
    int blocks3 = (total3 + 31) / 32;
    if (blocks3 > 8192) blocks3 = 8192;
    hipLaunchKernelGGL(conv_wgrad_reduce_x3_kernel, dim3(blocks3), dim3(256), 0, stream, (const float*)workspace, dw_oihw,
                       p.nslab, d_in->cout, d_in->cin, d->cout_pad, d->cin_pad, d->dy_pixel_shuffled);
    return fsr_check_launch("conv_wgrad_reduce_x3_kernel");
  }
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)workspace, dw_oihw,
                     p.nslab, d->cout, d->cin, d->cout_pad, d->cin_pad, d->dy_pixel_shuffled);
  return fsr_check_launch("conv_wgrad_reduce_kernel");
}
