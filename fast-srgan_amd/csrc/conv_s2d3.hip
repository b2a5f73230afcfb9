// Data gradient of the stride-2 3x3 convolutions with 64..512 channels: ALL FOUR parity classes of a tile in one persistent
// workgroup, on v_mfma_f32_32x32x16_{bf16,f16} with both operands arriving by LDS-DMA (the machinery of conv_tall3.hip).
// (Round 4: replaces the four-class launches of conv_igemm.hip's <16,128,4,1,32,1,2,1> configuration, which re-read the
// gradient tensor once per class -- 1.96x its algorithmic bytes -- and ran at 0.16..0.24 of the MFMA peak.)
//
// Replaces the autograd data gradient of /root/reference/model.py:148-183, strides 2 at :150, :162, :172, :182 (Discriminator
// 64->64 @384^2 -- with the neck's LeakyReLU(0.2) backward as the mask --, 128->128 @192^2, 256->256 @96^2, 512->512 @48^2;
// trainer.py:181 and :195 run it at batch 2B and B).  The 64->64 layer ran on a filter-resident 8-wave kernel with 8-byte
// stores before (conv64_s2dgrad_kernel, rounds 1-3: 225 / 429 us at batch 32 / 64 against 164 / 382 here).
//
// The arithmetic.  Forward: y[oy, ox] = sum_{ky,kx} x[2 oy + ky - 1, 2 ox + kx - 1] W[ky, kx].  The gradient of input pixel
// (2a + py, 2b + px) collects the taps whose parity matches -- ky = 1 for py = 0, ky in {0, 2} for py = 1 -- and tap (ky, kx)
// reads dy[a + (ky == 0), b + (kx == 0)]:
//   class (0,0): tap (1,1)                          class (0,1): taps (1,0) (1,2)
//   class (1,0): taps (0,1) (2,1)                   class (1,1): taps (0,0) (0,2) (2,0) (2,2)
// Every tap belongs to exactly ONE class, so one pass over the nine filter slices with four accumulator sets computes the
// 2 x 2 block of dx under every dy pixel with no wasted MAC, and dy is read ONCE (a (TH+1) x 17 halo per tile).
//
// Work decomposition
//   tile      8 x 16 dy pixels of one image (= 16 x 32 dx pixels) x 64 dx channels; four waves = two dy-row groups (4 rows:
//             two pixel fragments of 2 rows x 16) x two channel groups (32: one filter fragment); a workgroup walks tiles
//             persistently (tile += gridDim), two workgroups per CU
//   wave      4 classes x 2 pixel fragments x 16 accumulator registers = 128; per 32-channel chunk of dy and k half: 9 filter
//             fragments + the pixel fragments of the four halo offsets, 18 MFMAs
//   K loop    chunk of 32 dy channels x three STAGES of three taps, ordered so that taps sharing a halo offset share a stage:
//               stage 0: (1,1) (1,2) (2,1)   all at offset (0,0)
//               stage 1: (2,2) at (0,0); (1,0) (2,0) at (0,1)
//               stage 2: (0,1) (0,2) at (1,0); (0,0) at (1,1)
//             a substep = one k half (16 channels) of a stage: 6 MFMAs fed by 3 filter + 2 or 4 pixel fragment reads (38 reads
//             per 36 MFMAs; conv_tall3's 64-channel-block form runs 45 per 36).  Reads run one substep ahead.
// LDS (bytes)  halo[2][12 KB]  9 x 18 pixels (17 used) x 32 channels of chunk c / c+1, 64 B per pixel, swizzled like conv_tall3's
//              ring[4][3 taps][64 rows][64 B] = 48 KB, stage[4 waves][2 KB] (epilogue)    -> 80 KB: two workgroups per CU
//   The DMA runs ONE CHUNK ahead: stage (c, s) issues the filter pieces of stage (c+1, s) into the slot stage (c, s-1) just left,
//   stage (c, 0) also the three halo pieces of chunk c+1.  One s_barrier per stage (before its second substep) with a counted
//   vmcnt in front of it, as in conv_tall3.hip.
// Epilogue: a lane holds 16 consecutive channels of one dy pixel per class and fragment; each fragment is transposed through a
//   wave-private LDS buffer so that a store instruction writes 16 dx pixels (2a + py, 2b + px) x 64 contiguous bytes; the
//   optional mask (the saved forward input of the producing layer: LeakyReLU / ReLU backward) is loaded the same way.
#include "fsr_common.h"
#include "fsr_conv_args.h"
#include "fsr_host.h"

#include <stdlib.h>

// Ablation builds (tools/build_variant1.sh -DFSR_ABLS=<mask>; results WRONG on purpose, the product library is built with 0):
//   1 no stores   2 no DMA after the prologue   8 no fragment reads   32 no barriers   64 no DMA waits   128 no MFMAs
#ifndef FSR_ABLS
#define FSR_ABLS 0
#endif

namespace {

constexpr int S2D_ABL = FSR_ABLS;

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

// the forward tap (ky * 3 + kx) at position p of a chunk's nine taps (three stages of three)
constexpr int S2D_TAP[9] = {4, 5, 7, 8, 3, 6, 1, 2, 0};
constexpr int s2d_cls(int t) { return ((t / 3) != 1 ? 2 : 0) + ((t % 3) != 1 ? 1 : 0); }   // parity class py * 2 + px of tap t
constexpr int s2d_oy(int t) { return (t / 3) == 0 ? 1 : 0; }                               // halo offset of tap t
constexpr int s2d_ox(int t) { return (t % 3) == 0 ? 1 : 0; }
// a stage reads the pixel fragments of at most TWO halo offsets; which of the two a tap uses (0: the stage's first tap's)
constexpr int s2d_set(int pos) {
  const int t = S2D_TAP[pos], t0 = S2D_TAP[pos / 3 * 3];
  return (s2d_oy(t) == s2d_oy(t0) && s2d_ox(t) == s2d_ox(t0)) ? 0 : 1;
}
constexpr bool s2d_two_sets(int stage) { return s2d_set(stage * 3 + 1) == 1 || s2d_set(stage * 3 + 2) == 1; }
// halo offset of set `set` of a stage (set 1 = the first tap of the stage that differs from tap 0's offset)
constexpr int s2d_set_tap(int stage, int set) {
  if (set == 0) return S2D_TAP[stage * 3];
  return s2d_set(stage * 3 + 1) == 1 ? S2D_TAP[stage * 3 + 1] : S2D_TAP[stage * 3 + 2];
}

constexpr int S2D_BN = 64, S2D_NW = 4, S2D_TH = 8, S2D_MB = 2;
constexpr int S2D_P = 18, S2D_ROWB = S2D_P * 64;
constexpr int S2D_HUNITS = (S2D_TH + 1) * S2D_P * 4;           // 16-byte units of one halo chunk (648)
constexpr int S2D_HPW = 3;                                     // halo pieces per wave and chunk (12 pieces, 11 used)
constexpr int S2D_HALO_BYTES = S2D_HPW * S2D_NW * 1024;
constexpr int S2D_SLOT_BYTES = 3 * S2D_BN * 64;                // three taps x 64 rows x 64 B
constexpr int S2D_NSLOT = 4;
constexpr int S2D_STAGE_OFF = 2 * S2D_HALO_BYTES + S2D_NSLOT * S2D_SLOT_BYTES;   // the epilogue's transposing buffers: 2 KB per wave
constexpr int S2D_LDS = S2D_STAGE_OFF + S2D_NW * 2048;
static_assert(S2D_HUNITS <= S2D_HPW * S2D_NW * 64, "the halo fits its pieces");

__device__ __forceinline__ int s2d_swz_row(int R) { return (R >> 2) & 3; }
__device__ __forceinline__ int s2d_swz_col(int x) { return (x >> 1) & 3; }

template <typename V>
__device__ __forceinline__ V s2d_lds_read(const char* smem, unsigned off) {
  return *FSR_LDS_PTR(const V, smem + off);
}

// X3 (FSR_X3, T = bf16_t): dy is an x3 tensor seen as a bf16 tensor of a.Cin = 2 x logical channels (hi / lo chunks of 32), the
// filter pack alternates w_hi / w_lo chunks; the kernel walks three virtual chunks per channel group -- (dy_hi, w_hi), (dy_lo, w_hi),
// (dy_hi, w_lo) -- through the unchanged pipeline (conv_tall3.hip's scheme: only the DMA source offsets are mapped) and the
// epilogue sends the hi and the lo parts of a fragment through the transposing buffer one after the other (64 + 64 bytes per
// pixel and 32-channel group).
__device__ __forceinline__ int s2d_div3(int j) { return (int)(((unsigned)j * 0xAAABu) >> 17); }
template <bool X3> __device__ __forceinline__ int s2d_hmap(int j) {
  if constexpr (!X3) return j;
  const int g = s2d_div3(j), r = j - 3 * g;
  return __builtin_amdgcn_readfirstlane(2 * g + (r == 1 ? 1 : 0));
}
template <bool X3> __device__ __forceinline__ int s2d_fmap(int j) {
  if constexpr (!X3) return j;
  const int g = s2d_div3(j), r = j - 3 * g;
  return __builtin_amdgcn_readfirstlane(2 * g + (r == 2 ? 1 : 0));
}

template <typename T, bool X3 = false>
__global__ __launch_bounds__(S2D_NW * 64, 2) void conv_s2d3_kernel(const ConvKArgs a) {
  static_assert(!X3 || std::is_same<T, bf16_t>::value, "x3: bf16 planes");
  typedef typename std::conditional<X3, x3_t, T>::type ST;
  constexpr int BN = S2D_BN, NW = S2D_NW, TH = S2D_TH, MB = S2D_MB, HPW = S2D_HPW;
  HIP_DYNAMIC_SHARED(char, smem)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wpx = wave >> 1, wco = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5, l15 = lane & 15, lrow = (lane >> 4) & 1;

  const fsr_lds_addr_t halo_addr = FSR_LDS_ADDR(smem);
  const fsr_lds_addr_t ring_addr = halo_addr + 2 * S2D_HALO_BYTES;
  const fsr_buf_t in_buf = fsr_make_buf(a.in, (unsigned)((size_t)a.N * a.IH * a.IW * a.Cin * sizeof(T)));
  const fsr_buf_t w_buf = fsr_make_buf(a.wpk, (unsigned)((size_t)9 * a.CoutPad * a.Cin * sizeof(T)));
  const int nchunks = X3 ? 3 * (a.Cin >> 6) : a.Cin >> 5;       // x3: three virtual chunks per (hi, lo) pair of physical ones
  const unsigned wcs = a.wlin ? (unsigned)(9 * BN * 64) : 64u;      // byte step of the filter source from chunk to chunk

  // ---- loop-invariant per-lane addresses (the layouts of conv_tall3.hip) ------------------------------------------------
  // filter fragment of the tap at position g of the stage in ring slot `sl`, k half j: ring + sl + g * BN * 64 + aoff[j]
  unsigned aoff[2];
  {
    const int R = wco * 32 + l31;
#pragma unroll
    for (int j = 0; j < 2; ++j) aoff[j] = (unsigned)(2 * S2D_HALO_BYTES + R * 64 + (((2 * j + hi) ^ s2d_swz_row(R)) << 4));
  }
  // pixel fragment m at halo offset (oy, ox), k half j, halo buffer hb: hb + (2 m + oy) * ROWB + boff[ox][j]
  unsigned boff[2][2];
#pragma unroll
  for (int ox = 0; ox < 2; ++ox)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      boff[ox][j] = (unsigned)(((wpx * 2 * MB + lrow) * S2D_P + l15 + ox) * 64 + (((2 * j + hi) ^ s2d_swz_col(l15 + ox)) << 4));
  // DMA source of this wave's filter piece of a tap (rows 16 * wave .. + 15 of the 64-row block): LDS row i of a 32-row block
  // holds dx channel 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3), so a lane's 16 accumulator registers are 16 consecutive channels
  unsigned wvoff;
  {
    const int R = wave * 16 + (lane >> 2), ul = lane & 3, i = R & 31;
    const int co = (R & ~31) + 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3);
    wvoff = a.wlin ? (unsigned)(wave * 1024 + lane * 16)       // stage-contiguous pack: a piece = 1 KB of contiguous memory
                   : (unsigned)((co * a.Cin + ((ul ^ s2d_swz_row(R)) << 3)) * (int)sizeof(T));
  }

  f32x16 acc[4][MB];
  s16x8 fa[2][3], fb[2][2][MB];     // [register set][tap of the stage] / [register set][offset set of the stage][fragment]

  struct TileC { int img, gy0, gx0, nb; };
  TileC cur = {0, 0, 0, 0}, nxt = {0, 0, 0, 0};
  unsigned hv_cur[HPW], hv_nxt[HPW];
  unsigned ws_cur = 0, ws_nxt = 0;
  auto setup = [&](int tile, TileC& tc, unsigned (&hv)[HPW], unsigned& ws) {
    int L = tile;
    tc.nb = L % a.nblk_n; L /= a.nblk_n;
    const int tx = L % a.tiles_x; L /= a.tiles_x;
    const int ty = L % a.tiles_y;
    tc.img = L / a.tiles_y;
    tc.gy0 = ty * TH;
    tc.gx0 = tx * 16;
    ws = a.wlin ? (unsigned)(tc.nb * (a.Cin >> 5) * 9 * BN * 64) : (unsigned)(tc.nb * BN * a.Cin * (int)sizeof(T));
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
      const int U = (wave + k * NW) * 64 + lane;
      const int hp = U >> 2, ul = U & 3;
      const int hy = hp / S2D_P, hx = hp - hy * S2D_P;
      const int iy = tc.gy0 + hy, ix = tc.gx0 + hx;
      unsigned o = ~0u;                                          // beyond the buffer: the DMA writes zeros
      if (U < S2D_HUNITS && hx <= 16 && iy < a.IH && ix < a.IW)
        o = (unsigned)((((tc.img * a.IH + iy) * a.IW + ix) * a.Cin + ((ul ^ s2d_swz_col(hx)) << 3)) * (int)sizeof(T));
      hv[k] = o;
    }
  };
  auto dma_halo = [&](const unsigned (&hv)[HPW], int c, int k, unsigned hb) {
    if constexpr (!(S2D_ABL & 2))
    FSR_BLDS16(in_buf, hv[k], (unsigned)(s2d_hmap<X3>(c) * 64), halo_addr + (fsr_lds_addr_t)(hb + (wave + k * NW) * 1024));
  };
  // this wave's piece of the tap at position `pos` of chunk c's nine (slice a.t3_woff[forward tap]) into slot offset `dst`
  auto dma_filter = [&](unsigned ws, int c, unsigned woff_tap, unsigned dst) {
    if constexpr (!(S2D_ABL & 2))
    FSR_BLDS16(w_buf, wvoff, woff_tap + ws + (unsigned)s2d_fmap<X3>(c) * wcs, ring_addr + (fsr_lds_addr_t)(dst + wave * 1024));
  };
  auto slot_of = [&](int gs) { return (unsigned)((gs & (S2D_NSLOT - 1)) * S2D_SLOT_BYTES); };

  // fragment r of substep (stage si, k half j) in need order: a0, b[0][0..], a1, (b[1][..] | a2), ... into register set `buf`
  //   stage 0 (one offset set):   a0 b00 b01 a1 a2
  //   stage 1 (tap 0 | taps 1,2): a0 b00 b01 a1 b10 b11 a2
  //   stage 2 (taps 0,1 | tap 2): a0 b00 b01 a1 a2 b10 b11
  auto read_frag = [&](auto rc, auto bufc, auto sic, auto jc, unsigned sl, unsigned hb) {
    constexpr int r = decltype(rc)::value, buf = decltype(bufc)::value, si = decltype(sic)::value, j = decltype(jc)::value;
    constexpr bool two = s2d_two_sets(si);
    constexpr int NRD = two ? 7 : 5;
    if constexpr (r < NRD && !(S2D_ABL & 8)) {
      // decode r -> (kind, index)
      constexpr int seq1[7] = {0, 10, 11, 1, 20, 21, 2};      // stage 1 order: a = 0..2, b set 0 = 10 + m, b set 1 = 20 + m
      constexpr int seq2[7] = {0, 10, 11, 1, 2, 20, 21};      // stage 2 order
      constexpr int seq0[5] = {0, 10, 11, 1, 2};
      constexpr int code = !two ? seq0[r] : (s2d_set(si * 3 + 1) == 1 ? seq1[r] : seq2[r]);
      if constexpr (code < 10) {
        fa[buf][code] = s2d_lds_read<s16x8>(smem, aoff[j] + sl + (unsigned)(code * BN * 64));
      } else {
        constexpr int set = code / 10 - 1, m = code % 10;
        constexpr int t = s2d_set_tap(si, set);
        fb[buf][set][m] = s2d_lds_read<s16x8>(smem, boff[s2d_ox(t)][j] + hb + (unsigned)((2 * m + s2d_oy(t)) * S2D_ROWB));
      }
    }
  };

  const int nround = (int)gridDim.x;
  int tile = (int)blockIdx.x;
  auto logical = [&](int t) {
    const int r0 = (t / nround) * nround;
    const int cnt = a.t3_ntiles - r0 < nround ? a.t3_ntiles - r0 : nround;
    return r0 + xcd_remap(t - r0, cnt);
  };
  if (tile >= a.t3_ntiles) return;

  // ---- prologue: halo chunk 0, the three stages of chunk 0 ---------------------------------------------------------------
  setup(logical(tile), cur, hv_cur, ws_cur);
#pragma unroll
  for (int k = 0; k < HPW; ++k) dma_halo(hv_cur, 0, k, 0u);
  static_for<0, 9>([&](auto pc) {
    constexpr int pos = decltype(pc)::value;
    dma_filter(ws_cur, 0, a.t3_woff[S2D_TAP[pos]], (unsigned)((pos / 3) * S2D_SLOT_BYTES + (pos % 3) * BN * 64));
  });
  int next = tile + nround;
  bool has_nxt = next < a.t3_ntiles;
  FSR_WAIT_VM(0);
  FSR_BARRIER();
  auto acc_zero = [&]() {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][m][e] = 0.f;
  };
  acc_zero();
  static_for<0, 5>([&](auto rc) {   // fragments of substep (stage 0, k half 0)
    read_frag(rc, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, slot_of(0), 0u);
  });
  int gs = 0;                       // global stage counter: stage (chunk, si) lives in ring slot (gs + si) % 4
  unsigned hb = 0;                  // halo buffer (byte offset) of the current chunk
  bool full_prev = true;            // did the preceding stage issue its complete set of pieces (counted waits)

  for (;;) {
    for (int c = 0; c < nchunks; ++c, gs += 3, hb ^= (unsigned)S2D_HALO_BYTES) {
      const bool last = c + 1 == nchunks;
      if (last && has_nxt) setup(logical(next), nxt, hv_nxt, ws_nxt);     // from here on the DMA feeds the next tile
      const bool issue = !last || has_nxt;
      const int cD = last ? 0 : c + 1;
      const unsigned wsD = last ? ws_nxt : ws_cur;
      const unsigned hbN = hb ^ (unsigned)S2D_HALO_BYTES;
      static_for<0, 3>([&](auto sic) {
        constexpr int si = decltype(sic)::value;
        const unsigned sl = slot_of(gs + si);
        const unsigned slD = slot_of(gs + 3 + si);                  // = the slot stage (c, si - 1) has just left
        constexpr int siN = (si + 1) % 3;
        const bool has_next_stage = si < 2 || issue;
        const unsigned slN = slot_of(gs + si + 1);
        const unsigned hbNext = si < 2 ? hb : hbN;
        static_for<0, 2>([&](auto qc) {
          constexpr int q = decltype(qc)::value;       // = the k half
          constexpr int buf = (si * 2 + q) & 1;
          if constexpr (q == 1) {
            // publish stage s+1: its pieces were issued a chunk (three stages) ago; at most the pieces of the previous and of
            // this stage may still be in flight (6 in a chunk's first stage -- three halo pieces -- else 3).  Loads retire in
            // order, stores in the queue only lengthen the wait.  A stage that issued nothing (the end of the last tile)
            // makes the count meaningless: wait for everything.
            constexpr int n_this = si == 0 ? 6 : 3, n_prev = si == 1 ? 6 : 3;
            if constexpr (!(S2D_ABL & 64)) {
              if (full_prev && issue) FSR_WAIT_VM(n_this + n_prev);
              else FSR_WAIT_VM(0);
            }
            if constexpr (!(S2D_ABL & 32)) FSR_BARRIER();
          }
          static_for<0, 6>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int g = i / MB, m = i % MB;
            constexpr int pos = si * 3 + g, t = S2D_TAP[pos];
            if constexpr (!(S2D_ABL & 128)) acc[s2d_cls(t)][m] = Mfma32<T>::run(fa[buf][g], fb[buf][s2d_set(pos)][m], acc[s2d_cls(t)][m]);
            auto next_frag = [&](auto rc) {
              if constexpr (q == 0) {
                read_frag(rc, std::integral_constant<int, buf ^ 1>{}, sic, std::integral_constant<int, 1>{}, sl, hb);
              } else {
                if (has_next_stage)
                  read_frag(rc, std::integral_constant<int, buf ^ 1>{}, std::integral_constant<int, siN>{}, std::integral_constant<int, 0>{}, slN, hbNext);
              }
            };
            next_frag(ic);
            if constexpr (i == 0) next_frag(std::integral_constant<int, 6>{});     // the seventh read of a two-offset substep
            // DMA pieces of chunk c+1 (or of the next tile's chunk 0): first substep of a stage, one piece per MFMA slot
            if constexpr (q == 0) {
              if constexpr (si == 0 && i < 3) {
                if (issue) dma_halo(last ? hv_nxt : hv_cur, cD, i, hbN);
              }
              if constexpr (i >= 3) {
                if (issue) dma_filter(wsD, cD, a.t3_woff[S2D_TAP[si * 3 + (i - 3)]], slD + (unsigned)((i - 3) * BN * 64));
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          });
        });
        full_prev = issue;
      });
    }

    // ---- epilogue: dx pixel (2a + py, 2b + px) of every class ------------------------------------------------------------
    // After the MFMAs a lane holds 16 consecutive channels (32 B) of ONE dy pixel per class and fragment; stored from there a
    // wave instruction is 64 separate 16-byte pieces 512+ B apart, and the CU's vector-memory path -- shared with the other
    // workgroup's DMA pieces -- takes them one segment at a time (ablation, profiles/r04_s2d3_ablation.txt: without stores the
    // kernel ran 2.0..2.5x faster).  Each fragment (32 pixels x 32 channels = 2 KB) is therefore TRANSPOSED through a wave-private
    // 2 KB LDS buffer: written in accumulator layout, read back as four lanes per pixel, so a store instruction covers 16
    // pixels x 64 contiguous bytes (16 segments instead of 64).  The mask travels the other way (coalesced load, transposed
    // into accumulator layout) so that the gate is applied to the f32 accumulators exactly as before.
    ST* outp = (ST*)a.out;
    const ST* maskp = (const ST*)a.dmask;
    char* stg = smem + S2D_STAGE_OFF + wave * 2048;
    const int ct = lane >> 2, qt = lane & 3;                               // store layout: pixel column, 16-byte piece
    // Two swizzles (MI355X_MICROARCH.md, LDS: ds_write_b128 is served in groups of 8 contiguous lanes over 32 banks, ds_read_b128 in four
    // fixed 16-lane groups over 64): the OUTPUT is written in accumulator layout -- pieces 2 hi, 2 hi + 1 of pixel l31 -- so its units are
    // swizzled by (pixel >> 1) & 3 (the four same-parity pixels of a write group land on four different bank quads); the MASK is READ in
    // accumulator layout, units swizzled by (pixel >> 2) & 3 (conv_tall3's: the lanes {p, p + 12, p + 20, p + 24} of a read group differ).
    // The store-layout side (four lanes per pixel, 64 contiguous bytes) is conflict-free under either.  PMC: 12.7 % conflict cycles with
    // one swizzle for both (profiles/r04_pmc_sq_stride2.txt).
    const int fw = (l31 >> 1) & 3, fr = (l31 >> 2) & 3, fwt = (ct >> 1) & 3, frt = (ct >> 2) & 3;
    const unsigned sa0 = (unsigned)(l31 * 64 + (((2 * hi) ^ fw) << 4)), sa1 = (unsigned)(l31 * 64 + (((2 * hi + 1) ^ fw) << 4));   // output, accumulator layout
    const unsigned sb0 = (unsigned)(ct * 64 + ((qt ^ fwt) << 4)), sb1 = sb0 + 16 * 64;       // output, store layout: row 0 / row 1 of the fragment
    const unsigned ma0 = (unsigned)(l31 * 64 + (((2 * hi) ^ fr) << 4)), ma1 = (unsigned)(l31 * 64 + (((2 * hi + 1) ^ fr) << 4));   // mask, accumulator layout
    const unsigned mb0 = (unsigned)(ct * 64 + ((qt ^ frt) << 4)), mb1 = mb0 + 16 * 64;       // mask, load layout
    const int gb = cur.gx0 + ct;
    const int cob = cur.nb * BN + wco * 32 + qt * 8;
    // sign-bit mask: all eight words of the lane (class x fragment) are requested BEFORE the first store -- a load issued after a
    // store waits for that store (one counter), and loaded one by one inside the loop each would be a serial memory round trip
    unsigned sbits[4][MB];
    if (maskp && a.dmask_bits) {
      static_for<0, 4>([&](auto kc) {
        constexpr int k = decltype(kc)::value, py = k >> 1, px = k & 1;
        static_for<0, MB>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          const int ga = cur.gy0 + wpx * 2 * MB + 2 * m + lrow, gba = cur.gx0 + l15;
          const int oya = 2 * ga + py, oxa = 2 * gba + px;
          const bool oka = ga < a.IH && gba < a.IW && oya < a.FOH && oxa < a.FOW;
          const size_t bo = oka ? ((size_t)((cur.img * a.FOH + oya) * a.FOW + oxa) * (unsigned)(a.Cout >> 3) + (unsigned)((cur.nb * BN + wco * 32 + hi * 16) >> 3)) : 0;
          sbits[k][m] = *(const unsigned short*)((const unsigned char*)a.dmask + bo);     // (an out-of-range lane reads word 0: its result is never stored)
        });
      });
    }
    static_for<0, 4>([&](auto kc) {
      constexpr int k = decltype(kc)::value, py = k >> 1, px = k & 1;
      static_for<0, MB>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const int ga0 = cur.gy0 + wpx * 2 * MB + 2 * m;
        const int ox = 2 * gb + px;
        const bool okx = gb < a.IW && ox < a.FOW;
        const bool ok0 = okx && ga0 < a.IH && 2 * ga0 + py < a.FOH, ok1 = okx && ga0 + 1 < a.IH && 2 * ga0 + 2 + py < a.FOH;
        const unsigned off0 = (unsigned)((cur.img * a.FOH + 2 * ga0 + py) * a.FOW + ox) * (unsigned)a.Cout + (unsigned)cob;
        const unsigned off1 = off0 + 2u * (unsigned)a.FOW * (unsigned)a.Cout;
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = acc[k][m][e];
        if (maskp && a.dmask_bits) {
          // the producing layer's activation backward from its SIGN BITS ([N][FOH][FOW][Cout / 8], written by conv_c3_fwd_kernel): two
          // bytes per lane in accumulator layout instead of 32 -- the 64-channel 384^2 tensor of the discriminator's neck is 1.2 GB at
          // batch 64, its sign bits 75 MB
          const unsigned bits = sbits[k][m];
          const float ms = a.dmask_slope;
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = ((bits >> e) & 1u) ? v[e] : v[e] * ms;
        } else if (maskp) {   // fused activation backward of the producing layer: dz = dx * act'(y), y = its saved output
          u32x4 m0 = {0u, 0u, 0u, 0u}, m1 = {0u, 0u, 0u, 0u};
          // (x3: the hi parts carry the sign)
          if (ok0) m0 = *(const u32x4*)(X3 ? (const void*)x3_hi_ptr(maskp + off0) : (const void*)(maskp + off0));
          if (ok1) m1 = *(const u32x4*)(X3 ? (const void*)x3_hi_ptr(maskp + off1) : (const void*)(maskp + off1));
          *FSR_LDS_PTR(u32x4, stg + mb0) = m0;
          *FSR_LDS_PTR(u32x4, stg + mb1) = m1;
          FSR_WAVE_SYNC();
          const u32x4 k0 = *FSR_LDS_PTR(const u32x4, stg + ma0), k1 = *FSR_LDS_PTR(const u32x4, stg + ma1);
          FSR_WAVE_SYNC();
          const float ms = a.dmask_slope;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] = (int)(k0[e] << 16) > 0 ? v[2 * e] : v[2 * e] * ms;
            v[2 * e + 1] = (int)(k0[e] & 0xffff0000u) > 0 ? v[2 * e + 1] : v[2 * e + 1] * ms;
            v[8 + 2 * e] = (int)(k1[e] << 16) > 0 ? v[8 + 2 * e] : v[8 + 2 * e] * ms;
            v[8 + 2 * e + 1] = (int)(k1[e] & 0xffff0000u) > 0 ? v[8 + 2 * e + 1] : v[8 + 2 * e + 1] * ms;
          }
        }
        if constexpr (X3) {
          u32x4 h0, h1, l0, l1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            unsigned hh, ll;
            x3_split2(v[2 * e], v[2 * e + 1], hh, ll);
            h0[e] = hh; l0[e] = ll;
            x3_split2(v[8 + 2 * e], v[8 + 2 * e + 1], hh, ll);
            h1[e] = hh; l1[e] = ll;
          }
          char* d0 = (char*)x3_hi_ptr(outp + off0);
          char* d1 = (char*)x3_hi_ptr(outp + off1);
          *FSR_LDS_PTR(u32x4, stg + sa0) = h0;
          *FSR_LDS_PTR(u32x4, stg + sa1) = h1;
          FSR_WAVE_SYNC();
          const u32x4 o0 = *FSR_LDS_PTR(const u32x4, stg + sb0), o1 = *FSR_LDS_PTR(const u32x4, stg + sb1);
          FSR_WAVE_SYNC();
          *FSR_LDS_PTR(u32x4, stg + sa0) = l0;
          *FSR_LDS_PTR(u32x4, stg + sa1) = l1;
          FSR_WAVE_SYNC();
          const u32x4 q0 = *FSR_LDS_PTR(const u32x4, stg + sb0), q1 = *FSR_LDS_PTR(const u32x4, stg + sb1);
          FSR_WAVE_SYNC();
          if (!(S2D_ABL & 1)) {
            if (ok0) {
              fsr_st<16>((u32x4*)d0, o0);
              fsr_st<16>((u32x4*)(d0 + 64), q0);
            }
            if (ok1) {
              fsr_st<16>((u32x4*)d1, o1);
              fsr_st<16>((u32x4*)(d1 + 64), q1);
            }
          }
        } else {
        u32x4 p0, p1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          p0[e] = pack2<T>(v[2 * e], v[2 * e + 1]);
          p1[e] = pack2<T>(v[8 + 2 * e], v[8 + 2 * e + 1]);
        }
        *FSR_LDS_PTR(u32x4, stg + sa0) = p0;
        *FSR_LDS_PTR(u32x4, stg + sa1) = p1;
        FSR_WAVE_SYNC();
        const u32x4 o0 = *FSR_LDS_PTR(const u32x4, stg + sb0), o1 = *FSR_LDS_PTR(const u32x4, stg + sb1);
        FSR_WAVE_SYNC();
        if (!(S2D_ABL & 1)) {
          if (ok0) fsr_st<16>((u32x4*)((T*)outp + off0), o0);
          if (ok1) fsr_st<16>((u32x4*)((T*)outp + off1), o1);
        }
        }
      });
    });
    if (!has_nxt) break;
    cur = nxt;
#pragma unroll
    for (int k = 0; k < HPW; ++k) hv_cur[k] = hv_nxt[k];
    ws_cur = ws_nxt;
    tile = next;
    next = tile + nround;
    has_nxt = next < a.t3_ntiles;
    acc_zero();
  }
}

int s2d_cus() {
  // FSR_PERSIST_CUS=<n> (tests): number of CUs the persistent walk is sized for.  Read per launch (the tests change it).
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

template <typename T, bool X3 = false>
int s2d_launch(ConvKArgs& a, hipStream_t stream) {
  auto kern = conv_s2d3_kernel<T, X3>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, S2D_LDS);
    attr_set = true;
  }
  long long grid = (long long)s2d_cus() * 2;
  if (grid > a.t3_ntiles) grid = a.t3_ntiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(S2D_NW * 64), S2D_LDS, stream, a);
  fsr_note_kernel("conv_s2d3_kernel<%s>", X3 ? "x3" : (std::is_same<T, f16_t>::value ? "f16" : "bf16"));
  const int rc = fsr_check_launch("conv_s2d3_kernel");
  return rc ? rc : 1;
}

}  // namespace

// Stride-2 data gradient, 128..512 channels, all parity classes per tile: 1 = launched, 0 = not this kernel's shape (the
// caller falls through to the class launches of conv_igemm.hip), < 0 = error.  `a` as fsr_api.hip prepares it for a data
// gradient: in = dy (N, IH, IW, Cin), out = dx (N, FOH, FOW, Cout), wpk = the transposed pack [9][Cout][Cin] in forward tap order.
int fsr_conv_s2d3_try(int dtype, ConvKArgs& a, hipStream_t stream) {
  static const bool off = getenv("FSR_S2D3") && atoi(getenv("FSR_S2D3")) == 0;   // A/B switch
  if (off || (dtype != FSR_BF16 && dtype != FSR_F16 && dtype != FSR_X3)) return 0;
  if (dtype == FSR_X3 && ((a.Cin & 63) != 0 || (a.Cin >> 6) * 3 >= (1 << 15) || a.dmask_bits)) return 0;   // a.Cin: physical channels
  if (a.Cin < 64 || a.Cin % 32 != 0 || a.Cout % 64 != 0 || a.Cout < 64 || a.CoutPad != a.Cout) return 0;
  if (a.ps || a.in_ps || a.out_f32 || a.preact || a.oscale || a.bias || a.stats || a.pool2 || a.act != FSR_ACT_NONE || a.dmask_add) return 0;
  if (a.IH != (a.FOH - 1) / 2 + 1 || a.IW != (a.FOW - 1) / 2 + 1) return 0;
  if ((long long)a.N * a.IH * a.IW * a.Cin >= (1LL << 31) || (long long)a.N * a.FOH * a.FOW * a.Cout >= (1LL << 31)) return 0;
  ConvKArgs b = a;
  b.tiles_x = (b.IW + 15) / 16;
  b.tiles_y = (b.IH + S2D_TH - 1) / S2D_TH;
  b.nblk_n = b.Cout / S2D_BN;
  const long long ntiles = (long long)b.tiles_x * b.tiles_y * b.N * b.nblk_n;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) return 0;
  b.t3_ntiles = (int)ntiles;
  // stage-contiguous filter pack (fsr_pack_conv3x3_lin, 64-channel blocks) for 128 and more gradient channels: -2..11 % per layer;
  // with 64 (a row of the standard pack is one whole line already) it measured +8 %, so that layer keeps the standard pack
  const int want = b.Cin >= 128 ? S2D_BN : 0;
  if (b.query) {
    a.wlin_want = want;
    return 1;
  }
  if (b.wlin != 0 && b.wlin != S2D_BN)
    return fsr_fail(-2, "conv_s2d3: the filter pack is stage-contiguous in blocks of %d channels, this launch needs %d", b.wlin, S2D_BN);
  for (int t = 0; t < 9; ++t) b.t3_woff[t] = b.wlin ? (unsigned)(t * S2D_BN * 64) : (unsigned)((size_t)t * b.CoutPad * b.Cin * 2);
  const int rc = dtype == FSR_X3 ? s2d_launch<bf16_t, true>(b, stream) : (dtype == FSR_F16 ? s2d_launch<f16_t>(b, stream) : s2d_launch<bf16_t>(b, stream));
  if (rc == 1) a = b;
  return rc;
}
