// First-layer 3x3 convolutions (3 input channels), reading the float image directly: forward and weight gradient.
//
//   Generator.neck      /root/reference/model.py:75-78   Conv2d(3, 64, k3, p1) + PReLU
//   Discriminator.neck  /root/reference/model.py:143-146 Conv2d(3, 64, k3, p1) + LeakyReLU(0.2)
//   vgg19.features.0    /root/reference/model.py:8,20-22 (x+1)/2, (x-mean)/std, Conv2d(3, 64) + ReLU
//
// K = 27 is too thin for the implicit-GEMM kernel's (tap x 32-channel) steps: padding the image to 32 channels
// quadruples its traffic and spends 9 MFMA steps on 91 % zeros.  Here the whole 3x3x3 patch is ONE K = 32 step
// (27 real + 5 zero): the workgroup stages an 18x18x3 float patch of the (optionally normalised) image in LDS,
// each lane gathers its 8 patch values per pixel row with scalar LDS reads, and one MFMA per 16 output channels
// finishes a 16-pixel row.  The kernels are bound by writing / reading the 64-channel tensor (HBM).
//
// The weight gradient is the transposed problem: dW[co][k] = sum_pixels dz[pixel][co] * patch[pixel][k], K = pixels,
// with dz transposed on the way from LDS (ds_read_b64_tr_b16 / scalar reads in f32 mode) and the patch operand
// gathered as above.  Split over pixel slabs, partials reduced by a second kernel into the OIHW gradient.
//
// The image value that enters the MFMA is round_T(img * scale + shift), exactly what fsr_image_to_nhwc would have
// stored, so results match the padded-tensor path bit for bit in f32 and to bf16 rounding in bf16.
#include "fsr_common.h"
#include "fsr_host.h"

namespace {

struct C3Args {
  const float* img;
  long long sn, sc, sh, sw;
  int N, H, W;
  float scale[3], shift[3];
  const void* wpk;      // forward: [cout_pad][32] T, k = (ky*3+kx)*3 + ci
  const float* bias;
  const float* prelu;
  int act;
  float slope;
  int cout;             // multiple of 16
  void* out;            // forward: [N,H,W,cout] T
  void* preact;
  unsigned char* signs; // forward, optional: [N,H,W,cout/8] -- bit (c & 7) of byte c >> 3 = (out[c] > 0), read by conv_s2d3.hip as the activation-gradient mask
  const void* dz;       // wgrad: [N,H,W,cout] T
  float* ws;            // wgrad partials [slab][cout][32]
  int tiles_x, tiles_y, tiles_per_slab, ntiles;
};

constexpr int PW = 18;   // patch width of a 16-column tile

// k -> offset inside the [rows][18][3] float patch of the value tap k/3, channel k%3 (k < 27)
__device__ __forceinline__ int patch_off(int k) {
  const int tap = k / 3, ci = k - tap * 3;
  return ((tap / 3) * PW + (tap % 3)) * 3 + ci;
}

template <typename T> __device__ __forceinline__ float round_T(float v);
template <> __device__ __forceinline__ float round_T<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_T<bf16_t>(float v) { return bf2f(f2bf(v)); }
template <> __device__ __forceinline__ float round_T<f16_t>(float v) { return (float)(f16_t)v; }

// stage rows [y0-1, y0+TH] x cols [x0-1, x0+16] of image n, normalised and rounded to T, as floats.
// All loads of a thread are issued before the first use: the address of an out-of-image element is clamped to the
// image origin and its value discarded (a load under a divergent branch is followed by its own s_waitcnt, which made
// the patch four sequential memory round trips), and scale / shift are picked with selects, not indexed from memory.
template <typename T, int TH>
__device__ __forceinline__ void stage_patch(const C3Args& a, float* patch, int n, int y0, int x0, int tid) {
  constexpr int PN = (TH + 2) * PW * 3;
  constexpr int PPT = (PN + 255) / 256;
  const float* src = a.img + n * a.sn;
  float v[PPT];
  unsigned okmask = 0u;
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int i = tid + j * 256;
    const int ci = i % 3, p = i / 3;
    const int px = p % PW, py = p / PW;
    const int y = y0 - 1 + py, x = x0 - 1 + px;
    const bool ok = i < PN && y >= 0 && y < a.H && x >= 0 && x < a.W;
    okmask |= (ok ? 1u : 0u) << j;
    v[j] = src[ok ? ci * a.sc + y * a.sh + x * a.sw : 0];
  }
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int i = tid + j * 256;
    const int ci = i % 3;
    const float sc = ci == 0 ? a.scale[0] : (ci == 1 ? a.scale[1] : a.scale[2]);
    const float sh = ci == 0 ? a.shift[0] : (ci == 1 ? a.shift[1] : a.shift[2]);
    if (i < PN) patch[i] = ((okmask >> j) & 1u) ? round_T<T>(v[j] * sc + sh) : 0.f;
  }
}

// ------------------------------------------------------------------ forward
// ST = storage type of the 64-channel tensors.  FSR_X3 runs the exact-f32 arithmetic (T = float: the layer is bound by writing
// its output, and K = 27 f32 MFMA steps cost less than that) on x3 storage (ST = x3_t): better than the three-MFMA form, for free.
template <typename T, typename ST = T>
__global__ __launch_bounds__(256) void conv_c3_fwd_kernel(const C3Args a) {
  constexpr int TH = 16;
  __shared__ float patch[(TH + 2) * PW * 3];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  int bid = blockIdx.x;
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int n = bid / a.tiles_y;
  const int nb = blockIdx.y;                      // block of 64 output channels
  const int ntile = (a.cout - nb * 64 >= 64) ? 4 : (a.cout - nb * 64) / 16;
  stage_patch<T, TH>(a, patch, n, ty * TH, tx * 16, tid);

  // this lane's patch offsets and filter fragments (loop invariant)
  const T* wpk = (const T*)a.wpk;
  int koff[8];
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) koff[e] = (8 * lg + e < 27) ? patch_off(8 * lg + e) : -1;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) koff[j] = (lg + 4 * j < 27) ? patch_off(lg + 4 * j) : -1;
  }
  // bf16, full 64-channel block: MFMA tile t holds output channels (t>>1)*32 + (row>>2)*8 + (t&1)*4 + (row&3), so that
  // a lane ends up with two runs of 8 consecutive channels of its pixel = two 16-byte stores
  // (x3 storage, f32 arithmetic: the same channel deal, so that a lane stores 16 bytes of hi and 16 bytes of lo parts per run
  // instead of 8 + 8)
  constexpr bool X3 = std::is_same<ST, x3_t>::value;
  const bool wide = (sizeof(T) == 2 || X3) && ntile == 4;
  s16x8 wf16[4];
  float wf32[4][8];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < ntile) {
      const int ch = wide ? (t >> 1) * 32 + (l15 >> 2) * 8 + (t & 1) * 4 + (l15 & 3) : t * 16 + l15;
      const T* row = wpk + (size_t)(nb * 64 + ch) * 32;
      if constexpr (sizeof(T) == 2) {
        wf16[t] = *(const s16x8*)(row + 8 * lg);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) wf32[t][j] = row[lg + 4 * j];
      }
    }
  }
  float slope = (a.act == FSR_ACT_PRELU) ? a.prelu[0] : a.slope;
  if (a.act == FSR_ACT_NONE) slope = 1.f;
  if (a.act == FSR_ACT_RELU) slope = 0.f;
  __syncthreads();

  ST* outp = (ST*)a.out;
  ST* prep = (ST*)a.preact;
  const int gx = tx * 16 + l15;
  // the lane's bias values, fetched BEFORE the first store (a load issued after a store waits for that store: one
  // counter for both); wide layout: [pair][low / high run], plain: [tile]
  f32x4 bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (a.bias) {
    if (wide) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        bv[2 * p] = *(const f32x4*)(a.bias + nb * 64 + p * 32 + lg * 8);
        bv[2 * p + 1] = *(const f32x4*)(a.bias + nb * 64 + p * 32 + lg * 8 + 4);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t < ntile) bv[t] = *(const f32x4*)(a.bias + nb * 64 + t * 16 + lg * 4);
    }
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int r = wave * 4 + m;                   // row of the tile
    const int gy = ty * TH + r;
    const float* base = patch + (r * PW + l15) * 3;
    float xv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) xv[e] = koff[e] >= 0 ? base[koff[e]] : 0.f;
    s16x8 xf16;
    if constexpr (sizeof(T) == 2) {
      const unsigned p0 = pack2<T>(xv[0], xv[1]), p1 = pack2<T>(xv[2], xv[3]);
      const unsigned p2 = pack2<T>(xv[4], xv[5]), p3 = pack2<T>(xv[6], xv[7]);
      xf16 = __builtin_bit_cast(s16x8, (u32x4){p0, p1, p2, p3});
    }
    if constexpr (sizeof(T) == 2) {
      if (wide) {
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16<T>(wf16[t], xf16, (f32x4){0.f, 0.f, 0.f, 0.f});
        if (gx < a.W && gy < a.H) {
          const size_t off = (((size_t)n * a.H + gy) * a.W + gx) * a.cout + nb * 64 + lg * 8;
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            f32x4 lo = acc[2 * p], hi = acc[2 * p + 1];
            lo += bv[2 * p];
            hi += bv[2 * p + 1];
            if (prep)
              fsr_st<4>((u32x4*)((T*)prep + off + p * 32), (u32x4)((u32x4){pack2<T>(lo[0], lo[1]), pack2<T>(lo[2], lo[3]),
                                                       pack2<T>(hi[0], hi[1]), pack2<T>(hi[2], hi[3])}));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              lo[q] = fmaxf(lo[q], 0.f) + slope * fminf(lo[q], 0.f);
              hi[q] = fmaxf(hi[q], 0.f) + slope * fminf(hi[q], 0.f);
            }
            const u32x4 pk = (u32x4){pack2<T>(lo[0], lo[1]), pack2<T>(lo[2], lo[3]), pack2<T>(hi[0], hi[1]), pack2<T>(hi[2], hi[3])};
            fsr_st<4>((u32x4*)((T*)outp + off + p * 32), pk);
            if (a.signs) {      // the lane's eight channels nb * 64 + p * 32 + lg * 8 .. + 7 as one byte of sign bits of the STORED values
              unsigned b = 0u;
#pragma unroll
              for (int q = 0; q < 4; ++q) b |= ((int)(pk[q] << 16) > 0 ? 1u << (2 * q) : 0u) | ((int)(pk[q] & 0xffff0000u) > 0 ? 2u << (2 * q) : 0u);
              a.signs[(((size_t)n * a.H + gy) * a.W + gx) * (a.cout >> 3) + nb * 8 + p * 4 + lg] = (unsigned char)b;
            }
          }
        }
        continue;
      }
    }
    if constexpr (X3) {
      if (wide) {
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[t] = mfma_f32_16x16x4(wf32[t][j], xv[j], acc[t]);
        }
        if (gx < a.W && gy < a.H) {
          const size_t off = (((size_t)n * a.H + gy) * a.W + gx) * a.cout + nb * 64 + lg * 8;
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const f32x4 lo = acc[2 * p] + bv[2 * p], hi = acc[2 * p + 1] + bv[2 * p + 1];
            float v8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            if (prep) V16<x3_t, 4>::st((x3_t*)prep + off + p * 32, v8);
#pragma unroll
            for (int q = 0; q < 8; ++q) v8[q] = fmaxf(v8[q], 0.f) + slope * fminf(v8[q], 0.f);
            V16<x3_t, 4>::st((x3_t*)outp + off + p * 32, v8);
          }
        }
        continue;
      }
    }
    static_for<0, 4>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      if (t < ntile) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (sizeof(T) == 2) {
          acc = mfma16<T>(wf16[t], xf16, acc);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc = mfma_f32_16x16x4(wf32[t][j], xv[j], acc);
        }
        if (gx < a.W && gy < a.H) {
          const int co = nb * 64 + t * 16 + lg * 4;
          const size_t off = (((size_t)n * a.H + gy) * a.W + gx) * a.cout + co;
          acc += bv[t];
          if constexpr (sizeof(T) == 2) {
            if (prep) {
              u32x2 pk;
              pk.x = pack2<T>(acc[0], acc[1]);
              pk.y = pack2<T>(acc[2], acc[3]);
              fsr_st<4>((u32x2*)((T*)prep + off), (u32x2)(pk));
            }
          } else if constexpr (std::is_same<ST, x3_t>::value) {
            if (prep) x3_st4(prep + off, acc);
          } else {
            if (prep) *(f32x4*)(prep + off) = acc;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = fmaxf(acc[q], 0.f) + slope * fminf(acc[q], 0.f);
          if constexpr (sizeof(T) == 2) {
            u32x2 pk;
            pk.x = pack2<T>(acc[0], acc[1]);
            pk.y = pack2<T>(acc[2], acc[3]);
            fsr_st<4>((u32x2*)((T*)outp + off), (u32x2)(pk));
          } else if constexpr (std::is_same<ST, x3_t>::value) {
            x3_st4(outp + off, acc);
          } else {
            *(f32x4*)(outp + off) = acc;
          }
        }
      }
    });
  }
}

// ------------------------------------------------------------------ weight gradient
// Workgroup = a slab of 8x16-pixel tiles; wave w owns output channels [16w, 16w+16) of the current 64-channel block
// (blockIdx.y) x 32 patch columns: two accumulators for the whole slab.
template <typename T, typename ST = T>
__global__ __launch_bounds__(256) void conv_c3_wgrad_kernel(const C3Args a) {
  constexpr int TH = 8;
  constexpr int PA = 64 + 16;                     // dz tile pitch (elements): [128 px][64 co]
  __shared__ float patch[(TH + 2) * PW * 3];
  __shared__ __attribute__((aligned(16))) T dzt[TH * 16 * PA];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const int slab = blockIdx.x, nb = blockIdx.y;
  const int cvalid = a.cout - nb * 64;            // channels of this block that exist (multiple of 16)
  const bool active = wave * 16 < cvalid;
  const ST* dzg = (const ST*)a.dz;
  constexpr int EPB = 16 / (int)sizeof(T);
  constexpr bool X3 = std::is_same<ST, x3_t>::value;

  // patch offsets of this lane's two columns j = l15 and j = 16 + l15 (k = column index; k >= 27 is padding)
  const int off0 = patch_off(l15), off1 = (16 + l15 < 27) ? patch_off(16 + l15) : -1;
  // column 27 (the first padding column) is a column of ones: dW[co][27] = sum over pixels of dz = the BIAS gradient,
  // for free (tile pixels outside the image have dz = 0)
  const float ones27 = (l15 == 11) ? 1.f : 0.f;
  f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int t0 = slab * a.tiles_per_slab;
  const int t1 = (t0 + a.tiles_per_slab < a.ntiles) ? t0 + a.tiles_per_slab : a.ntiles;
  for (int tile = t0; tile < t1; ++tile) {
    const int tx = tile % a.tiles_x;
    const int ty = (tile / a.tiles_x) % a.tiles_y;
    const int n = tile / (a.tiles_x * a.tiles_y);
    __syncthreads();
    stage_patch<T, TH>(a, patch, n, ty * TH, tx * 16, tid);
    if constexpr (X3) {   // x3 dz -> float tile: 8-channel units, hi and lo 16 bytes each, joined on the way to LDS
      constexpr int DU = TH * 16 * 8, DPT = DU / 256;
      u32x4 dh[DPT], dl[DPT];
#pragma unroll
      for (int j = 0; j < DPT; ++j) {
        const int u = tid + j * 256;
        const int unit = u % 8, p = u / 8;
        const int y = ty * TH + p / 16, x = tx * 16 + (p & 15);
        const bool ok = y < a.H && x < a.W && unit * 8 < cvalid;
        const char* hp = (const char*)x3_hi_ptr(dzg + (ok ? (((size_t)n * a.H + y) * a.W + x) * a.cout + nb * 64 + unit * 8 : 0));
        const u32x4 th = *(const u32x4*)hp, tl = *(const u32x4*)(hp + 64);
        dh[j] = ok ? th : (u32x4){0u, 0u, 0u, 0u};
        dl[j] = ok ? tl : (u32x4){0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int j = 0; j < DPT; ++j) {
        const int u = tid + j * 256;
        float* d = (float*)dzt + (u / 8) * PA + (u % 8) * 8;
        *(f32x4*)d = (f32x4){x3_join_lo(dh[j][0], dl[j][0]), x3_join_hi(dh[j][0], dl[j][0]), x3_join_lo(dh[j][1], dl[j][1]), x3_join_hi(dh[j][1], dl[j][1])};
        *(f32x4*)(d + 4) = (f32x4){x3_join_lo(dh[j][2], dl[j][2]), x3_join_hi(dh[j][2], dl[j][2]), x3_join_lo(dh[j][3], dl[j][3]), x3_join_hi(dh[j][3], dl[j][3])};
      }
    } else {   // dz tile: all loads of a thread first, then the LDS writes (a load consumed under its own branch costs one
        // memory round trip each)
      constexpr int DU = TH * 16 * (64 / EPB), DPT = DU / 256;
      static_assert(DU % 256 == 0, "dz tile units per thread");
      u32x4 dv[DPT];
#pragma unroll
      for (int j = 0; j < DPT; ++j) {
        const int u = tid + j * 256;
        const int unit = u % (64 / EPB), p = u / (64 / EPB);
        const int y = ty * TH + p / 16, x = tx * 16 + (p & 15);
        const bool ok = y < a.H && x < a.W && unit * EPB < cvalid;
        const u32x4 t = *(const u32x4*)((const T*)dzg + (ok ? (((size_t)n * a.H + y) * a.W + x) * a.cout + nb * 64 + unit * EPB : 0));
        dv[j] = ok ? t : (u32x4){0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int j = 0; j < DPT; ++j) {
        const int u = tid + j * 256;
        *(u32x4*)(dzt + (u / (64 / EPB)) * PA + (u % (64 / EPB)) * EPB) = dv[j];
      }
    }
    __syncthreads();
    if (!active) continue;
    if constexpr (sizeof(T) == 2) {
      // K step = 32 pixels = tile rows 2s, 2s+1; lane group lg owns pixels 8lg..8lg+7: row 2s + (lg>>1), cols 8(lg&1)+e
      const int qrow = l15 >> 2, qch = (l15 & 3) * 4;
#pragma unroll
      for (int s = 0; s < TH / 2; ++s) {
        const int r = 2 * s + (lg >> 1), c0 = 8 * (lg & 1);
        const T* pa = dzt + (r * 16 + c0 + qrow) * PA + wave * 16 + qch;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(FSR_LDS_PTR(s16x4, pa));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(FSR_LDS_PTR(s16x4, pa + 4 * PA));
        const s16x8 af = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const float* pb = patch + (r * PW + c0) * 3;
        float b0[8], b1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          b0[e] = pb[e * 3 + off0];
          b1[e] = off1 >= 0 ? pb[e * 3 + off1] : ones27;
        }
        const s16x8 bf0 = __builtin_bit_cast(s16x8, (u32x4){pack2<T>(b0[0], b0[1]), pack2<T>(b0[2], b0[3]),
                                                           pack2<T>(b0[4], b0[5]), pack2<T>(b0[6], b0[7])});
        const s16x8 bf1 = __builtin_bit_cast(s16x8, (u32x4){pack2<T>(b1[0], b1[1]), pack2<T>(b1[2], b1[3]),
                                                           pack2<T>(b1[4], b1[5]), pack2<T>(b1[6], b1[7])});
        acc0 = mfma16<T>(af, bf0, acc0);
        acc1 = mfma16<T>(af, bf1, acc1);
      }
    } else {
      // K step = 4 pixels of one row: pixel k = lg -> (row s>>2, column 4(s&3) + lg)
#pragma unroll 4
      for (int s = 0; s < TH * 4; ++s) {
        const int r = s >> 2, c = 4 * (s & 3) + lg;
        const float av = dzt[(r * 16 + c) * PA + wave * 16 + l15];
        const float* pb = patch + (r * PW + c) * 3;
        acc0 = mfma_f32_16x16x4(av, pb[off0], acc0);
        acc1 = mfma_f32_16x16x4(av, off1 >= 0 ? pb[off1] : ones27, acc1);
      }
    }
  }
  if (!active) return;
  // D layout: lane column = patch column (l15 / 16 + l15), rows 4*lg + r = output channel inside the wave's tile
  float* o = a.ws + ((size_t)slab * a.cout + nb * 64 + wave * 16 + lg * 4) * 32 + l15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    o[r * 32] = acc0[r];
    o[r * 32 + 16] = acc1[r];
  }
}

// dw[co][ci][ky][kx] += sum_slab ws[slab][co][k], k = (ky*3+kx)*3 + ci;  k == 27: the bias gradient.
// Block = 32 consecutive k of one output channel x 8 slab lanes: lane j adds slabs j, j+8, ... in order, the eight lane
// sums are added in order and one thread adds the result (no atomics: bit-reproducible).
// transposed: the kernel ran with the roles swapped (image = the 3-channel gradient of a cout -> 3 convolution, dz = that
// convolution's 64-channel input), so row co / column (tap, c3) is dW[c3][co][8 - tap] of the 3-output filter.
__global__ __launch_bounds__(256) void conv_c3_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nslab,
                                                                   int cout, float* __restrict__ dbias, int transposed) {
  __shared__ float red[8][32];
  const int total = cout * 32;
  const int k = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int co = blockIdx.x;
  const int i = co * 32 + k;
  float s = 0.f;
  const bool live = k < 27 || (k == 27 && dbias);
  if (live) {
    int j = pl;
    for (; j + 24 < nslab; j += 32) {
      const float v0 = ws[(size_t)j * total + i], v1 = ws[(size_t)(j + 8) * total + i];
      const float v2 = ws[(size_t)(j + 16) * total + i], v3 = ws[(size_t)(j + 24) * total + i];
      s = ((s + v0) + v1) + v2 + v3;
    }
    for (; j < nslab; j += 8) s += ws[(size_t)j * total + i];
  }
  red[pl][k] = s;
  __syncthreads();
  if (pl == 0 && live) {
    float r = red[0][k];
#pragma unroll
    for (int j = 1; j < 8; ++j) r += red[j][k];
    const int tap = k / 3, ci = k - tap * 3;
    float* o = (k == 27) ? dbias + co : (transposed ? dw + ((size_t)ci * cout + co) * 9 + (8 - tap) : dw + ((size_t)co * 3 + ci) * 9 + tap);
    *o += r;
  }
}

// [cout_pad][32] filter image for the forward kernel
template <typename T>
__global__ void pack_c3_kernel(const float* __restrict__ w, T* __restrict__ out, int cout, int rows_pad, int transposed) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows_pad * 32; i += gridDim.x * blockDim.x) {
    const int k = i & 31, row = i >> 5;
    float v = 0.f;
    if (row < cout && k < 27) {
      const int tap = k / 3, ci = k - tap * 3;
      // transposed: w is the [3][cout][3][3] filter of a cout -> 3 convolution; its data gradient is the 3 -> cout
      // convolution with rows = input channels and the taps flipped
      v = transposed ? w[((size_t)ci * cout + row) * 9 + (8 - tap)] : w[((size_t)row * 3 + ci) * 9 + tap];
    }
    ElemIO<T>::st(out + i, v);
  }
}

int fill_args(C3Args& a, const char* what, int dtype, const float* img, long long sn, long long sc, long long sh,
              long long sw, int n, int h, int w, const float (&scale3)[3], const float (&shift3)[3], int cout) {
  if (dtype != FSR_F32 && dtype != FSR_BF16 && dtype != FSR_F16 && dtype != FSR_X3) return fsr_fail(-2, "%s: unknown dtype %d", what, dtype);
  if (dtype == FSR_X3 && cout % 32) return fsr_fail(-2, "%s: x3 tensors have a multiple of 32 channels", what);
  if (!img) return fsr_fail(-1, "%s: null image", what);
  if (n <= 0 || h <= 0 || w <= 0) return fsr_fail(-2, "%s: bad dims", what);
  if (cout <= 0 || cout % 16) return fsr_fail(-2, "%s: cout=%d is not a multiple of 16", what, cout);
  if ((long long)n * h * w * cout >= (1LL << 31)) return fsr_fail(-2, "%s: tensors with 2^31 or more elements are not supported", what);
  a.img = img;
  a.sn = sn; a.sc = sc; a.sh = sh; a.sw = sw;
  a.N = n; a.H = h; a.W = w;
  for (int i = 0; i < 3; ++i) {
    a.scale[i] = scale3[i];
    a.shift[i] = shift3[i];
  }
  a.cout = cout;
  return 0;
}

}  // namespace

extern "C" int fsr_pack_conv3x3_c3(int dtype, const float* w_oihw, int cout, void* packed, int transposed, fsr_stream_t stream_) {
  if (!w_oihw || !packed || cout <= 0) return fsr_fail(-1, "fsr_pack_conv3x3_c3: bad argument");
  const int rows_pad = (cout + 15) / 16 * 16;
  const int blocks = (rows_pad * 32 + 255) / 256;
  if (dtype == FSR_BF16)
    hipLaunchKernelGGL(pack_c3_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, w_oihw, (bf16_t*)packed, cout, rows_pad, transposed);
  else if (dtype == FSR_F16)
    hipLaunchKernelGGL(pack_c3_kernel<f16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, w_oihw, (f16_t*)packed, cout, rows_pad, transposed);
  else if (dtype == FSR_F32 || dtype == FSR_X3)      // x3: the first-layer kernels compute in f32 (float filter image)
    hipLaunchKernelGGL(pack_c3_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, w_oihw, (float*)packed, cout, rows_pad, transposed);
  else
    return fsr_fail(-2, "fsr_pack_conv3x3_c3: unknown dtype %d", dtype);
  return fsr_check_launch("pack_c3_kernel");
}

extern "C" int fsr_conv3x3_c3_fwd(int dtype, const float* img, long long sn, long long sc, long long sh, long long sw, int n,
                                  int h, int w, float scale0, float scale1, float scale2, float shift0, float shift1,
                                  float shift2, const void* packed_w, const float* bias, int act, float slope, const float* prelu_weight, int cout, void* out,
                                  void* preact, void* signs, fsr_stream_t stream_) {
  C3Args a = {};
  const float scale3[3] = {scale0, scale1, scale2}, shift3[3] = {shift0, shift1, shift2};
  if (int rc = fill_args(a, "fsr_conv3x3_c3_fwd", dtype, img, sn, sc, sh, sw, n, h, w, scale3, shift3, cout)) return rc;
  if (!packed_w || !out) return fsr_fail(-1, "fsr_conv3x3_c3_fwd: null argument");
  if (act == FSR_ACT_PRELU && !prelu_weight) return fsr_fail(-1, "fsr_conv3x3_c3_fwd: PReLU needs its weight");
  if (act == FSR_ACT_TANH) return fsr_fail(-2, "fsr_conv3x3_c3_fwd: tanh is not a first-layer activation");
  a.wpk = packed_w;
  a.bias = bias;
  a.prelu = prelu_weight;
  a.act = act;
  a.slope = slope;
  a.out = out;
  a.preact = preact;
  if (dtype == FSR_X3 && ((((size_t)out | (size_t)preact) & 127) != 0)) return fsr_fail(-2, "fsr_conv3x3_c3_fwd: x3 tensors must be 128-byte aligned");
  if (signs && (dtype == FSR_F32 || dtype == FSR_X3 || cout % 64 != 0)) return fsr_fail(-2, "fsr_conv3x3_c3_fwd: sign bits are written by the 16-bit kernels for cout %% 64 == 0");
  a.signs = (unsigned char*)signs;
  a.tiles_x = (w + 15) / 16;
  a.tiles_y = (h + 15) / 16;
  const long long nwg = (long long)a.tiles_x * a.tiles_y * n;
  if (nwg > 0x7fffffffLL) return fsr_fail(-2, "fsr_conv3x3_c3_fwd: bad grid");
  const dim3 grid((unsigned)nwg, (unsigned)((cout + 63) / 64));
  if (dtype == FSR_BF16) hipLaunchKernelGGL(conv_c3_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream_, a);
  else if (dtype == FSR_F16) hipLaunchKernelGGL(conv_c3_fwd_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream_, a);
  else if (dtype == FSR_X3) hipLaunchKernelGGL((conv_c3_fwd_kernel<float, x3_t>), grid, dim3(256), 0, (hipStream_t)stream_, a);
  else hipLaunchKernelGGL(conv_c3_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream_, a);
  return fsr_check_launch("conv_c3_fwd_kernel");
}

static int c3_wgrad_slabs(int n, int h, int w, int* tiles_x, int* tiles_y, int* per) {
  *tiles_x = (w + 15) / 16;
  *tiles_y = (h + 7) / 8;
  const long long ntiles = (long long)*tiles_x * *tiles_y * n;
  long long want = 1024;                          // bandwidth-bound: four workgroups per CU in flight
  if (want > ntiles) want = ntiles;
  *per = (int)((ntiles + want - 1) / want);
  return (int)((ntiles + *per - 1) / *per);
}

extern "C" size_t fsr_conv3x3_c3_wgrad_workspace(int n, int h, int w, int cout) {
  int tx, ty, per;
  if (n <= 0 || h <= 0 || w <= 0 || cout <= 0) return 0;
  return (size_t)c3_wgrad_slabs(n, h, w, &tx, &ty, &per) * cout * 32 * sizeof(float);
}

extern "C" int fsr_conv3x3_c3_wgrad(int dtype, const float* img, long long sn, long long sc, long long sh, long long sw,
                                    int n, int h, int w, float scale0, float scale1, float scale2, float shift0,
                                    float shift1, float shift2, const void* dz, int cout, float* dw_oihw, float* dbias, void* workspace, int transposed,
                                    fsr_stream_t stream_) {
  C3Args a = {};
  const float scale3[3] = {scale0, scale1, scale2}, shift3[3] = {shift0, shift1, shift2};
  if (int rc = fill_args(a, "fsr_conv3x3_c3_wgrad", dtype, img, sn, sc, sh, sw, n, h, w, scale3, shift3, cout)) return rc;
  if (!dz || !dw_oihw || !workspace) return fsr_fail(-1, "fsr_conv3x3_c3_wgrad: null argument");
  if (dtype == FSR_X3 && (((size_t)dz & 127) != 0)) return fsr_fail(-2, "fsr_conv3x3_c3_wgrad: x3 tensors must be 128-byte aligned");
  a.dz = dz;
  a.ws = (float*)workspace;
  const int nslab = c3_wgrad_slabs(n, h, w, &a.tiles_x, &a.tiles_y, &a.tiles_per_slab);
  a.ntiles = a.tiles_x * a.tiles_y * n;
  const dim3 grid((unsigned)nslab, (unsigned)((cout + 63) / 64));
  if (dtype == FSR_BF16) hipLaunchKernelGGL(conv_c3_wgrad_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream_, a);
  else if (dtype == FSR_F16) hipLaunchKernelGGL(conv_c3_wgrad_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream_, a);
  else if (dtype == FSR_X3) hipLaunchKernelGGL((conv_c3_wgrad_kernel<float, x3_t>), grid, dim3(256), 0, (hipStream_t)stream_, a);
  else hipLaunchKernelGGL(conv_c3_wgrad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream_, a);
  if (int rc = fsr_check_launch("conv_c3_wgrad_kernel")) return rc;
  hipLaunchKernelGGL(conv_c3_wgrad_reduce_kernel, dim3(cout), dim3(256), 0, (hipStream_t)stream_, (const float*)workspace,
                     dw_oihw, nslab, cout, dbias, transposed);
  return fsr_check_launch("conv_c3_wgrad_reduce_kernel");
}
