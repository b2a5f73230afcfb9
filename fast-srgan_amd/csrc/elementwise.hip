// HBM-bound elementwise / reduction kernels of the Fast-SRGAN hot path (NHWC activations).
//
//   InstanceNorm2d apply (+PReLU / LeakyReLU, + residual)   /root/reference/model.py:55-56,65,69,94,115,132-133
//   backward of the same (two per-(n,c) reductions + apply)  autograd of the above
//   activation backward of fused conv epilogues + bias grad  model.py:37,77,145 ; vgg ReLU model.py:8
//   3-channel image <-> zero-padded NHWC, tanh backward      model.py:20-22 (VGG normalise), :109
//   MaxPool2d(2,2) forward / backward                        vgg19.features pools (model.py:8)
//
// Every kernel moves 16 bytes per lane per access (8 bf16 / 4 f32), grid-strides over a capped grid
// and keeps all arithmetic in f32.  They are bandwidth-bound: bytes moved per element are listed
// in DESIGN.md.
#include "fsr_common.h"
#include "fsr_host.h"

namespace {

constexpr float kEps = 1e-5f;  // torch.nn.InstanceNorm2d default eps

// one 16-byte unit of T viewed as floats (fsr_common.h), stored with this family's non-temporal setting
template <typename T> struct EV : V16<T, 8> {};
// derivative with respect to the pre-activation, evaluated from the pre-activation
__device__ __forceinline__ float act_dz(float z, int act, float slope) {
  if (act == FSR_ACT_NONE) return 1.f;
  if (act == FSR_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  return z > 0.f ? 1.f : slope;
}

// (sum, sum of squares) pairs of E consecutive channels -> mean / rstd, with 16-byte loads (2E floats, 8E-byte aligned)
template <int E>
__device__ __forceinline__ void load_mean_rstd(const float* __restrict__ st, float inv, float (&mean)[E], float (&rstd)[E]) {
#pragma unroll
  for (int i = 0; i < E / 2; ++i) {
    const f32x4 v = *(const f32x4*)(st + 4 * i);
    mean[2 * i] = v[0] * inv;
    rstd[2 * i] = rsqrtf(fmaxf(v[1] * inv - mean[2 * i] * mean[2 * i], 0.f) + kEps);
    mean[2 * i + 1] = v[2] * inv;
    rstd[2 * i + 1] = rsqrtf(fmaxf(v[3] * inv - mean[2 * i + 1] * mean[2 * i + 1], 0.f) + kEps);
  }
}

// blocks along x for a (blocks, images) grid: about 2048 workgroups in total, each thread doing >= 1 unit
inline int image_blocks(long long units_per_image, int n) {
  long long b = (units_per_image + 255) / 256;
  const long long cap = (2048 + n - 1) / n;
  if (b > cap) b = cap;
  return b < 1 ? 1 : (int)b;
}

inline int row_blocks(int units_per_row) {
  int b = (units_per_row + 255) / 256;
  return b < 1 ? 1 : (b > 64 ? 64 : b);
}

// ------------------------------------------------------------------ InstanceNorm apply
// Grid = (pixel blocks, images).  256 % (c / E) == 0 (host checked), so a thread keeps ONE channel unit for its
// whole grid-stride walk: mean / rstd of its E channels are computed once, and the loop body is two 16-byte
// loads, E fmas and one 16-byte store -- no integer division, no per-element rsqrt.
template <typename T>
__global__ __launch_bounds__(256) void instnorm_act_fwd_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                               const T* __restrict__ res, int act, float slope_in,
                                                               const float* __restrict__ prelu, T* __restrict__ out,
                                                               int hw, int c) {
  constexpr int E = EV<T>::N;
  const int cu = c / E;
  const int n = blockIdx.y;
  const float slope = (act == FSR_ACT_PRELU) ? prelu[0] : (act == FSR_ACT_NONE ? 1.f : (act == FSR_ACT_RELU ? 0.f : slope_in));
  const float inv = 1.f / (float)hw;
  const int unit = threadIdx.x % cu;
  float mean[E], rstd[E];
  {
    load_mean_rstd<E>(stats + ((size_t)n * c + unit * E) * 2, inv, mean, rstd);
  }
  const unsigned units = (unsigned)hw * cu;          // per image
  const size_t img = (size_t)n * units * E;
  const unsigned step = gridDim.x * 256;
  auto apply = [&](float (&v)[E], const float (&r)[E]) {
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const float z = (v[i] - mean[i]) * rstd[i];
      v[i] = fmaxf(z, 0.f) + slope * fminf(z, 0.f) + (res ? r[i] : 0.f);
    }
  };
  unsigned u = blockIdx.x * 256 + threadIdx.x;
  for (; u + step < units; u += 2 * step) {      // two units per trip: up to four 16-byte loads in flight per lane
    float va[E], ra[E], vb[E], rb[E];
    EV<T>::ld(x + img + (size_t)u * E, va);
    EV<T>::ld(x + img + (size_t)(u + step) * E, vb);
    if (res) {
      EV<T>::ld(res + img + (size_t)u * E, ra);
      EV<T>::ld(res + img + (size_t)(u + step) * E, rb);
    }
    apply(va, ra);
    apply(vb, rb);
    EV<T>::st(out + img + (size_t)u * E, va);
    EV<T>::st(out + img + (size_t)(u + step) * E, vb);
  }
  if (u < units) {
    float v[E], r[E];
    EV<T>::ld(x + img + (size_t)u * E, v);
    if (res) EV<T>::ld(res + img + (size_t)u * E, r);
    apply(v, r);
    EV<T>::st(out + img + (size_t)u * E, v);
  }
}

// Backward phase 1.  Workgroup = one image x a slab of pixels x all channels: thread t owns channel
// unit t % cu and walks pixels t / cu, t / cu + 256 / cu, ...; partial sums meet in LDS (added in a fixed
// order) and the workgroup's vector goes to its slot [n][slab][c][2] of the scratch buffer (reduce.hip adds the
// slabs in order); the PReLU-slope partial goes to pp[n * slabs + slab].
template <typename T>
__global__ __launch_bounds__(256) void instnorm_act_bwd_reduce_kernel(const T* __restrict__ g, const T* __restrict__ x,
                                                                      const float* __restrict__ stats, int act,
                                                                      float slope_in, const float* __restrict__ prelu,
                                                                      float* __restrict__ sums, float* __restrict__ dprelu,
                                                                      int hw, int c, int slabs) {
  constexpr int E = EV<T>::N;
  __shared__ float red[4][256 / 4 * 2 * E];   // [wave][unit slot][2E]: one value per (wave, channel unit, quantity)
  __shared__ float red_p[4];
  const int cu = c / E;            // <= 256 (host checked)
  const int rows = 256 / cu;       // pixel rows walked in parallel
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x / slabs, slab = blockIdx.x % slabs;
  const int per = (hw + slabs - 1) / slabs;
  const int p0 = slab * per, p1 = (p0 + per < hw) ? p0 + per : hw;
  const float slope = (act == FSR_ACT_PRELU) ? prelu[0] : slope_in;
  const float inv = 1.f / (float)hw;
  const int unit = tid % cu, row = tid / cu;
  float s1[E], s2[E], mean[E], rstd[E];
  float dp = 0.f;
#pragma unroll
  for (int i = 0; i < E; ++i) s1[i] = s2[i] = 0.f;
  if (row < rows) {
    load_mean_rstd<E>(stats + ((size_t)n * c + unit * E) * 2, inv, mean, rstd);
    auto body = [&](const float (&gv)[E], const float (&xv)[E]) {
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const float xh = (xv[i] - mean[i]) * rstd[i];
        const float gz = gv[i] * act_dz(xh, act, slope);
        s1[i] += gz;
        s2[i] += gz * xh;
        dp += gv[i] * fminf(xh, 0.f);
      }
    };
    int p = p0 + row;
    for (; p + rows < p1; p += 2 * rows) {     // two pixels per trip: four 16-byte loads in flight per lane
      const size_t off = ((size_t)n * hw + p) * c + unit * E;
      float ga[E], xa[E], gb[E], xb[E];
      EV<T>::ld(g + off, ga);
      EV<T>::ld(x + off, xa);
      EV<T>::ld(g + off + (size_t)rows * c, gb);
      EV<T>::ld(x + off + (size_t)rows * c, xb);
      body(ga, xa);
      body(gb, xb);
    }
    if (p < p1) {
      const size_t off = ((size_t)n * hw + p) * c + unit * E;
      float ga[E], xa[E];
      EV<T>::ld(g + off, ga);
      EV<T>::ld(x + off, xa);
      body(ga, xa);
    }
  }
  // Lanes of a wave that own the same channel unit (lane % cu equal; cu a power of two <= 64, or every lane its own unit) meet
  // by a fixed xor butterfly; the four waves then meet in LDS and are added in order.  (The pixel order of a thread and the
  // butterfly are fixed, so the sums are bit-reproducible.)
  const int ucu = cu < 64 ? cu : 64;          // distinct units inside one wave
#pragma unroll
  for (int i = 0; i < E; ++i) {
    for (int o = 32; o >= ucu; o >>= 1) {
      s1[i] += __shfl_xor(s1[i], o, 64);
      s2[i] += __shfl_xor(s2[i], o, 64);
    }
  }
  if (lane < ucu) {
#pragma unroll
    for (int i = 0; i < E; ++i) {
      red[wave][lane * 2 * E + 2 * i] = s1[i];
      red[wave][lane * 2 * E + 2 * i + 1] = s2[i];
    }
  }
  if (dprelu) {
    dp = wave_sum(dp);
    if (lane == 0) red_p[wave] = dp;
  }
  __syncthreads();
  // output t: channel unit u = t / (2E), quantity j = t % (2E).  Unit u lives in waves w with (w * 64 + lane) % cu == u:
  // cu <= 64: lane u of every wave; cu > 64: lane u % 64 of the waves w = u / 64, u / 64 + cu / 64, ...
  for (int t = tid; t < cu * 2 * E; t += 256) {
    const int u = t / (2 * E), j = t - u * 2 * E;
    float s = 0.f;
    if (cu <= 64) {
#pragma unroll
      for (int w = 0; w < 4; ++w) s += red[w][u * 2 * E + j];
    } else {
      for (int w = u >> 6; w < 4; w += cu >> 6) s += red[w][(u & 63) * 2 * E + j];
    }
    sums[((size_t)blockIdx.x * c + u * E + (j >> 1)) * 2 + (j & 1)] = s;
  }
  if (dprelu && tid == 0) dprelu[blockIdx.x] = red_p[0] + red_p[1] + red_p[2] + red_p[3];
}

template <typename T>
__global__ __launch_bounds__(256) void instnorm_act_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ x,
                                                                     const float* __restrict__ stats,
                                                                     const float* __restrict__ sums, int act,
                                                                     float slope_in, const float* __restrict__ prelu,
                                                                     T* __restrict__ dx, int hw, int c) {
  constexpr int E = EV<T>::N;
  const int cu = c / E;
  const int n = blockIdx.y;
  const float slope = (act == FSR_ACT_PRELU) ? prelu[0] : (act == FSR_ACT_NONE ? 1.f : (act == FSR_ACT_RELU ? 0.f : slope_in));
  const float inv = 1.f / (float)hw;
  const int unit = threadIdx.x % cu;
  float mean[E], rstd[E], m1[E], m2[E];
  {
    load_mean_rstd<E>(stats + ((size_t)n * c + unit * E) * 2, inv, mean, rstd);
    const float* sm = sums + ((size_t)n * c + unit * E) * 2;
#pragma unroll
    for (int i = 0; i < E / 2; ++i) {
      const f32x4 v = *(const f32x4*)(sm + 4 * i);
      m1[2 * i] = v[0] * inv;
      m2[2 * i] = v[1] * inv;
      m1[2 * i + 1] = v[2] * inv;
      m2[2 * i + 1] = v[3] * inv;
    }
  }
  const unsigned units = (unsigned)hw * cu;
  const size_t img = (size_t)n * units * E;
  const unsigned step = gridDim.x * 256;
  auto apply = [&](const float (&gv)[E], float (&xv)[E]) {
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const float xh = (xv[i] - mean[i]) * rstd[i];
      const float gz = gv[i] * (xh > 0.f ? 1.f : slope);
      xv[i] = rstd[i] * (gz - m1[i] - xh * m2[i]);
    }
  };
  unsigned u = blockIdx.x * 256 + threadIdx.x;
  for (; u + step < units; u += 2 * step) {      // two units per trip: four 16-byte loads in flight per lane
    float ga[E], xa[E], gb[E], xb[E];
    EV<T>::ld(g + img + (size_t)u * E, ga);
    EV<T>::ld(x + img + (size_t)u * E, xa);
    EV<T>::ld(g + img + (size_t)(u + step) * E, gb);
    EV<T>::ld(x + img + (size_t)(u + step) * E, xb);
    apply(ga, xa);
    apply(gb, xb);
    EV<T>::st(dx + img + (size_t)u * E, xa);
    EV<T>::st(dx + img + (size_t)(u + step) * E, xb);
  }
  if (u < units) {
    float gv[E], xv[E];
    EV<T>::ld(g + img + (size_t)u * E, gv);
    EV<T>::ld(x + img + (size_t)u * E, xv);
    apply(gv, xv);
    EV<T>::st(dx + img + (size_t)u * E, xv);
  }
}

// ------------------------------------------------------------------ activation backward of a fused conv epilogue
// Workgroup layout as in the reduce kernel above (channel unit x pixel rows) so that the bias
// gradient is a column sum.  Pixel-shuffled tensors: the bias index depends on the pixel parity.
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ g, const T* __restrict__ saved, int act,
                                                      float slope_in, const float* __restrict__ prelu, T* __restrict__ dz,
                                                      float* __restrict__ dbias, float* __restrict__ dprelu, int h, int w,
                                                      int c, int ps, int slabs) {
  constexpr int E = EV<T>::N;
  __shared__ float red[256 * E];
  __shared__ float red_p[4];
  const int cu = c / E;
  const int rows = 256 / cu;
  const int tid = threadIdx.x;
  const int hw = h * w;
  // ps: a slab is one (row parity, column parity) class of pixels so that a thread's column sums
  // belong to one bias entry per channel
  const int nclass = ps ? 4 : 1;
  int b = blockIdx.x;
  const int cls = b % nclass;
  b /= nclass;
  const int slab = b % slabs, n = b / slabs;
  const int ch = ps ? h / 2 : h, cw = ps ? w / 2 : w;  // class grid
  const int cnt = ch * cw;
  const int per = (cnt + slabs - 1) / slabs;
  const int p0 = slab * per, p1 = (p0 + per < cnt) ? p0 + per : cnt;
  const float slope = (act == FSR_ACT_PRELU) ? prelu[0] : slope_in;
  const int unit = tid % cu, row = tid / cu;
  float s1[E];
  float dp = 0.f;
#pragma unroll
  for (int i = 0; i < E; ++i) s1[i] = 0.f;
  if (row < rows) {
    for (int p = p0 + row; p < p1; p += rows) {
      int py = p / cw, px = p % cw;
      if (ps) {
        py = 2 * py + (cls >> 1);
        px = 2 * px + (cls & 1);
      }
      const size_t off = ((size_t)n * hw + (size_t)py * w + px) * c + unit * E;
      float gv[E], sv[E];
      EV<T>::ld(g + off, gv);
      EV<T>::ld(saved + off, sv);
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const float d = gv[i] * act_dz(sv[i], act, slope);
        dp += gv[i] * fminf(sv[i], 0.f);
        s1[i] += d;
        gv[i] = d;
      }
      if (dz) EV<T>::st(dz + off, gv);
    }
  }
  if (dprelu) {
    dp = wave_sum(dp);
    if ((tid & 63) == 0) red_p[tid >> 6] = dp;
  }
  if (dbias) {
#pragma unroll
    for (int i = 0; i < E; ++i) red[i * 256 + tid] = s1[i];
  }
  __syncthreads();
  // one partial vector per (image, slab) in the scratch buffer: [n * slabs][nclass * c] bias sums, [n * slabs * nclass]
  // slope sums; reduce.hip adds them in order
  if (dbias) {
    for (int t = tid; t < cu * E; t += 256) {
      const int q = t / cu, un = t % cu;
      float s = 0.f;
      for (int r = 0; r < rows; ++r) s += red[q * 256 + r * cu + un];
      const int chn = un * E + q;
      dbias[(size_t)(n * slabs + slab) * (nclass * c) + (ps ? 4 * chn + cls : chn)] = s;
    }
  }
  if (dprelu && tid == 0) dprelu[blockIdx.x] = red_p[0] + red_p[1] + red_p[2] + red_p[3];
}

// ------------------------------------------------------------------ 3-channel images <-> padded NHWC
// Grid = (image rows n*h, column blocks): all index arithmetic is 32-bit and per row.
template <typename T>
__global__ __launch_bounds__(256) void image_to_nhwc_kernel(const float* __restrict__ img, long long sn, long long sc,
                                                            long long sh, long long sw, int h, int w, float a0, float a1,
                                                            float a2, float b0, float b1, float b2, T* __restrict__ out,
                                                            int cpad) {
  constexpr int E = EV<T>::N;
  const int upp = cpad / E;  // 16-byte units per pixel
  const int row = blockIdx.x;
  const int n = row / h, y = row - n * h;
  const float* src = img + n * sn + y * sh;
  T* dst = out + (size_t)row * w * cpad;
  const unsigned units = (unsigned)w * upp;
  for (unsigned u = blockIdx.y * 256 + threadIdx.x; u < units; u += gridDim.y * 256) {
    const unsigned x = u / upp, un = u - x * upp;
    float v[E];
#pragma unroll
    for (int i = 0; i < E; ++i) v[i] = 0.f;
    if (un == 0) {
      const float* s = src + x * sw;
      v[0] = s[0] * a0 + b0;
      v[1] = s[sc] * a1 + b1;
      v[2] = s[2 * sc] * a2 + b2;
    }
    EV<T>::st(dst + (size_t)u * E, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void tanh_bwd_to_nhwc_kernel(const float* __restrict__ g, long long sn, long long sc,
                                                               long long sh, long long sw, const float* __restrict__ y,
                                                               int h, int w, T* __restrict__ dz, int cpad, int nrows,
                                                               float* __restrict__ dbias) {
  constexpr int E = EV<T>::N;
  __shared__ float red[3][4];
  const int upp = cpad / E;
  float b0 = 0.f, b1 = 0.f, b2 = 0.f;
  const unsigned units = (unsigned)w * upp;
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {   // few workgroups: the bias sums end in 3 atomics each
    const int n = row / h, yy = row - n * h;
    const float* src = g + n * sn + yy * sh;
    const float* ty = y + (size_t)row * w * 3;
    T* dst = dz + (size_t)row * w * cpad;
    for (unsigned u = blockIdx.y * 256 + threadIdx.x; u < units; u += gridDim.y * 256) {
      const unsigned x = u / upp, un = u - x * upp;
      float v[E];
#pragma unroll
      for (int i = 0; i < E; ++i) v[i] = 0.f;
      if (un == 0) {
        const float* s = src + x * sw;
        const float* t = ty + x * 3;
        v[0] = s[0] * (1.f - t[0] * t[0]);
        v[1] = s[sc] * (1.f - t[1] * t[1]);
        v[2] = s[2 * sc] * (1.f - t[2] * t[2]);
        b0 += v[0];
        b1 += v[1];
        b2 += v[2];
      }
      EV<T>::st(dst + (size_t)u * E, v);
    }
  }
  if (dbias) {
    b0 = wave_sum(b0);
    b1 = wave_sum(b1);
    b2 = wave_sum(b2);
    if ((threadIdx.x & 63) == 0) {
      red[0][threadIdx.x >> 6] = b0;
      red[1][threadIdx.x >> 6] = b1;
      red[2][threadIdx.x >> 6] = b2;
    }
    __syncthreads();
    if (threadIdx.x < 3)   // this workgroup's slot of the scratch buffer (reduce.hip adds the slots in order)
      dbias[(blockIdx.y * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
  }
}

// uint8 frames -> float in [-1,1]: x / 127.5 - 1 (inference.py:48; correctly rounded f32 division, as torch's)
__global__ __launch_bounds__(256) void u8_to_image_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, long long count) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256)
    dst[i] = (float)src[i] / 127.5f - 1.0f;
}

// ------------------------------------------------------------------ out = a + b (gradient accumulation of a tensor with two consumers)
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, long long units) {
  constexpr int E = EV<T>::N;
  for (long long u = (long long)blockIdx.x * 256 + threadIdx.x; u < units; u += (long long)gridDim.x * 256) {
    float va[E], vb[E];
    EV<T>::ld(a + u * E, va);
    EV<T>::ld(b + u * E, vb);
#pragma unroll
    for (int i = 0; i < E; ++i) va[i] += vb[i];
    EV<T>::st(out + u * E, va);
  }
}

// ------------------------------------------------------------------ MaxPool2d(2,2)
// Grid = (output rows n*oh, column blocks); 32-bit index arithmetic per row.
// `idx` (optional, training): one byte per pooled element -- bits 0..1 = which of the four inputs (row-major, the FIRST one equal
// to the maximum) is the arg-max, bit 2 = maximum > 0 -- is all maxpool2_bwd_kernel needs: the backward pass then reads the pooled
// gradient and these bytes (0.375 tensor-equivalents) instead of the gradient, the pooled output and the full-resolution input (1.5).
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, unsigned char* __restrict__ idx,
                                                           int h, int w, int c) {
  constexpr int E = EV<T>::N;
  const int cu = c / E, oh = h / 2, ow = w / 2;
  const int row = blockIdx.x;
  const int n = row / oh, oy = row - n * oh;
  const T* src = x + ((size_t)n * h + 2 * oy) * w * c;
  T* dst = y + (size_t)row * ow * c;
  const unsigned units = (unsigned)ow * cu;
  const size_t rs = (size_t)w * c;
  for (unsigned u = blockIdx.y * 256 + threadIdx.x; u < units; u += gridDim.y * 256) {
    const unsigned ox = u / cu, un = u - ox * cu;
    const T* s = src + (size_t)(2 * ox) * c + un * E;
    float a[E], b[E], d[E], e[E];
    EV<T>::ld(s, a);
    EV<T>::ld(s + c, b);
    EV<T>::ld(s + rs, d);
    EV<T>::ld(s + rs + c, e);
    unsigned code[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const float m = fmaxf(fmaxf(a[i], b[i]), fmaxf(d[i], e[i]));
      code[i] = (a[i] == m ? 0u : (b[i] == m ? 1u : (d[i] == m ? 2u : 3u))) | (m > 0.f ? 4u : 0u);
      a[i] = m;
    }
    EV<T>::st(dst + (size_t)u * E, a);
    if (idx) {
      unsigned char* ip = idx + (size_t)row * ow * c + (size_t)u * E;
      if constexpr (E == 8) {
        *(u32x2*)ip = (u32x2){code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24), code[4] | (code[5] << 8) | (code[6] << 16) | (code[7] << 24)};
      } else {
        *(unsigned*)ip = code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24);
      }
    }
  }
}

// backward from the arg-max bytes of the forward pass: dx[k] = (k == arg-max && (!relu_mask || max > 0)) ? g : 0
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_idx_kernel(const T* __restrict__ g, const unsigned char* __restrict__ idx,
                                                               T* __restrict__ dx, int h, int w, int c, int relu_mask) {
  constexpr int E = EV<T>::N;
  const int cu = c / E, oh = h / 2, ow = w / 2;
  const int row = blockIdx.x;
  const int n = row / oh, oy = row - n * oh;
  const size_t in0 = ((size_t)n * h + 2 * oy) * w * c;
  const size_t out0 = (size_t)row * ow * c;
  const unsigned units = (unsigned)ow * cu;
  const size_t rs = (size_t)w * c;
  for (unsigned u = blockIdx.y * 256 + threadIdx.x; u < units; u += gridDim.y * 256) {
    const unsigned ox = u / cu, un = u - ox * cu;
    const size_t base = in0 + (size_t)(2 * ox) * c + un * E;
    float gv[E], o[E];
    unsigned code[E];
    EV<T>::ld(g + out0 + (size_t)u * E, gv);
    const unsigned char* ip = idx + out0 + (size_t)u * E;
    if constexpr (E == 8) {
      const u32x2 t = *(const u32x2*)ip;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        code[i] = (t.x >> (8 * i)) & 0xffu;
        code[4 + i] = (t.y >> (8 * i)) & 0xffu;
      }
    } else {
      const unsigned t = *(const unsigned*)ip;
#pragma unroll
      for (int i = 0; i < 4; ++i) code[i] = (t >> (8 * i)) & 0xffu;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int i = 0; i < E; ++i) o[i] = ((code[i] & 3u) == (unsigned)k && (!relu_mask || (code[i] & 4u))) ? gv[i] : 0.f;
      EV<T>::st(dx + base + (size_t)(k >> 1) * rs + (k & 1) * c, o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* __restrict__ g, const T* __restrict__ x,
                                                           const T* __restrict__ y, T* __restrict__ dx, int h, int w, int c,
                                                           int relu_mask) {
  constexpr int E = EV<T>::N;
  const int cu = c / E, oh = h / 2, ow = w / 2;
  const int row = blockIdx.x;
  const int n = row / oh, oy = row - n * oh;
  const size_t in0 = ((size_t)n * h + 2 * oy) * w * c;
  const size_t out0 = (size_t)row * ow * c;
  const unsigned units = (unsigned)ow * cu;
  const size_t rs = (size_t)w * c;
  for (unsigned u = blockIdx.y * 256 + threadIdx.x; u < units; u += gridDim.y * 256) {
    const unsigned ox = u / cu, un = u - ox * cu;
    const size_t base = in0 + (size_t)(2 * ox) * c + un * E;
    float gv[E], yv[E], xv[4][E], o[E];
    bool taken[E];
    EV<T>::ld(g + out0 + (size_t)u * E, gv);
    EV<T>::ld(y + out0 + (size_t)u * E, yv);
#pragma unroll
    for (int k = 0; k < 4; ++k) EV<T>::ld(x + base + (size_t)(k >> 1) * rs + (k & 1) * c, xv[k]);
#pragma unroll
    for (int i = 0; i < E; ++i) taken[i] = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const bool hit = !taken[i] && xv[k][i] == yv[i];
        o[i] = (hit && !(relu_mask && xv[k][i] <= 0.f)) ? gv[i] : 0.f;
        taken[i] = taken[i] || hit;
      }
      EV<T>::st(dx + base + (size_t)(k >> 1) * rs + (k & 1) * c, o);
    }
  }
}

// Head backward straight to a 3-channel float image [n,h,w,3]: dz = g * (1 - y^2).  The first-layer kernels then read it in
// place (data gradient = the 3 -> 64 convolution with the transposed, flipped filter; weight gradient with the roles of
// image and gradient swapped) -- no 32-channel zero-padded copy of a 3-channel tensor is ever written.
__global__ __launch_bounds__(256) void tanh_bwd_image_kernel(const float* __restrict__ g, long long sn, long long sc, long long sh,
                                                             long long sw, const float* __restrict__ y, int h, int w,
                                                             float* __restrict__ dz, int nrows, float* __restrict__ dbias) {
  __shared__ float red[3][4];
  float b0 = 0.f, b1 = 0.f, b2 = 0.f;
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int n = row / h, yy = row - n * h;
    const float* src = g + n * sn + yy * sh;
    const float* ty = y + (size_t)row * w * 3;
    float* dst = dz + (size_t)row * w * 3;
    for (int x = blockIdx.y * 256 + threadIdx.x; x < w; x += gridDim.y * 256) {
      const float* s = src + x * sw;
      const float* t = ty + x * 3;
      const float v0 = s[0] * (1.f - t[0] * t[0]), v1 = s[sc] * (1.f - t[1] * t[1]), v2 = s[2 * sc] * (1.f - t[2] * t[2]);
      dst[x * 3] = v0;
      dst[x * 3 + 1] = v1;
      dst[x * 3 + 2] = v2;
      b0 += v0;
      b1 += v1;
      b2 += v2;
    }
  }
  if (dbias) {
    b0 = wave_sum(b0);
    b1 = wave_sum(b1);
    b2 = wave_sum(b2);
    if ((threadIdx.x & 63) == 0) {
      red[0][threadIdx.x >> 6] = b0;
      red[1][threadIdx.x >> 6] = b1;
      red[2][threadIdx.x >> 6] = b2;
    }
    __syncthreads();
    if (threadIdx.x < 3)
      dbias[(blockIdx.y * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
  }
}

template <typename T> T* P(void* p) { return (T*)p; }
template <typename T> const T* P(const void* p) { return (const T*)p; }

int check_c(const char* what, int dtype, int c) {
  const int e = dtype == FSR_X3 ? 32 : (dtype != FSR_F32 ? 8 : 4);   // x3: whole hi / lo groups of 32 channels
  if (dtype != FSR_F32 && dtype != FSR_BF16 && dtype != FSR_F16 && dtype != FSR_X3) return fsr_fail(-2, "%s: unknown dtype %d", what, dtype);
  if (c <= 0 || c % e != 0) return fsr_fail(-2, "%s: channel count %d is not a multiple of %d", what, c, e);
  return 0;
}
// x3 tensors: the split hi / lo addressing needs 128-byte aligned bases
int check_x3(const char* what, int dtype, const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
  if (dtype != FSR_X3) return 0;
  if ((((size_t)a | (size_t)b | (size_t)c | (size_t)d) & 127) != 0) return fsr_fail(-2, "%s: x3 tensors must be 128-byte aligned", what);
  return 0;
}
int check_reduce_c(const char* what, int dtype, int c) {
  int rc = check_c(what, dtype, c);
  if (rc) return rc;
  const int cu = c / (dtype != FSR_F32 ? 8 : 4);
  if (cu > 256 || 256 % cu != 0) return fsr_fail(-2, "%s: %d channels do not tile a 256-thread workgroup", what, c);
  return 0;
}
int slabs_for(int n, int hw) {
  // enough workgroups to fill 256 CUs a few times over, at least ~64 pixels of work per row walk
  int s = (256 * 4 + n - 1) / n;
  const int maxs = (hw + 63) / 64;
  if (s > maxs) s = maxs;
  return s < 1 ? 1 : s;
}

}  // namespace

#define FSR_DISPATCH_T(dtype, ...)                    \
  if ((dtype) == FSR_BF16) {                          \
    typedef bf16_t T;                                 \
    __VA_ARGS__                                       \
  } else if ((dtype) == FSR_F16) {                    \
    typedef f16_t T;                                  \
    __VA_ARGS__                                       \
  } else if ((dtype) == FSR_X3) {                     \
    typedef x3_t T;                                   \
    __VA_ARGS__                                       \
  } else {                                            \
    typedef float T;                                  \
    __VA_ARGS__                                       \
  }

extern "C" int fsr_instnorm_act_fwd(int dtype, const void* x, const float* stats, const void* res, int act, float slope,
                                    const float* prelu_weight, void* out, int n, int hw, int c, fsr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !stats || !out) return fsr_fail(-1, "fsr_instnorm_act_fwd: null argument");
  if (act == FSR_ACT_PRELU && !prelu_weight) return fsr_fail(-1, "fsr_instnorm_act_fwd: PReLU needs its weight");
  if (int rc = check_reduce_c("fsr_instnorm_act_fwd", dtype, c)) return rc;
  if (int rc = check_x3("fsr_instnorm_act_fwd", dtype, x, res, out)) return rc;
  const long long units = (long long)hw * (c / (dtype != FSR_F32 ? 8 : 4));   // per image
  if (units >= (1LL << 31)) return fsr_fail(-2, "fsr_instnorm_act_fwd: image too large");
  FSR_DISPATCH_T(dtype, hipLaunchKernelGGL(instnorm_act_fwd_kernel<T>, dim3(image_blocks(units, n), n), dim3(256), 0,
                                           stream, P<T>(x), stats, P<T>(res), act, slope, prelu_weight, P<T>(out), hw, c);)
  return fsr_check_launch("instnorm_act_fwd_kernel");
}

extern "C" size_t fsr_instnorm_act_bwd_scratch(int n, int hw, int c) {
  if (n <= 0 || hw <= 0 || c <= 0) return 0;
  return (size_t)n * slabs_for(n, hw) * ((size_t)c * 2 + 1) * sizeof(float);
}

extern "C" int fsr_instnorm_act_bwd_reduce(int dtype, const void* g, const void* x, const float* stats, int act,
                                           float slope, const float* prelu_weight, float* sums, float* dprelu, void* scratch,
                                           int n, int hw, int c, fsr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!g || !x || !stats || !sums || !scratch) return fsr_fail(-1, "fsr_instnorm_act_bwd_reduce: null argument");
  if (act == FSR_ACT_PRELU && !prelu_weight) return fsr_fail(-1, "fsr_instnorm_act_bwd_reduce: PReLU needs its weight");
  if (int rc = check_reduce_c("fsr_instnorm_act_bwd_reduce", dtype, c)) return rc;
  if (int rc = check_x3("fsr_instnorm_act_bwd_reduce", dtype, g, x)) return rc;
  const int slabs = slabs_for(n, hw);
  float* part = (float*)scratch;                         // [n][slabs][c][2]
  float* part_p = part + (size_t)n * slabs * c * 2;      // [n * slabs]
  FSR_DISPATCH_T(dtype, hipLaunchKernelGGL(instnorm_act_bwd_reduce_kernel<T>, dim3(n * slabs), dim3(256), 0, stream,
                                           P<T>(g), P<T>(x), stats, act, slope, prelu_weight, part, dprelu ? part_p : nullptr,
                                           hw, c, slabs);)
  if (int rc = fsr_check_launch("instnorm_act_bwd_reduce_kernel")) return rc;
  if (int rc = fsr_launch_reduce_partials(part, sums, n, slabs, c * 2, c * 2, 0, 0, 1.f, 0, stream)) return rc;
  if (dprelu) return fsr_launch_reduce_partials(part_p, dprelu, 1, n * slabs, 1, 1, 0, 0, 1.f, 0, stream);
  return 0;
}

extern "C" int fsr_instnorm_act_bwd_apply(int dtype, const void* g, const void* x, const float* stats, const float* sums,
                                          int act, float slope, const float* prelu_weight, void* dx, int n, int hw, int c,
                                          fsr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!g || !x || !stats || !sums || !dx) return fsr_fail(-1, "fsr_instnorm_act_bwd_apply: null argument");
  if (act == FSR_ACT_PRELU && !prelu_weight) return fsr_fail(-1, "fsr_instnorm_act_bwd_apply: PReLU needs its weight");
  if (int rc = check_reduce_c("fsr_instnorm_act_bwd_apply", dtype, c)) return rc;
  if (int rc = check_x3("fsr_instnorm_act_bwd_apply", dtype, g, x, dx)) return rc;
  const long long units = (long long)hw * (c / (dtype != FSR_F32 ? 8 : 4));
  if (units >= (1LL << 31)) return fsr_fail(-2, "fsr_instnorm_act_bwd_apply: image too large");
  FSR_DISPATCH_T(dtype, hipLaunchKernelGGL(instnorm_act_bwd_apply_kernel<T>, dim3(image_blocks(units, n), n), dim3(256), 0,
                                           stream, P<T>(g), P<T>(x), stats, sums, act, slope, prelu_weight, P<T>(dx), hw, c);)
  return fsr_check_launch("instnorm_act_bwd_apply_kernel");
}

extern "C" size_t fsr_act_bwd_scratch(int n, int h, int w, int c, int pixel_shuffled) {
  if (n <= 0 || h <= 0 || w <= 0 || c <= 0) return 0;
  const int nclass = pixel_shuffled ? 4 : 1;
  const size_t rows = (size_t)n * slabs_for(n * nclass, h * w / nclass);
  return rows * ((size_t)nclass * c + nclass) * sizeof(float);
}

extern "C" int fsr_act_bwd(int dtype, const void* g, const void* saved, int act, float slope, const float* prelu_weight,
                           void* dz, float* dbias, float* dprelu, void* scratch, int n, int h, int w, int c,
                           int pixel_shuffled, fsr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!g) return fsr_fail(-1, "fsr_act_bwd: null argument");
  if ((dbias || dprelu) && !scratch) return fsr_fail(-1, "fsr_act_bwd: reductions need the scratch buffer");
  if (!dz && !dbias && !dprelu) return fsr_fail(-1, "fsr_act_bwd: nothing to compute");
  if (act != FSR_ACT_NONE && !saved) return fsr_fail(-1, "fsr_act_bwd: the activation needs the saved tensor");
  if (act == FSR_ACT_PRELU && !prelu_weight) return fsr_fail(-1, "fsr_act_bwd: PReLU needs its weight");
  if (act == FSR_ACT_TANH) return fsr_fail(-2, "fsr_act_bwd: tanh is handled by fsr_tanh_bwd_to_nhwc");
  if (int rc = check_reduce_c("fsr_act_bwd", dtype, c)) return rc;
  if (int rc = check_x3("fsr_act_bwd", dtype, g, saved, dz)) return rc;
  if (pixel_shuffled && ((h | w) & 1)) return fsr_fail(-2, "fsr_act_bwd: pixel-shuffled tensors have even extents");
  if (!saved) saved = g;  // FSR_ACT_NONE: only the bias gradient is wanted; act_dz ignores the value
  const int nclass = pixel_shuffled ? 4 : 1;
  const int slabs = slabs_for(n * nclass, h * w / nclass);
  const int rows = n * slabs;
  float* part_b = (float*)scratch;                                   // [rows][nclass * c]
  float* part_p = part_b + (size_t)rows * nclass * c;                // [rows * nclass]
  FSR_DISPATCH_T(dtype, hipLaunchKernelGGL(act_bwd_kernel<T>, dim3(n * slabs * nclass), dim3(256), 0, stream, P<T>(g),
                                           P<T>(saved), act, slope, prelu_weight, P<T>(dz), dbias ? part_b : nullptr,
                                           dprelu ? part_p : nullptr, h, w, c, pixel_shuffled, slabs);)
  if (int rc = fsr_check_launch("act_bwd_kernel")) return rc;
  if (dbias)
    if (int rc = fsr_launch_reduce_partials(part_b, dbias, 1, rows, nclass * c, nclass * c, 0, 0, 1.f, 0, stream)) return rc;
  if (dprelu) return fsr_launch_reduce_partials(part_p, dprelu, 1, rows * nclass, 1, 1, 0, 0, 1.f, 0, stream);
  return 0;
}

extern "C" int fsr_image_to_nhwc(int dtype, const float* img, long long sn, long long sc, long long sh, long long sw,
                                 int n, int h, int w, float scale0, float scale1, float scale2, float shift0, float shift1,
                                 float shift2, void* out, int cpad, fsr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!img || !out) return fsr_fail(-1, "fsr_image_to_nhwc: null argument");
  if (int rc = check_c("fsr_image_to_nhwc", dtype, cpad)) return rc;
  if (int rc = check_x3("fsr_image_to_nhwc", dtype, out)) return rc;
  const int rowunits = w * (cpad / (dtype != FSR_F32 ? 8 : 4));
  FSR_DISPATCH_T(dtype, hipLaunchKernelGGL(image_to_nhwc_kernel<T>, dim3(n * h, row_blocks(rowunits)), dim3(256), 0, stream,
                                           img, sn, sc, sh, sw, h, w, scale0, scale1, scale2, shift0, shift1, shift2,
                                           P<T>(out), cpad);)
  return fsr_check_launch("image_to_nhwc_kernel");
}

extern "C" int fsr_u8_to_image(const uint8_t* frames, float* img, long long count, fsr_stream_t stream_) {
  if (!frames || !img || count <= 0) return fsr_fail(-1, "fsr_u8_to_image: bad argument");
  long long blocks = (count + 1023) / 1024;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(u8_to_image_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, frames, img, count);
  return fsr_check_launch("u8_to_image_kernel");
}

extern "C" size_t fsr_tanh_bwd_scratch(void) { return (size_t)512 * 64 * 3 * sizeof(float); }

extern "C" int fsr_tanh_bwd_to_nhwc(int dtype, const float* g, long long sn, long long sc, long long sh, long long sw,
                                    const float* y_nhwc3, int n, int h, int w, void* dz, int cpad, float* dbias,
                                    void* scratch, fsr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!g || !y_nhwc3 || !dz) return fsr_fail(-1, "fsr_tanh_bwd_to_nhwc: null argument");
  if (dbias && !scratch) return fsr_fail(-1, "fsr_tanh_bwd_to_nhwc: the bias gradient needs the scratch buffer");
  if (int rc = check_c("fsr_tanh_bwd_to_nhwc", dtype, cpad)) return rc;
  if (int rc = check_x3("fsr_tanh_bwd_to_nhwc", dtype, dz)) return rc;
  const int rowunits = w * (cpad / (dtype != FSR_F32 ? 8 : 4));
  const int gx = n * h < 512 ? n * h : 512, gy = row_blocks(rowunits);   // <= 512 x 64 workgroups
  FSR_DISPATCH_T(dtype, hipLaunchKernelGGL(tanh_bwd_to_nhwc_kernel<T>, dim3(gx, gy), dim3(256), 0,
                                           stream, g, sn, sc, sh, sw, y_nhwc3, h, w, P<T>(dz), cpad, n * h,
                                           dbias ? (float*)scratch : nullptr);)
  if (int rc = fsr_check_launch("tanh_bwd_to_nhwc_kernel")) return rc;
  if (dbias) return fsr_launch_reduce_partials((const float*)scratch, dbias, 1, gx * gy, 3, 3, 0, 0, 1.f, 0, stream);
  return 0;
}

extern "C" int fsr_tanh_bwd_image(const float* g, long long sn, long long sc, long long sh, long long sw, const float* y_nhwc3,
                                  int n, int h, int w, float* dz_nhwc3, float* dbias, void* scratch, fsr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!g || !y_nhwc3 || !dz_nhwc3) return fsr_fail(-1, "fsr_tanh_bwd_image: null argument");
  if (dbias && !scratch) return fsr_fail(-1, "fsr_tanh_bwd_image: the bias gradient needs the scratch buffer");
  if (n <= 0 || h <= 0 || w <= 0) return fsr_fail(-2, "fsr_tanh_bwd_image: bad dims");
  const int gx = n * h < 512 ? n * h : 512, gy = (w + 255) / 256 < 64 ? (w + 255) / 256 : 64;
  hipLaunchKernelGGL(tanh_bwd_image_kernel, dim3(gx, gy), dim3(256), 0, stream, g, sn, sc, sh, sw, y_nhwc3, h, w, dz_nhwc3, n * h,
                     dbias ? (float*)scratch : nullptr);
  if (int rc = fsr_check_launch("tanh_bwd_image_kernel")) return rc;
  if (dbias) return fsr_launch_reduce_partials((const float*)scratch, dbias, 1, gx * gy, 3, 3, 0, 0, 1.f, 0, stream);
  return 0;
}

extern "C" int fsr_add(int dtype, const void* a, const void* b, void* out, long long count, fsr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!a || !b || !out || count <= 0) return fsr_fail(-1, "fsr_add: bad argument");
  if (dtype != FSR_F32 && dtype != FSR_BF16 && dtype != FSR_F16 && dtype != FSR_X3) return fsr_fail(-2, "fsr_add: unknown dtype %d", dtype);
  const int e = dtype == FSR_X3 ? 32 : (dtype != FSR_F32 ? 8 : 4);
  if (count % e) return fsr_fail(-2, "fsr_add: count %lld is not a multiple of %d", count, e);
  if (int rc = check_x3("fsr_add", dtype, a, b, out)) return rc;
  const long long units = count / (dtype != FSR_F32 ? 8 : 4);
  long long blocks = (units + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  FSR_DISPATCH_T(dtype, hipLaunchKernelGGL(add_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, stream, P<T>(a), P<T>(b), P<T>(out), units);)
  return fsr_check_launch("add_kernel");
}

extern "C" int fsr_maxpool2_fwd(int dtype, const void* x, void* y, void* argmax, int n, int h, int w, int c, fsr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !y) return fsr_fail(-1, "fsr_maxpool2_fwd: null argument");
  if (int rc = check_c("fsr_maxpool2_fwd", dtype, c)) return rc;
  if (int rc = check_x3("fsr_maxpool2_fwd", dtype, x, y)) return rc;
  if ((h | w) & 1) return fsr_fail(-2, "fsr_maxpool2_fwd: odd extent %dx%d", h, w);
  const int rowunits = (w / 2) * (c / (dtype != FSR_F32 ? 8 : 4));
  FSR_DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool2_fwd_kernel<T>, dim3(n * (h / 2), row_blocks(rowunits)), dim3(256), 0,
                                           stream, P<T>(x), P<T>(y), (unsigned char*)argmax, h, w, c);)
  return fsr_check_launch("maxpool2_fwd_kernel");
}

extern "C" int fsr_maxpool2_bwd_argmax(int dtype, const void* g, const void* argmax, void* dx, int n, int h, int w, int c,
                                       int relu_mask, fsr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!g || !argmax || !dx) return fsr_fail(-1, "fsr_maxpool2_bwd_argmax: null argument");
  if (int rc = check_c("fsr_maxpool2_bwd_argmax", dtype, c)) return rc;
  if (int rc = check_x3("fsr_maxpool2_bwd_argmax", dtype, g, dx)) return rc;
  if ((h | w) & 1) return fsr_fail(-2, "fsr_maxpool2_bwd_argmax: odd extent %dx%d", h, w);
  const int rowunits = (w / 2) * (c / (dtype != FSR_F32 ? 8 : 4));
  FSR_DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool2_bwd_idx_kernel<T>, dim3(n * (h / 2), row_blocks(rowunits)), dim3(256), 0,
                                           stream, P<T>(g), (const unsigned char*)argmax, P<T>(dx), h, w, c, relu_mask);)
  return fsr_check_launch("maxpool2_bwd_idx_kernel");
}

extern "C" int fsr_maxpool2_bwd(int dtype, const void* g, const void* x, const void* y, void* dx, int n, int h, int w,
                                int c, int relu_mask, fsr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!g || !x || !y || !dx) return fsr_fail(-1, "fsr_maxpool2_bwd: null argument");
  if (int rc = check_c("fsr_maxpool2_bwd", dtype, c)) return rc;
  if (int rc = check_x3("fsr_maxpool2_bwd", dtype, g, x, y, dx)) return rc;
  if ((h | w) & 1) return fsr_fail(-2, "fsr_maxpool2_bwd: odd extent %dx%d", h, w);
  const int rowunits = (w / 2) * (c / (dtype != FSR_F32 ? 8 : 4));
  FSR_DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool2_bwd_kernel<T>, dim3(n * (h / 2), row_blocks(rowunits)), dim3(256), 0,
                                           stream, P<T>(g), P<T>(x), P<T>(y), P<T>(dx), h, w, c, relu_mask);)
  return fsr_check_launch("maxpool2_bwd_kernel");
}
