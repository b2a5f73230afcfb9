// Loss reductions, the discriminator's 1x1 output convolution and the fused AdamW step.
//
//   BCEWithLogitsLoss (mean)        /root/reference/trainer.py:41,177,178,188
//   SmoothL1Loss (beta 1, mean)     /root/reference/trainer.py:43,192 (VGG features), :109 (pixels)
//   Conv2d(512 -> 1, k=1)           /root/reference/model.py:184-186
//   AdamW(lr, fused=True)           /root/reference/trainer.py:33-38
//
// Reductions are wavefront-level: each lane accumulates a grid-stride partial, the 64 lanes are
// summed with DPP/shuffle steps, the 4 waves of a workgroup meet in LDS and the workgroup's sum goes to
// its slot of a scratch buffer; reduce.hip adds the slots in a fixed order (bit-reproducible losses and
// gradients, no float atomics).
#include "fsr_common.h"
#include "fsr_host.h"

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* red4) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
  __syncthreads();
  return red4[0] + red4[1] + red4[2] + red4[3];
}

inline int red_blocks(long long items) {
  long long b = (items + 256 * 8 - 1) / (256 * 8);
  if (b > 1024) b = 1024;
  return b < 1 ? 1 : (int)b;
}

__global__ __launch_bounds__(256) void bce_logits_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                             float* __restrict__ loss, long long count) {
  __shared__ float red4[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
    const float xv = x[i];
    s += fmaxf(xv, 0.f) - xv * t[i] + log1pf(expf(-fabsf(xv)));
  }
  s = block_sum_256(s, red4);
  if (threadIdx.x == 0) loss[blockIdx.x] = s;   // partial slot; the mean is finished by reduce.hip
}

__global__ __launch_bounds__(256) void bce_logits_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                             const float* __restrict__ gscale, float* __restrict__ dx,
                                                             long long count, float inv) {
  const float gs = gscale[0] * inv;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
    const float xv = x[i];
    const float sg = 1.f / (1.f + expf(-xv));
    dx[i] = gs * (sg - t[i]);
  }
}

template <typename T> struct LV : V16<T, 0> {};   // 16-byte unit loader (fsr_common.h)
template <typename T>
__global__ __launch_bounds__(256) void smooth_l1_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                            float* __restrict__ loss, long long units) {
  constexpr int E = LV<T>::N;
  __shared__ float red4[4];
  float s = 0.f;
  for (long long u = (long long)blockIdx.x * 256 + threadIdx.x; u < units; u += (long long)gridDim.x * 256) {
    float av[E], bv[E];
    LV<T>::ld(a + u * E, av);
    LV<T>::ld(b + u * E, bv);
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const float d = fabsf(av[i] - bv[i]);
      s += d < 1.f ? 0.5f * d * d : d - 0.5f;
    }
  }
  s = block_sum_256(s, red4);
  if (threadIdx.x == 0) loss[blockIdx.x] = s;   // partial slot; the mean is finished by reduce.hip
}

template <typename T>
__global__ __launch_bounds__(256) void smooth_l1_bwd_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                            const float* __restrict__ gscale, T* __restrict__ da,
                                                            long long units, float inv) {
  constexpr int E = LV<T>::N;
  const float gs = gscale[0] * inv;
  for (long long u = (long long)blockIdx.x * 256 + threadIdx.x; u < units; u += (long long)gridDim.x * 256) {
    float av[E], bv[E];
    LV<T>::ld(a + u * E, av);
    LV<T>::ld(b + u * E, bv);
#pragma unroll
    for (int i = 0; i < E; ++i) av[i] = gs * fminf(fmaxf(av[i] - bv[i], -1.f), 1.f);
    LV<T>::st(da + u * E, av);
  }
}

// one wave per pixel: lanes stride over the channel units, shuffle-reduce the dot product
template <typename T>
__global__ __launch_bounds__(256) void conv1x1_c1_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ logits,
                                                             int npix, int c) {
  constexpr int E = LV<T>::N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cu = c / E;
  for (int p = blockIdx.x * 4 + wave; p < npix; p += gridDim.x * 4) {
    float s = 0.f;
    for (int u = lane; u < cu; u += 64) {
      float xv[E];
      LV<T>::ld(x + (size_t)p * c + u * E, xv);
#pragma unroll
      for (int i = 0; i < E; ++i) s += xv[i] * w[u * E + i];
    }
    s = wave_sum(s);
    if (lane == 0) logits[p] = s + b[0];
  }
}

// dx[p][c] = g[p]*w[c]; dw[c] += sum_p g[p]*x[p][c]; db += sum_p g[p].  Thread = channel unit x pixel row.
template <typename T>
__global__ __launch_bounds__(256) void conv1x1_c1_bwd_kernel(const float* __restrict__ g, const T* __restrict__ x,
                                                             const float* __restrict__ w, T* __restrict__ dx,
                                                             float* __restrict__ dw, int npix, int c) {
  constexpr int E = LV<T>::N;
  __shared__ float red[256 * E];
  const int cu = c / E, rows = 256 / cu;
  const int tid = threadIdx.x, unit = tid % cu, row = tid / cu;
  float acc[E], wv[E];
  float gsum = 0.f;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    acc[i] = 0.f;
    wv[i] = w[unit * E + i];
  }
  for (int p = blockIdx.x * rows + row; p < npix; p += gridDim.x * rows) {
    const float gp = g[p];
    float xv[E], o[E];
    LV<T>::ld(x + (size_t)p * c + unit * E, xv);
#pragma unroll
    for (int i = 0; i < E; ++i) {
      acc[i] += gp * xv[i];
      o[i] = gp * wv[i];
    }
    if (dx) LV<T>::st(dx + (size_t)p * c + unit * E, o);
    if (unit == 0) gsum += gp;
  }
#pragma unroll
  for (int i = 0; i < E; ++i) red[i * 256 + tid] = acc[i];
  __syncthreads();
  for (int t = tid; t < cu * E; t += 256) {
    const int q = t / cu, un = t % cu;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += red[q * 256 + r * cu + un];
    dw[(size_t)blockIdx.x * (c + 1) + un * E + q] = s;   // partial slot [block][c + 1] (reduce.hip adds the blocks in order)
  }
  __syncthreads();
  red[tid] = gsum;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += red[r * cu];
    dw[(size_t)blockIdx.x * (c + 1) + c] = s;
  }
}

// The step count lives in device memory (a float) so that a captured hipGraph replays correctly: a tick kernel
// increments it, the update kernel derives the bias corrections from it.
__global__ void adamw_tick_kernel(float* __restrict__ step) {
  if (threadIdx.x == 0) step[0] += 1.f;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long count, float lr, float b1, float b2,
                                                    float eps, float wd, const float* __restrict__ step, float gscale) {
  const float t = step[0];
  const float bc1 = 1.f - powf(b1, t);
  const float rsqrt_bc2 = 1.f / sqrtf(1.f - powf(b2, t));
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
    const float gi = g[i] * gscale;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);   // torch: exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
  }
}

// ---- dynamic loss scaling (fp16 mode).  state: float[4] on the device = {scale, clean steps, non-finite flag, skipped steps}.
// Everything is decided on the device, so a captured hipGraph keeps adapting while it replays.
__global__ __launch_bounds__(256) void grad_nonfinite_kernel(const float* __restrict__ g, long long count, float* __restrict__ state) {
  bool bad = false;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
    const float x = g[i];
    bad |= !(fabsf(x) <= 3.0e38f);   // inf or NaN
  }
  if (bad) state[2] = 1.f;           // every writer stores the same value: no atomic needed
}

__global__ __launch_bounds__(256) void adamw_scaled_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, long long count, float lr, float b1, float b2,
                                                           float eps, float wd, float* __restrict__ step, float gscale,
                                                           const float* __restrict__ state) {
  if (state[2] != 0.f) return;       // a gradient overflowed somewhere: no update at all, moments and step count untouched
  const float t = step[0] + 1.f;     // (the step counter is advanced by adamw_tick_if_clean_kernel, enqueued behind this kernel)
  const float gs = gscale / state[0];
  const float bc1 = 1.f - powf(b1, t);
  const float rsqrt_bc2 = 1.f / sqrtf(1.f - powf(b2, t));
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
    const float gi = g[i] * gs;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
  }
}
__global__ void adamw_tick_if_clean_kernel(float* __restrict__ step, const float* __restrict__ state) {
  if (threadIdx.x == 0 && state[2] == 0.f) step[0] += 1.f;
}
__global__ void loss_scale_update_kernel(float* __restrict__ state, float growth_interval, float growth, float backoff) {
  if (threadIdx.x != 0) return;
  if (state[2] != 0.f) {             // overflow in this iteration: back off, start counting again
    state[0] = fmaxf(state[0] * backoff, 1.f);
    state[1] = 0.f;
    state[3] += 1.f;
    state[2] = 0.f;
  } else {
    state[1] += 1.f;
    if (state[1] >= growth_interval) {
      state[0] = fminf(state[0] * growth, 16777216.f);
      state[1] = 0.f;
    }
  }
}

}  // namespace

extern "C" int fsr_grad_nonfinite(const float* g, long long count, float* scale_state, fsr_stream_t stream_) {
  if (!g || !scale_state || count <= 0) return fsr_fail(-1, "fsr_grad_nonfinite: bad argument");
  long long blocks = (count + 256 * 8 - 1) / (256 * 8);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(grad_nonfinite_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, g, count, scale_state);
  return fsr_check_launch("grad_nonfinite_kernel");
}

extern "C" int fsr_adamw_step_scaled(float* p, const float* g, float* m, float* v, long long count, float lr, float beta1,
                                     float beta2, float eps, float weight_decay, float* step_counter, float grad_scale,
                                     const float* scale_state, fsr_stream_t stream_) {
  if (!p || !g || !m || !v || !step_counter || !scale_state || count <= 0) return fsr_fail(-1, "fsr_adamw_step_scaled: bad argument");
  long long blocks = (count + 256 * 4 - 1) / (256 * 4);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adamw_scaled_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, p, g, m, v, count, lr, beta1,
                     beta2, eps, weight_decay, step_counter, grad_scale, scale_state);
  hipLaunchKernelGGL(adamw_tick_if_clean_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, step_counter, scale_state);
  return fsr_check_launch("adamw_scaled_kernel");
}

extern "C" int fsr_loss_scale_update(float* scale_state, float growth_interval, float growth, float backoff, fsr_stream_t stream_) {
  if (!scale_state || !(growth_interval >= 1.f) || !(growth >= 1.f) || !(backoff > 0.f && backoff <= 1.f))
    return fsr_fail(-1, "fsr_loss_scale_update: bad argument");
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, scale_state, growth_interval, growth, backoff);
  return fsr_check_launch("loss_scale_update_kernel");
}

extern "C" size_t fsr_loss_scratch(void) { return 1024 * sizeof(float); }   // red_blocks() caps the grid at 1024 workgroups

extern "C" int fsr_bce_logits_fwd(const float* x, const float* t, float* loss, void* scratch, long long count,
                                  fsr_stream_t stream_) {
  if (!x || !t || !loss || !scratch || count <= 0) return fsr_fail(-1, "fsr_bce_logits_fwd: bad argument");
  const int blocks = red_blocks(count);
  hipLaunchKernelGGL(bce_logits_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, x, t, (float*)scratch, count);
  if (int rc = fsr_check_launch("bce_logits_fwd_kernel")) return rc;
  return fsr_launch_reduce_partials((const float*)scratch, loss, 1, blocks, 1, 1, 0, 0, 1.f / (float)count, 0, (hipStream_t)stream_);
}
extern "C" int fsr_bce_logits_bwd(const float* x, const float* t, const float* gscale, float* dx, long long count,
                                  fsr_stream_t stream_) {
  if (!x || !t || !gscale || !dx || count <= 0) return fsr_fail(-1, "fsr_bce_logits_bwd: bad argument");
  hipLaunchKernelGGL(bce_logits_bwd_kernel, dim3(red_blocks(count)), dim3(256), 0, (hipStream_t)stream_, x, t, gscale, dx,
                     count, 1.f / (float)count);
  return fsr_check_launch("bce_logits_bwd_kernel");
}

extern "C" int fsr_smooth_l1_fwd(int dtype, const void* a, const void* b, float* loss, void* scratch, long long count,
                                 fsr_stream_t stream_) {
  if (!a || !b || !loss || !scratch || count <= 0) return fsr_fail(-1, "fsr_smooth_l1_fwd: bad argument");
  const int e = dtype != FSR_F32 ? 8 : 4;
  if (count % (dtype == FSR_X3 ? 32 : e)) return fsr_fail(-2, "fsr_smooth_l1_fwd: count %lld is not a multiple of %d", count, dtype == FSR_X3 ? 32 : e);
  if (dtype == FSR_X3 && ((((size_t)a | (size_t)b) & 127) != 0)) return fsr_fail(-2, "fsr_smooth_l1_fwd: x3 tensors must be 128-byte aligned");
  const long long units = count / e;
  const int blocks = red_blocks(units * 4);
  if (dtype == FSR_F16)
    hipLaunchKernelGGL(smooth_l1_fwd_kernel<f16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_,
                       (const f16_t*)a, (const f16_t*)b, (float*)scratch, units);
  else if (dtype == FSR_BF16)
    hipLaunchKernelGGL(smooth_l1_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_,
                       (const bf16_t*)a, (const bf16_t*)b, (float*)scratch, units);
  else if (dtype == FSR_X3)
    hipLaunchKernelGGL(smooth_l1_fwd_kernel<x3_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_,
                       (const x3_t*)a, (const x3_t*)b, (float*)scratch, units);
  else if (dtype == FSR_F32)
    hipLaunchKernelGGL(smooth_l1_fwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_,
                       (const float*)a, (const float*)b, (float*)scratch, units);
  else
    return fsr_fail(-2, "fsr_smooth_l1_fwd: unknown dtype %d", dtype);
  if (int rc = fsr_check_launch("smooth_l1_fwd_kernel")) return rc;
  return fsr_launch_reduce_partials((const float*)scratch, loss, 1, blocks, 1, 1, 0, 0, 1.f / (float)count, 0, (hipStream_t)stream_);
}
extern "C" int fsr_smooth_l1_bwd(int dtype, const void* a, const void* b, const float* gscale, void* da, long long count,
                                 fsr_stream_t stream_) {
  if (!a || !b || !gscale || !da || count <= 0) return fsr_fail(-1, "fsr_smooth_l1_bwd: bad argument");
  const int e = dtype != FSR_F32 ? 8 : 4;
  if (count % (dtype == FSR_X3 ? 32 : e)) return fsr_fail(-2, "fsr_smooth_l1_bwd: count %lld is not a multiple of %d", count, dtype == FSR_X3 ? 32 : e);
  if (dtype == FSR_X3 && ((((size_t)a | (size_t)b | (size_t)da) & 127) != 0)) return fsr_fail(-2, "fsr_smooth_l1_bwd: x3 tensors must be 128-byte aligned");
  const long long units = count / e;
  if (dtype == FSR_F16)
    hipLaunchKernelGGL(smooth_l1_bwd_kernel<f16_t>, dim3(red_blocks(units * 4)), dim3(256), 0, (hipStream_t)stream_,
                       (const f16_t*)a, (const f16_t*)b, gscale, (f16_t*)da, units, 1.f / (float)count);
  else if (dtype == FSR_BF16)
    hipLaunchKernelGGL(smooth_l1_bwd_kernel<bf16_t>, dim3(red_blocks(units * 4)), dim3(256), 0, (hipStream_t)stream_,
                       (const bf16_t*)a, (const bf16_t*)b, gscale, (bf16_t*)da, units, 1.f / (float)count);
  else if (dtype == FSR_X3)
    hipLaunchKernelGGL(smooth_l1_bwd_kernel<x3_t>, dim3(red_blocks(units * 4)), dim3(256), 0, (hipStream_t)stream_,
                       (const x3_t*)a, (const x3_t*)b, gscale, (x3_t*)da, units, 1.f / (float)count);
  else if (dtype == FSR_F32)
    hipLaunchKernelGGL(smooth_l1_bwd_kernel<float>, dim3(red_blocks(units * 4)), dim3(256), 0, (hipStream_t)stream_,
                       (const float*)a, (const float*)b, gscale, (float*)da, units, 1.f / (float)count);
  else
    return fsr_fail(-2, "fsr_smooth_l1_bwd: unknown dtype %d", dtype);
  return fsr_check_launch("smooth_l1_bwd_kernel");
}

static int c1_check(const char* what, int dtype, int c, const void* x, const void* dx = nullptr) {
  if (dtype != FSR_F32 && dtype != FSR_BF16 && dtype != FSR_F16 && dtype != FSR_X3) return fsr_fail(-2, "%s: unknown dtype %d", what, dtype);
  if (dtype == FSR_X3 && ((((size_t)x | (size_t)dx) & 127) != 0)) return fsr_fail(-2, "%s: x3 tensors must be 128-byte aligned", what);
  const int e = dtype == FSR_X3 ? 32 : (dtype != FSR_F32 ? 8 : 4);
  if (c <= 0 || c % e) return fsr_fail(-2, "%s: %d channels is not a multiple of %d", what, c, e);
  return 0;
}

extern "C" int fsr_conv1x1_c1_fwd(int dtype, const void* x, const float* w, const float* b, float* logits, int npix, int c,
                                  fsr_stream_t stream_) {
  if (!x || !w || !b || !logits || npix <= 0) return fsr_fail(-1, "fsr_conv1x1_c1_fwd: bad argument");
  if (int rc = c1_check("fsr_conv1x1_c1_fwd", dtype, c, x)) return rc;
  int blocks = (npix + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  if (dtype == FSR_F16)
    hipLaunchKernelGGL(conv1x1_c1_fwd_kernel<f16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, (const f16_t*)x, w,
                       b, logits, npix, c);
  else if (dtype == FSR_BF16)
    hipLaunchKernelGGL(conv1x1_c1_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, (const bf16_t*)x, w,
                       b, logits, npix, c);
  else if (dtype == FSR_X3)
    hipLaunchKernelGGL(conv1x1_c1_fwd_kernel<x3_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, (const x3_t*)x, w,
                       b, logits, npix, c);
  else
    hipLaunchKernelGGL(conv1x1_c1_fwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, (const float*)x, w, b,
                       logits, npix, c);
  return fsr_check_launch("conv1x1_c1_fwd_kernel");
}

extern "C" size_t fsr_conv1x1_c1_bwd_scratch(int c) { return c > 0 ? (size_t)512 * (c + 1) * sizeof(float) : 0; }

extern "C" int fsr_conv1x1_c1_bwd(int dtype, const float* g, const void* x, const float* w, void* dx, float* dw, float* db,
                                  void* scratch, int npix, int c, fsr_stream_t stream_) {
  if (!g || !x || !w || !dw || !db || !scratch || npix <= 0) return fsr_fail(-1, "fsr_conv1x1_c1_bwd: bad argument");
  if (int rc = c1_check("fsr_conv1x1_c1_bwd", dtype, c, x, dx)) return rc;
  const int cu = c / (dtype != FSR_F32 ? 8 : 4);
  if (cu > 256 || 256 % cu) return fsr_fail(-2, "fsr_conv1x1_c1_bwd: %d channels do not tile a 256-thread workgroup", c);
  const int rows = 256 / cu;
  int blocks = (npix + rows * 8 - 1) / (rows * 8);
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  float* part = (float*)scratch;   // [blocks][c + 1]: weight-gradient partials, then the bias-gradient partial
  if (dtype == FSR_F16)
    hipLaunchKernelGGL(conv1x1_c1_bwd_kernel<f16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, g, (const f16_t*)x,
                       w, (f16_t*)dx, part, npix, c);
  else if (dtype == FSR_BF16)
    hipLaunchKernelGGL(conv1x1_c1_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, g, (const bf16_t*)x,
                       w, (bf16_t*)dx, part, npix, c);
  else if (dtype == FSR_X3)
    hipLaunchKernelGGL(conv1x1_c1_bwd_kernel<x3_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, g, (const x3_t*)x,
                       w, (x3_t*)dx, part, npix, c);
  else
    hipLaunchKernelGGL(conv1x1_c1_bwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, g, (const float*)x, w,
                       (float*)dx, part, npix, c);
  if (int rc = fsr_check_launch("conv1x1_c1_bwd_kernel")) return rc;
  // dw[c] and db[0] from the same partial rows (row stride c + 1)
  if (int rc = fsr_launch_reduce_partials(part, dw, 1, blocks, c, c + 1, 0, 0, 1.f, 0, (hipStream_t)stream_)) return rc;
  return fsr_launch_reduce_partials(part + c, db, 1, blocks, 1, c + 1, 0, 0, 1.f, 0, (hipStream_t)stream_);
}

extern "C" int fsr_adamw_step(float* p, const float* g, float* m, float* v, long long count, float lr, float beta1,
                              float beta2, float eps, float weight_decay, float* step_counter, float grad_scale,
                              fsr_stream_t stream_) {
  if (!p || !g || !m || !v || !step_counter || count <= 0) return fsr_fail(-1, "fsr_adamw_step: bad argument");
  hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, step_counter);
  long long blocks = (count + 256 * 4 - 1) / (256 * 4);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, p, g, m, v, count, lr, beta1,
                     beta2, eps, weight_decay, step_counter, grad_scale);
  return fsr_check_launch("adamw_kernel");
}
