// Validation metrics on the device: SSIM and the squared error behind PSNR in ONE pass over the two images.
//
// Replaces torchmetrics' StructuralSimilarityIndexMeasure(data_range=1.0, reduction="none") and
// PeakSignalNoiseRatio(data_range=1.0, reduction="none") as /root/reference/trainer.py:46-51 builds them and
// :53-69 feeds them with (1 + G(lr)) / 2 and (1 + hr) / 2.  torchmetrics (pinned 1.4.0, Pipfile) is not a dependency;
// its published algorithm, restated:
//   SSIM : gaussian window 11x11 (size int(3.5 sigma + .5) * 2 + 1), sigma 1.5, k1 .01, k2 .03; the images are
//          reflect-padded by 5, the five moments E[a], E[b], E[aa], E[bb], E[ab] are depthwise convolutions with the
//          window, var = max(E[xx] - E[x]^2, 0), cov = E[ab] - E[a]E[b],
//          ssim = (2 E[a]E[b] + c1)(2 cov + c2) / ((E[a]^2 + E[b]^2 + c1)(var_a + var_b + c2)),
//          and the map is CROPPED by 5 on every side before the per-image mean -- i.e. only windows that lie entirely
//          inside the image count, so the padding never contributes;
//   PSNR : 10 log10(1 / (sum of squared errors / number of elements)) over everything the metric has seen.
// HBM-bound: both images are read once (24 B per pixel of 3 float channels).  The window is separable: a workgroup
// stages a 42x42 patch of one channel of both images in LDS, runs the 11-tap horizontal pass for the five moments into
// LDS, then the vertical pass for its 32x32 output pixels.  Tile sums go to per-workgroup partial slots and are added
// in a fixed order (reduce.hip): bit-reproducible.
#include "fsr_common.h"
#include "fsr_host.h"

#include <math.h>

namespace {

constexpr int KW = 11, PADW = 5, TILE = 32, REG = TILE + KW - 1;   // 42

struct GaussTaps {
  float g[KW];
};

struct MetricArgs {
  const float* a;
  const float* b;
  long long asn, asc, ash, asw;   // element strides of a (n, c, h, w)
  long long bsn, bsc, bsh, bsw;
  int h, w, tiles_x, tiles_y;
  float* part;                    // [n][3 * tiles][2]  (ssim sum, squared-error sum)
  float c1, c2;
  GaussTaps taps;
};

__global__ __launch_bounds__(256) void ssim_sse_kernel(const MetricArgs p) {
  __shared__ float A[REG][REG + 1], B[REG][REG + 1];
  __shared__ float Hm[5][REG][TILE + 1];
  __shared__ float red[2][4];
  const int tid = threadIdx.x;
  const int tx = blockIdx.x % p.tiles_x, ty = blockIdx.x / p.tiles_x;
  const int ch = blockIdx.y, n = blockIdx.z;
  const int y0 = ty * TILE, x0 = tx * TILE;
  const float* pa = p.a + n * p.asn + ch * p.asc;
  const float* pb = p.b + n * p.bsn + ch * p.bsc;
  // the last tile row / column also owns the image's last rows / columns for the squared error
  const int own_y1 = (ty == p.tiles_y - 1) ? p.h : (y0 + TILE < p.h ? y0 + TILE : p.h);
  const int own_x1 = (tx == p.tiles_x - 1) ? p.w : (x0 + TILE < p.w ? x0 + TILE : p.w);
  float sse = 0.f;
  for (int i = tid; i < REG * REG; i += 256) {
    const int r = i / REG, c = i - r * REG;
    const int y = y0 + r, x = x0 + c;
    float va = 0.f, vb = 0.f;
    if (y < p.h && x < p.w) {
      va = (1.f + pa[y * p.ash + x * p.asw]) * 0.5f;      // trainer.py:63-65
      vb = (1.f + pb[y * p.bsh + x * p.bsw]) * 0.5f;
      if (y < own_y1 && x < own_x1) {
        const float d = va - vb;
        sse += d * d;
      }
    }
    A[r][c] = va;
    B[r][c] = vb;
  }
  __syncthreads();
  for (int i = tid; i < REG * TILE; i += 256) {
    const int r = i / TILE, c = i - r * TILE;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const float g = p.taps.g[k], va = A[r][c + k], vb = B[r][c + k];
      m0 += g * va;
      m1 += g * vb;
      m2 += g * (va * va);
      m3 += g * (vb * vb);
      m4 += g * (va * vb);
    }
    Hm[0][r][c] = m0;
    Hm[1][r][c] = m1;
    Hm[2][r][c] = m2;
    Hm[3][r][c] = m3;
    Hm[4][r][c] = m4;
  }
  __syncthreads();
  float ssim = 0.f;
  const int vh = p.h - 2 * PADW, vw = p.w - 2 * PADW;     // windows entirely inside the image
  for (int i = tid; i < TILE * TILE; i += 256) {
    const int r = i / TILE, c = i - r * TILE;
    if (y0 + r < vh && x0 + c < vw) {
      float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        const float g = p.taps.g[k];
#pragma unroll
        for (int q = 0; q < 5; ++q) m[q] += g * Hm[q][r + k][c];
      }
      const float mu_aa = m[0] * m[0], mu_bb = m[1] * m[1], mu_ab = m[0] * m[1];
      const float var_a = fmaxf(m[2] - mu_aa, 0.f), var_b = fmaxf(m[3] - mu_bb, 0.f), cov = m[4] - mu_ab;
      ssim += ((2.f * mu_ab + p.c1) * (2.f * cov + p.c2)) / ((mu_aa + mu_bb + p.c1) * (var_a + var_b + p.c2));
    }
  }
  ssim = wave_sum(ssim);
  sse = wave_sum(sse);
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = ssim;
    red[1][tid >> 6] = sse;
  }
  __syncthreads();
  if (tid < 2) {
    const int ntile = p.tiles_x * p.tiles_y;
    const size_t slot = (size_t)n * 3 * ntile + (size_t)ch * ntile + blockIdx.x;
    p.part[slot * 2 + tid] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
  }
}

int metric_tiles(int h, int w, int* tx, int* ty) {
  const int vh = h - 2 * PADW, vw = w - 2 * PADW;
  if (vh <= 0 || vw <= 0) return 0;
  *tx = (vw + TILE - 1) / TILE;
  *ty = (vh + TILE - 1) / TILE;
  return *tx * *ty;
}

}  // namespace

extern "C" size_t fsr_ssim_sse_scratch(int n, int h, int w) {
  int tx, ty;
  const int t = metric_tiles(h, w, &tx, &ty);
  return (n > 0 && t > 0) ? (size_t)n * 3 * t * 2 * sizeof(float) : 0;
}

extern "C" int fsr_ssim_sse(const float* a, long long asn, long long asc, long long ash, long long asw, const float* b,
                            long long bsn, long long bsc, long long bsh, long long bsw, int n, int h, int w, float* out,
                            void* scratch, fsr_stream_t stream_) {
  if (!a || !b || !out || !scratch) return fsr_fail(-1, "fsr_ssim_sse: null argument");
  if (n <= 0) return fsr_fail(-2, "fsr_ssim_sse: empty batch");
  MetricArgs p;
  p.tiles_x = p.tiles_y = 0;
  const int ntile = metric_tiles(h, w, &p.tiles_x, &p.tiles_y);
  if (ntile <= 0) return fsr_fail(-2, "fsr_ssim_sse: images must be larger than the 11x11 window (got %dx%d)", h, w);
  if (n > 65535) return fsr_fail(-2, "fsr_ssim_sse: batch too large");
  p.a = a; p.asn = asn; p.asc = asc; p.ash = ash; p.asw = asw;
  p.b = b; p.bsn = bsn; p.bsc = bsc; p.bsh = bsh; p.bsw = bsw;
  p.h = h; p.w = w;
  p.part = (float*)scratch;
  p.c1 = 0.01f * 0.01f;   // (k1 * data_range)^2, data_range 1.0 (trainer.py:46-48)
  p.c2 = 0.03f * 0.03f;
  float s = 0.f;
  for (int k = 0; k < KW; ++k) {          // torchmetrics _gaussian: exp(-(d / sigma)^2 / 2), normalised, float32
    const float d = (float)(k - PADW) / 1.5f;
    p.taps.g[k] = expf(-(d * d) / 2.f);
    s += p.taps.g[k];
  }
  for (int k = 0; k < KW; ++k) p.taps.g[k] /= s;
  hipLaunchKernelGGL(ssim_sse_kernel, dim3(ntile, 3, n), dim3(256), 0, (hipStream_t)stream_, p);
  if (int rc = fsr_check_launch("ssim_sse_kernel")) return rc;
  // out[n][2] = (sum of the ssim map over channels and valid pixels, sum of squared errors over the whole image)
  return fsr_launch_reduce_partials((const float*)scratch, out, n, 3 * ntile, 2, 2, 0, 0, 1.f, 0, (hipStream_t)stream_);
}
