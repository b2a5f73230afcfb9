// Fixed-order second level of every cross-workgroup reduction of the hot path.
//
// The kernels that reduce over pixels (InstanceNorm statistics in the conv epilogues, the InstanceNorm / activation
// backward sums, bias and PReLU-slope gradients, the loss means, the 1x1 output convolution's weight gradient) write ONE
// partial vector per workgroup into a scratch buffer with plain stores; the kernels below -- enqueued by the same C-ABI
// call right behind the producer, so the kernel boundary is the only synchronisation -- add the partials of a result in
// a fixed order.  No float atomics anywhere: two runs on the same inputs give the same bits, whatever the dispatch order
// (SURVEY.md section 7 "deterministic two-level reduce"; torch.nn.InstanceNorm2d itself, model.py:55,65,94,132, is
// deterministic on the reference's CPU path).
#include "fsr_common.h"
#include "fsr_host.h"

namespace {

// out[b][i] (+)= scale * sum_{p < cnt(b)} part[(b * P + p) * stride + i],  i < len <= stride
// Block = 32 consecutive outputs x 8 part lanes; lane j adds parts j, j+8, ... in order, thread j == 0 then adds the eight
// lane sums in order.  per > 0: batch b (an image) only owns the first cnt(b) = last_w - first_w + 1 slots, where
// first_w / last_w are the first / last tile range [w * per, (w+1) * per) that intersects the image's tiles
// [b * tpi, (b+1) * tpi) -- the slot rule of the persistent convolution kernels (conv64_persistent.hip).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, float* __restrict__ out, int P,
                                                              int len, int stride, int tpi, int per, float scale,
                                                              int accumulate) {
  __shared__ float red[8][32];
  const int b = blockIdx.y;
  const int oi = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + oi;
  int cnt = P;
  if (per > 0) {
    const int first_w = (b * tpi) / per, last_w = ((b + 1) * tpi - 1) / per;
    cnt = last_w - first_w + 1;
    if (cnt > P) cnt = P;
  }
  float s = 0.f;
  if (i < len) {
    const float* p = part + ((size_t)b * P) * stride + i;
    int k = pl;
    for (; k + 24 < cnt; k += 32) {   // four loads in flight, added in order
      const float v0 = p[(size_t)k * stride], v1 = p[(size_t)(k + 8) * stride];
      const float v2 = p[(size_t)(k + 16) * stride], v3 = p[(size_t)(k + 24) * stride];
      s = ((s + v0) + v1) + v2 + v3;
    }
    for (; k < cnt; k += 8) s += p[(size_t)k * stride];
  }
  red[pl][oi] = s;
  __syncthreads();
  if (pl == 0 && i < len) {
    float t = red[0][oi];
#pragma unroll
    for (int j = 1; j < 8; ++j) t += red[j][oi];
    t *= scale;
    float* o = out + (size_t)b * len + i;
    *o = accumulate ? *o + t : t;
  }
}

}  // namespace

int fsr_launch_reduce_partials(const float* part, float* out, int batches, int P, int len, int stride, int tpi, int per,
                               float scale, int accumulate, hipStream_t stream) {
  if (batches <= 0 || P <= 0 || len <= 0 || stride < len) return fsr_fail(-2, "reduce_partials: bad extents");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((len + 31) / 32, batches), dim3(256), 0, stream, part, out, P, len, stride,
                     tpi, per, scale, accumulate);
  return fsr_check_launch("reduce_partials_kernel");
}
