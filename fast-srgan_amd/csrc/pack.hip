// Filter repacking: torch OIHW float parameters -> [tap][rows_pad][K] in the compute dtype.
// See include/fsr_hip.h (fsr_pack_conv3x3) for the four layouts.
#include "fsr_common.h"
#include "fsr_host.h"

// X3 (FSR_X3, T = bf16_t): K counts PHYSICAL channels, 2 x the logical K; physical index kp holds the hi (bit 5 clear) or
// lo (bit 5 set) part of logical channel (kp >> 6) * 32 + (kp & 31) -- the chunk order of an x3 activation tensor.
template <typename T, bool X3 = false>
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, T* __restrict__ out, int cout, int cin, int mode,
                                    int rows, int rows_pad, int K, int Kreal) {
  const long long total = 9LL * rows_pad * K;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int kp = (int)(i % K);
    const int k = X3 ? (kp >> 6) * 32 + (kp & 31) : kp;
    const int row = (int)((i / K) % rows_pad);
    const int t = (int)(i / ((long long)K * rows_pad));
    float v = 0.f;
    if (row < rows && k < Kreal) {
      int co, ci;
      if (mode == FSR_PACK_FWD || mode == FSR_PACK_FWD_PS) {
        co = row;
        ci = k;
        if (mode == FSR_PACK_FWD_PS) {
          const int cps = cout >> 2;
          co = 4 * (row % cps) + row / cps;
        }
      } else {
        ci = row;
        co = k;
        if (mode == FSR_PACK_DGRAD_PS) {
          const int cps = cout >> 2;
          co = 4 * (k % cps) + k / cps;
        }
      }
      v = w[((size_t)co * cin + ci) * 9 + t];
    }
    if constexpr (X3) {
      const bf16_t hi = f2bf(v);
      out[i] = (kp & 32) ? f2bf(v - bf2f(hi)) : hi;
    } else {
      ElemIO<T>::st(out + i, v);
    }
  }
}

// Stage-contiguous form of the same four layouts (conv_tall3.hip, conv_s2d3.hip): element
//   (((((nb * nchunks + c) * 9 + slice) * B + R) * 4 + u) * 8 + e   holds   std[slice][nb * B + ch(R)][32 c + 8 (u ^ ((R >> 2) & 3)) + e]
// with ch(R) = (R & ~31) + 16 ((i >> 2) & 1) + (i & 3) + 4 (i >> 3), i = R & 31: the rows of a B-channel block in the kernels'
// LDS order (a lane's 16 accumulator registers = 16 consecutive channels) and the 16-byte units of a 32-channel chunk in
// their XOR-swizzled LDS positions -- the LDS image of one (block, chunk, slice) is B * 64 contiguous bytes of this pack.
// X3: K counts physical channels (chunks alternate hi / lo parts of a logical 32-channel group, as in pack_conv3x3_kernel).
template <typename T, bool X3 = false>
__global__ void pack_conv3x3_lin_kernel(const float* __restrict__ w, T* __restrict__ out, int cout, int cin, int mode,
                                        int rows, int K, int B) {
  const int nchunks = K >> 5;
  const long long total = 9LL * rows * K;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    long long i = idx;
    const int e = (int)(i & 7); i >>= 3;
    const int u = (int)(i & 3); i >>= 2;
    const int R = (int)(i % B); i /= B;
    const int t = (int)(i % 9); i /= 9;
    const int c = (int)(i % nchunks);
    const int nb = (int)(i / nchunks);
    const int ir = R & 31;
    const int row = nb * B + (R & ~31) + 16 * ((ir >> 2) & 1) + (ir & 3) + 4 * (ir >> 3);
    const int kp = 32 * c + 8 * (u ^ ((R >> 2) & 3)) + e;
    const int k = X3 ? (kp >> 6) * 32 + (kp & 31) : kp;
    int co, ci;
    if (mode == FSR_PACK_FWD || mode == FSR_PACK_FWD_PS) {
      co = row;
      ci = k;
      if (mode == FSR_PACK_FWD_PS) {
        const int cps = cout >> 2;
        co = 4 * (row % cps) + row / cps;
      }
    } else {
      ci = row;
      co = k;
      if (mode == FSR_PACK_DGRAD_PS) {
        const int cps = cout >> 2;
        co = 4 * (k % cps) + k / cps;
      }
    }
    const float v = w[((size_t)co * cin + ci) * 9 + t];
    if constexpr (X3) {
      const bf16_t hi = f2bf(v);
      out[idx] = (kp & 32) ? f2bf(v - bf2f(hi)) : hi;
    } else {
      ElemIO<T>::st(out + idx, v);
    }
  }
}

extern "C" int fsr_pack_conv3x3_lin(int dtype, int mode, const float* w_oihw, int cout, int cin, int block, void* packed,
                                    fsr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!w_oihw || !packed) return fsr_fail(-1, "fsr_pack_conv3x3_lin: null argument");
  if (mode < FSR_PACK_FWD || mode > FSR_PACK_DGRAD_PS) return fsr_fail(-2, "fsr_pack_conv3x3_lin: bad mode %d", mode);
  if (block != 64 && block != 128) return fsr_fail(-2, "fsr_pack_conv3x3_lin: block must be 64 or 128");
  if (dtype != FSR_BF16 && dtype != FSR_F16 && dtype != FSR_X3) return fsr_fail(-2, "fsr_pack_conv3x3_lin: 16-bit dtypes and x3 only");
  const bool fwd = (mode == FSR_PACK_FWD || mode == FSR_PACK_FWD_PS);
  const int rows = fwd ? cout : cin, K = (fwd ? cin : cout) * (dtype == FSR_X3 ? 2 : 1);
  if (dtype == FSR_X3 && K % 64 != 0) return fsr_fail(-2, "fsr_pack_conv3x3_lin: x3 packs need K %% 32 == 0");
  if (rows % block != 0 || K % 32 != 0) return fsr_fail(-2, "fsr_pack_conv3x3_lin: rows %d must be a multiple of the block, K %d of 32", rows, K);
  if ((mode == FSR_PACK_FWD_PS || mode == FSR_PACK_DGRAD_PS) && cout % 4 != 0)
    return fsr_fail(-2, "fsr_pack_conv3x3_lin: pixel-shuffle packing needs cout %% 4 == 0");
  const long long total = 9LL * rows * K;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (dtype == FSR_F16)
    hipLaunchKernelGGL(pack_conv3x3_lin_kernel<f16_t>, dim3(blocks), dim3(256), 0, stream, w_oihw, (f16_t*)packed, cout, cin, mode, rows, K, block);
  else if (dtype == FSR_X3)
    hipLaunchKernelGGL((pack_conv3x3_lin_kernel<bf16_t, true>), dim3(blocks), dim3(256), 0, stream, w_oihw, (bf16_t*)packed, cout, cin, mode, rows, K, block);
  else
    hipLaunchKernelGGL(pack_conv3x3_lin_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, w_oihw, (bf16_t*)packed, cout, cin, mode, rows, K, block);
  return fsr_check_launch("pack_conv3x3_lin_kernel");
}

extern "C" int fsr_pack_conv3x3(int dtype, int mode, const float* w_oihw, int cout, int cin, int k_pad, void* packed,
                                fsr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!w_oihw || !packed) return fsr_fail(-1, "fsr_pack_conv3x3: null argument");
  if (mode < FSR_PACK_FWD || mode > FSR_PACK_DGRAD_PS) return fsr_fail(-2, "fsr_pack_conv3x3: bad mode %d", mode);
  if ((mode == FSR_PACK_FWD_PS || mode == FSR_PACK_DGRAD_PS) && cout % 4 != 0)
    return fsr_fail(-2, "fsr_pack_conv3x3: pixel-shuffle packing needs cout %% 4 == 0");
  const bool fwd = (mode == FSR_PACK_FWD || mode == FSR_PACK_FWD_PS);
  const int rows = fwd ? cout : cin, Kreal = fwd ? cin : cout;
  if (k_pad < Kreal) return fsr_fail(-2, "fsr_pack_conv3x3: k_pad %d < K %d", k_pad, Kreal);
  if (dtype == FSR_X3 && k_pad % 32) return fsr_fail(-2, "fsr_pack_conv3x3: x3 packs need k_pad %% 32 == 0");
  const int K = dtype == FSR_X3 ? 2 * k_pad : k_pad;      // x3: physical channels, hi / lo chunks of 32
  const int rows_pad = (rows + 15) / 16 * 16;
  const long long total = 9LL * rows_pad * K;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (dtype == FSR_F16)
    hipLaunchKernelGGL(pack_conv3x3_kernel<f16_t>, dim3(blocks), dim3(256), 0, stream, w_oihw, (f16_t*)packed, cout,
                       cin, mode, rows, rows_pad, K, Kreal);
  else if (dtype == FSR_BF16)
    hipLaunchKernelGGL(pack_conv3x3_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, w_oihw, (bf16_t*)packed, cout,
                       cin, mode, rows, rows_pad, K, Kreal);
  else if (dtype == FSR_X3)
    hipLaunchKernelGGL((pack_conv3x3_kernel<bf16_t, true>), dim3(blocks), dim3(256), 0, stream, w_oihw, (bf16_t*)packed, cout,
                       cin, mode, rows, rows_pad, K, Kreal);
  else if (dtype == FSR_F32)
    hipLaunchKernelGGL(pack_conv3x3_kernel<float>, dim3(blocks), dim3(256), 0, stream, w_oihw, (float*)packed, cout,
                       cin, mode, rows, rows_pad, K, Kreal);
  else
    return fsr_fail(-2, "fsr_pack_conv3x3: unknown dtype %d", dtype);
  return fsr_check_launch("pack_conv3x3_kernel");
}
