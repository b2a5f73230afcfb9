/*
 * fsr_hip.h -- C ABI of libfsr_hip.so, the MI355X (gfx950) kernel library behind the
 * Fast-SRGAN drop-in modules (Generator / Discriminator / VGG19 / Trainer / NumpyImagesDataset).
 *
 * The reference (HasnainRaz/Fast-SRGAN) has no FFI layer: every hot-path operator is a stock
 * torch.nn call.  Each entry point below therefore cites the reference call site(s) whose
 * arithmetic it replaces (paths relative to /root/reference).  The binding a maintainer would
 * add on the reference side is a ctypes stub; see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed from the caller (a torch tensor's data_ptr());
 *     the library never allocates, frees or synchronises; calls only enqueue work on `stream`
 *     (a hipStream_t passed as void*), so they are legal inside hipGraph stream capture;
 *   - activations are NHWC; `dtype` is the storage type of activations and packed filters:
 *     FSR_F32 (exact-f32 MFMA, parity mode) or FSR_BF16 (bf16 MFMA, f32 accumulate);
 *     parameters, biases, statistics, gradients of parameters and losses are always float;
 *   - return value 0 = enqueued, < 0 = rejected (nothing enqueued); fsr_last_error() gives the
 *     reason for the calling thread.  The functions are thread-compatible.
 */
#ifndef FSR_HIP_H
#define FSR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSR_ABI_VERSION 1

enum { FSR_F32 = 0, FSR_BF16 = 1 };
enum { FSR_ACT_NONE = 0, FSR_ACT_RELU = 1, FSR_ACT_LEAKY = 2, FSR_ACT_PRELU = 3, FSR_ACT_TANH = 4 };
enum { FSR_CONV_FWD = 0, FSR_CONV_DGRAD = 1 };
enum { FSR_PACK_FWD = 0, FSR_PACK_FWD_PS = 1, FSR_PACK_DGRAD = 2, FSR_PACK_DGRAD_PS = 3 };
enum { FSR_C3_IN_PLAIN = 0, FSR_C3_IN_VGG_NORM = 1, FSR_C3_IN_TANH_BWD = 2 };

typedef void* fsr_stream_t; /* hipStream_t */

int fsr_version(void);
const char* fsr_last_error(void);
/* "name=<marketing name>;arch=<gcnArchName>;cus=<n>;hbm_bytes=<n>" of the current device. */
int fsr_device_info(char* buf, size_t buflen);

/* ------------------------------------------------------------------ filter packing
 * torch Conv2d weights are OIHW float (state_dict layout, model.py:30-35,47-64,...).  The
 * kernels consume [9 taps][rows_pad][K] in `dtype` with K contiguous:
 *   FSR_PACK_FWD       rows = cout, K = cin                  slice index = ky*3+kx
 *   FSR_PACK_FWD_PS    as FWD, rows permuted r = (co%4)*(cout/4) + co/4 so that the epilogue of
 *                      the PixelShuffle(2) convs (model.py:36,40) stores contiguous channels
 *   FSR_PACK_DGRAD     rows = cin,  K = cout                 (transposed filter for dL/dx)
 *   FSR_PACK_DGRAD_PS  as DGRAD with K permuted like FWD_PS rows
 * rows_pad = 16 when rows < 16 (head conv, model.py:103-108; first-layer data gradients), else
 * rows; padded rows are zero filled.  `packed` holds 9*rows_pad*K elements. */
int fsr_pack_conv3x3(int dtype, int mode, const float* w_oihw, int cout, int cin, void* packed,
                     fsr_stream_t stream);

/* ------------------------------------------------------------------ 3x3 convolution, pad 1
 * Forward:  torch.nn.Conv2d(k=3, p=1, stride 1|2) at model.py:47-64, 86-93, 30-35, 103-108,
 *           124-131, 148-183 and vgg19.features convs (model.py:8), fused with the bias add, an
 *           activation (ReLU model.py:8 / LeakyReLU model.py:133,145 / PReLU model.py:37 /
 *           Tanh model.py:109), PixelShuffle(2) (model.py:36) and the per-(n,c) sum / sum of
 *           squares InstanceNorm2d (model.py:55,65,94,132) needs.
 * Dgrad:    the same kernel on the transposed filter: dL/dx of those convolutions.
 *   mode FSR_CONV_FWD  : in [n,ih,iw,cin] -> out [n,oh,ow,cout], oh = (ih-1)/stride+1
 *   mode FSR_CONV_DGRAD: in = dL/dy [n,ih,iw,cin], out = dL/dx [n,oh,ow,cout] where
 *                        (ih,iw,cin) are the forward OUTPUT dims and (oh,ow,cout) the forward
 *                        INPUT dims; stride is the forward stride.
 * pixel_shuffle (FWD): out is [n,2oh,2ow,cout/4]; filters packed FSR_PACK_FWD_PS.
 * in_pixel_shuffled (DGRAD): in is [n,2ih,2iw,cin/4]; filters packed FSR_PACK_DGRAD_PS.
 * stats (optional): float [n][cout][2], must be zeroed by the caller; accumulates the sum and
 *   sum of squares of the pre-activation over pixels.
 * addend (optional): tensor shaped/typed like out, added before the activation. */
typedef struct fsr_conv_desc {
  int dtype;
  int mode;
  int n, ih, iw, cin;
  int oh, ow, cout;
  int stride;
  int act;
  float slope;
  int pixel_shuffle;
  int in_pixel_shuffled;
  int out_f32;
} fsr_conv_desc;

int fsr_conv3x3(const fsr_conv_desc* desc, const void* in, const void* packed_w, const float* bias,
                const float* prelu_weight, const void* addend, void* out, float* stats,
                fsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FSR_HIP_H */
