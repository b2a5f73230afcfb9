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
 *   - activations are NHWC; `dtype` is the storage type of activations, activation gradients and
 *     packed filters: FSR_F32 (exact-f32 MFMA, parity mode), FSR_BF16 (bf16 MFMA, f32 accumulate) or
 *     FSR_F16 (fp16 MFMA, f32 accumulate: BASELINE configs[4]); parameters, biases, statistics,
 *     parameter gradients and losses are float;
 *   - FSR_X3 ("split bf16", the fast mode INSIDE the reference's fp32 tolerance): an element is 4 bytes, the pair
 *     hi = bf16(v), lo = bf16(v - hi), and every product of a convolution is three bf16 MFMAs into one f32 accumulator,
 *     x_hi*w_hi + x_lo*w_hi + x_hi*w_lo (the dropped x_lo*w_lo term is <= 2^-16 of the product).  Storage: an NHWC tensor of
 *     C channels (C %% 32 == 0) holds, per pixel and per group of 32 channels, 64 bytes of hi[32] followed by 64 bytes of
 *     lo[32] -- the same bytes as float32 (the torch container IS a float32 tensor of the logical shape), and to the MFMA
 *     kernels a bf16 tensor of 2C channels whose 32-channel chunks alternate hi / lo.  Tensor bases are 128-byte aligned.
 *     Packed filters are bf16 [9][rows][2K] with the same hi / lo chunk order along K.  Everything that is float in the
 *     other modes (parameters, statistics, gradients of parameters, losses, 3-channel images) is float here too;
 *   - channel counts of NHWC activations are multiples of FSR_CPAD(dtype) = 16 (f32) / 32 (bf16, f16, x3);
 *     3-channel images are stored zero-padded to that width;
 *   - reductions over pixels (InstanceNorm statistics, backward sums, bias / PReLU-slope gradients, loss
 *     means) are two-level and ORDER-FIXED: the producing kernel stores one partial vector per workgroup into
 *     the caller's `scratch` buffer (size from the matching fsr_*_scratch query; contents are don't-care before
 *     and after the call, the buffer may be reused by the next call on the same stream), and a second
 *     kernel enqueued by the same call adds the partials in a fixed order and ASSIGNS the result.  There are
 *     no float atomics anywhere: the same inputs give the same bits on every run.  Only the weight gradients
 *     (`dw_oihw`, and the `dbias` of fsr_conv3x3_c3_wgrad) ACCUMULATE into their outputs ("+=", one thread
 *     per element): the caller zeroes them once per optimizer step;
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

#define FSR_ABI_VERSION 10

enum { FSR_F32 = 0, FSR_BF16 = 1, FSR_F16 = 2, FSR_X3 = 3 };
enum { FSR_ACT_NONE = 0, FSR_ACT_RELU = 1, FSR_ACT_LEAKY = 2, FSR_ACT_PRELU = 3, FSR_ACT_TANH = 4 };
enum { FSR_CONV_FWD = 0, FSR_CONV_DGRAD = 1 };
enum { FSR_PACK_FWD = 0, FSR_PACK_FWD_PS = 1, FSR_PACK_DGRAD = 2, FSR_PACK_DGRAD_PS = 3 };
enum { FSR_OUT_DTYPE = 0, FSR_OUT_F32 = 1, FSR_OUT_U8 = 2 };   /* fsr_conv_desc.out_f32 */

typedef void* fsr_stream_t; /* hipStream_t */

int fsr_version(void);
const char* fsr_last_error(void);
/* Name of the kernel configuration the calling thread's most recent fsr_conv3x3 dispatched (profiling aid). */
const char* fsr_last_kernel(void);
/* "name=<marketing name>;arch=<gcnArchName>;cus=<n>;hbm_bytes=<n>" of the current device. */
int fsr_device_info(char* buf, size_t buflen);

/* ------------------------------------------------------------------ filter packing
 * torch Conv2d weights are OIHW float (state_dict layout, model.py:30-35,47-64,...).  The
 * kernels consume [9 taps][rows_pad][K_pad] in `dtype` with K contiguous:
 *   FSR_PACK_FWD       rows = cout, K = cin                  slice index = ky*3+kx
 *   FSR_PACK_FWD_PS    as FWD, rows permuted r = (co%4)*(cout/4) + co/4 so that the epilogue of
 *                      the PixelShuffle(2) convs (model.py:36,40) stores contiguous channels
 *   FSR_PACK_DGRAD     rows = cin,  K = cout                 (transposed filter for dL/dx)
 *   FSR_PACK_DGRAD_PS  as DGRAD with K permuted like FWD_PS rows
 * rows_pad = rows rounded up to 16 (head conv, model.py:103-108; first-layer data gradients:
 * 3 -> 16); K_pad = `k_pad` >= K (3-channel inputs: K zero-padded to FSR_CPAD).  Padding is zero
 * filled.  `packed` holds 9*rows_pad*k_pad elements. */
int fsr_pack_conv3x3(int dtype, int mode, const float* w_oihw, int cout, int cin, int k_pad, void* packed,
                     fsr_stream_t stream);
/* The same four layouts, STAGE-CONTIGUOUS for the 128..512-channel kernels (16-bit dtypes; rows a multiple of `block` = 64 or
 * 128, K a multiple of 32, no padding): [rows / block][K / 32][9 slices][block rows][32 channels], the rows of a block in the
 * kernels' LDS order and the 16-byte units of a 32-channel chunk XOR-swizzled as they lie in LDS, so the LDS-DMA of one
 * (block, chunk, slice) reads block * 64 contiguous bytes (whole 128-byte lines) instead of half lines 2 K bytes apart.
 * Which launches take it: fsr_conv3x3_pack_block().  `packed` holds 9 * rows * K elements. */
int fsr_pack_conv3x3_lin(int dtype, int mode, const float* w_oihw, int cout, int cin, int block, void* packed,
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
 * cin is the (padded) channel count of `in` and the K_pad of the packed filter.
 * pixel_shuffle (FWD): out is [n,2oh,2ow,cout/4]; filters packed FSR_PACK_FWD_PS.
 * in_pixel_shuffled (DGRAD): in is [n,2ih,2iw,cin/4]; filters packed FSR_PACK_DGRAD_PS.
 * out_f32: FSR_OUT_DTYPE (0): `dtype` output; FSR_OUT_F32 (1): store float whatever `dtype` is (3-channel outputs:
 *   head images, image gradients); FSR_OUT_U8 (2, FSR_ACT_TANH heads, forward): store the uint8 HWC image
 *   (unsigned char)(((tanh(z) + 1) / 2) * 255) -- inference.py:53-56's post-processing with its truncating cast,
 *   out [n,oh,ow,cout] bytes (cout = 3: the finished RGB frame).
 * pool2 (inference / no-grad passes of vgg19.features, model.py:8,191): the kernel's epilogue takes the 2x2 maximum of the
 *   activated outputs and stores only the pooled tensor -- the full-resolution tensor, which only a backward pass would
 *   read, is never written.  oh, ow even; cout %% 16 == 0; no stats / preact / mask tensors.
 * stats (optional): float [n][cout][2]; receives the sum and the sum of squares of the pre-activation
 *   over pixels (needs `scratch` of fsr_conv3x3_scratch(desc) bytes; FWD and stride-1 DGRAD launches).
 * preact (optional): tensor like out; receives the pre-activation (training: PReLU backward
 *   needs its sign, which the output of a negative-slope PReLU does not reveal).
 * oscale (optional): float [cout] multiplied into the result before bias/activation (the
 *   1/(2 std) of VGG19.forward's normalisation, model.py:21-22, applied to input gradients).
 * dact_mask (optional): tensor like out; the result is multiplied by (mask > 0 ? 1 : dact_slope).  A
 *   data-gradient launch uses it to apply the ReLU / LeakyReLU backward of the layer that PRODUCED the
 *   forward input (mask = that layer's output), saving a separate elementwise pass.
 *   With desc.mask_is_addend the same tensor is ADDED to the result instead (before the activation): the data gradient of
 *   the first convolution of a ResidualBlock (model.py:67-69) takes the gradient of the block's skip connection there, so
 *   dL/dx = conv_dgrad(dz) + dL/d(skip) leaves the kernel in one piece and autograd has nothing to accumulate (round 2 ran
 *   nine at::add<bf16> kernels per iteration for this). */
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
  int pool2; /* FWD, 16-bit dtypes, no pixel shuffle: out is the MaxPool2d(2,2) of the activated result, [n,oh/2,ow/2,cout] */
  int mask_is_addend; /* 0: dact_mask gates the result by its sign; 1: it is ADDED to the result instead (see below); 2 (stride-2 data
                         gradients, 16-bit dtypes): it is the PACKED SIGN-BIT tensor [n][oh][ow][cout / 8] of the producing layer's output
                         (bit c & 7 of byte c >> 3 = output > 0: fsr_conv3x3_c3_fwd's `signs`), gating like 0 at a sixteenth of the bytes */
  int pack_lin; /* 0: packed_w is fsr_pack_conv3x3's layout; 64 / 128: fsr_pack_conv3x3_lin's with that block (must equal fsr_conv3x3_pack_block) */
} fsr_conv_desc;

/* Block size of the stage-contiguous filter pack (fsr_pack_conv3x3_lin) that the kernel fsr_conv3x3 would dispatch for `desc`
 * reads, or 0 if it reads the standard pack (always a valid choice: every kernel also accepts pack_lin = 0).  The optional
 * tensors of the call take part in the dispatch: say which ones will be given.  Nothing is launched.  < 0: error. */
enum { FSR_OPT_BIAS = 1, FSR_OPT_PRELU = 2, FSR_OPT_OSCALE = 4, FSR_OPT_MASK = 8, FSR_OPT_PREACT = 16, FSR_OPT_STATS = 32 };
int fsr_conv3x3_pack_block(const fsr_conv_desc* desc, int optional_tensors);

size_t fsr_conv3x3_scratch(const fsr_conv_desc* desc);
int fsr_conv3x3(const fsr_conv_desc* desc, const void* in, const void* packed_w, const float* bias,
                const float* prelu_weight, const float* oscale, const void* dact_mask, float dact_slope, void* out,
                void* preact, float* stats, void* scratch, fsr_stream_t stream);

/* Weight gradient of the same convolutions: dW[co][ci][ky][kx] (OIHW float, torch .grad layout)
 *   += sum_{n,y,x} dy[n,y,x,co] * x[n, y*stride+ky-1, x*stride+kx-1, ci]      (autograd of model.py convs)
 * x [n,ih,iw,cin_pad], dy [n,oh,ow,cout_pad] (or [n,2oh,2ow,cout/4] when dy_pixel_shuffled).
 * Only dW[:cout][:cin] is written (cin/cout may be smaller than the padded tensor widths).
 * `workspace` (float, fsr_conv3x3_wgrad_workspace() bytes) holds split-K partials; dw accumulates. */
typedef struct fsr_wgrad_desc {
  int dtype;
  int n, ih, iw, cin_pad, cin;
  int oh, ow, cout_pad, cout;
  int stride;
  int dy_pixel_shuffled;
} fsr_wgrad_desc;
size_t fsr_conv3x3_wgrad_workspace(const fsr_wgrad_desc* desc);
int fsr_conv3x3_wgrad(const fsr_wgrad_desc* desc, const void* x, const void* dy, float* dw_oihw, void* workspace,
                      fsr_stream_t stream);
/* The same weight gradient for `nlayers` (1..32) layers of ONE shape in one launch: layer l reads x[l], dy[l] and
 * accumulates into dw_oihw[l] (host arrays of device pointers, read during the call only).  For the generator's 64 -> 64
 * 3x3 blocks (model.py:62-80; 17 identical layers at trainer.py:121-129's backward): cout = cin = 64 unpadded.
 * The layers share the launch's workgroups, so the split-K partials (workspace:
 * fsr_conv3x3_wgrad_grouped_workspace(desc, nlayers) bytes) shrink by the group size and one reduce serves all layers.
 * Summation order is fixed by (shape, nlayers): bit-reproducible, but not bit-equal to nlayers separate calls. */
size_t fsr_conv3x3_wgrad_grouped_workspace(const fsr_wgrad_desc* desc, int nlayers);
int fsr_conv3x3_wgrad_grouped(const fsr_wgrad_desc* desc, int nlayers, const void* const* x, const void* const* dy,
                              float* const* dw_oihw, void* workspace, fsr_stream_t stream);

/* ------------------------------------------------------------------ InstanceNorm2d (+ activation, + residual)
 * torch.nn.InstanceNorm2d defaults (model.py:55,65,94,132: biased variance, eps 1e-5, no affine)
 * applied to the raw conv output using the statistics the conv epilogue accumulated, fused with
 * PReLU (model.py:56) / LeakyReLU (model.py:133) and the residual add (model.py:69,115):
 *   out = act((x - mean) * rstd) + res.          x, res, out: [n,hw,c] `dtype`; stats [n][c][2]. */
int fsr_instnorm_act_fwd(int dtype, const void* x, const float* stats, const void* res, int act, float slope,
                         const float* prelu_weight, void* out, int n, int hw, int c, fsr_stream_t stream);
/* Backward, phase 1: sums[n][c][2] = (sum gz, sum gz*xhat) with gz = g * act'(xhat);
 * dprelu[0] = sum g*min(xhat,0) (optional); scratch: fsr_instnorm_act_bwd_scratch(n,hw,c) bytes.
 * Phase 2: dx = rstd*(gz - mean(gz) - xhat*mean(gz*xhat)). */
size_t fsr_instnorm_act_bwd_scratch(int n, int hw, int c);
int fsr_instnorm_act_bwd_reduce(int dtype, const void* g, const void* x, const float* stats, int act, float slope,
                                const float* prelu_weight, float* sums, float* dprelu, void* scratch, int n, int hw,
                                int c, fsr_stream_t stream);
int fsr_instnorm_act_bwd_apply(int dtype, const void* g, const void* x, const float* stats, const float* sums,
                               int act, float slope, const float* prelu_weight, void* dx, int n, int hw, int c,
                               fsr_stream_t stream);

/* ------------------------------------------------------------------ activation backward of a fused conv epilogue
 * dz = g * act'(.) for the activations fsr_conv3x3 fuses; `saved` is the conv OUTPUT for
 * ReLU / LeakyReLU(slope > 0) and the saved PRE-activation for PReLU.  Tensors [n,h,w,c] `dtype`.
 * dbias (optional, float [cb]) = per-channel sums of dz; with pixel_shuffled != 0 the tensors are
 * the depth-to-space outputs of a cb = 4c channel conv and dbias index = 4*ch + 2*(y&1) + (x&1).
 * dprelu (optional) = sum g*min(saved,0).  dz may be null when only the reductions are wanted (bias gradient of a
 * layer whose activation backward was already applied by its consumer's data-gradient epilogue).
 * scratch (needed with dbias / dprelu): fsr_act_bwd_scratch(n,h,w,c,pixel_shuffled) bytes. */
size_t fsr_act_bwd_scratch(int n, int h, int w, int c, int pixel_shuffled);
int fsr_act_bwd(int dtype, const void* g, const void* saved, int act, float slope, const float* prelu_weight,
                void* dz, float* dbias, float* dprelu, void* scratch, int n, int h, int w, int c, int pixel_shuffled,
                fsr_stream_t stream);

/* out = a + b over `count` elements of `dtype` tensors of one layout (out may alias a): the gradient accumulation of an activation
 * with two consumers (the generator's long skip, model.py:115) -- in the x3 mode torch's own add cannot do it (the container's
 * float32 arithmetic is not the arithmetic of the pairs it holds). */
int fsr_add(int dtype, const void* a, const void* b, void* out, long long count, fsr_stream_t stream);

/* ------------------------------------------------------------------ 3-channel images <-> padded NHWC
 * img: float, element strides (sn, sc, sh, sw) -- any of NCHW (dataloader.py:36-38 tensors) or
 * NHWC (the head's output).  out[n,h,w,cpad] = img*scale[c] + shift[c] for c < 3, else 0.
 * (VGG19.forward's (x+1)/2, (x-mean)/std, model.py:21-22, is scale = 1/(2 std), shift = (0.5-mean)/std.) */
int fsr_image_to_nhwc(int dtype, const float* img, long long sn, long long sc, long long sh, long long sw, int n,
                      int h, int w, float scale0, float scale1, float scale2, float shift0, float shift1,
                      float shift2, void* out, int cpad, fsr_stream_t stream);
/* Head backward (model.py:102-110): dz = g * (1 - y^2) for y = tanh(z), written zero-padded NHWC;
 * g has element strides (sn,sc,sh,sw), y is the head output [n,h,w,3] float.  dbias[3] (optional) = sums;
 * scratch (with dbias): fsr_tanh_bwd_scratch() bytes.
 * fsr_tanh_bwd_image: the same gradient as a 3-channel float image dz [n,h,w,3] (no padding), for the first-layer kernels. */
size_t fsr_tanh_bwd_scratch(void);
int fsr_tanh_bwd_image(const float* g, long long sn, long long sc, long long sh, long long sw, const float* y_nhwc3, int n, int h,
                       int w, float* dz_nhwc3, float* dbias, void* scratch, fsr_stream_t stream);
int fsr_tanh_bwd_to_nhwc(int dtype, const float* g, long long sn, long long sc, long long sh, long long sw,
                         const float* y_nhwc3, int n, int h, int w, void* dz, int cpad, float* dbias, void* scratch,
                         fsr_stream_t stream);

/* uint8 HWC frames [n,h,w,3] -> float [n,h,w,3], x / 127.5 - 1 (inference.py:48: the first-layer kernels read the result
 * in place as an NCHW-shaped tensor of strides (3hw, 1, 3w, 3)). */
int fsr_u8_to_image(const uint8_t* frames, float* img, long long count, fsr_stream_t stream);

/* ------------------------------------------------------------------ first-layer convolutions straight from the image
 * Conv2d(3 -> cout, k3, p1) of Generator.neck (model.py:75-78), Discriminator.neck (model.py:143-146) and
 * vgg19.features.0 (model.py:8, with VGG19.forward's normalisation model.py:20-22 folded in) without the padded
 * NHWC copy of the image: the kernels read img (float, element strides sn,sc,sh,sw) and use
 * round_dtype(img*scale[c] + shift[c]) -- the value fsr_image_to_nhwc would have stored -- as the conv input.
 * cout must be a multiple of 16.
 *   fsr_pack_conv3x3_c3 : OIHW float [cout][3][3][3] -> `dtype` [round_up(cout,16)][32], k = (ky*3+kx)*3 + ci.
 *   fsr_conv3x3_c3_fwd  : out[n,h,w,cout] = act(conv + bias); act NONE/RELU/LEAKY/PRELU; preact optional; signs optional (16-bit
 *                         dtypes, cout % 64 == 0): uint8 [n,h,w,cout/8], bit c & 7 of byte c >> 3 = (out[..., c] > 0) -- the activation-gradient
 *                         mask of the layer as fsr_conv3x3's mask_is_addend = 2 reads it.
 *   fsr_conv3x3_c3_wgrad: dw_oihw (float [cout][3][3][3]) += d loss / d weight for dz [n,h,w,cout] `dtype`;
 *                         dbias (optional, float [cout]) += per-channel sums of dz (the bias gradient: a column of
 *                         ones in the padded K dimension of the same MFMAs);
 *                         workspace of fsr_conv3x3_c3_wgrad_workspace(n,h,w,cout) bytes.
 * transposed != 0 serves the OTHER 3-channel end, the head Conv2d(cout -> 3) + Tanh (model.py:102-110), whose backward has the
 * same shape with the roles swapped -- img is then the 3-channel gradient dz = g (1 - y^2) (fsr_tanh_bwd_image):
 *   fsr_pack_conv3x3_c3(transposed): w_oihw is the head's [3][cout][3][3] filter; the pack holds its transposed, tap-flipped
 *                         form, and fsr_conv3x3_c3_fwd of `img` with it IS the head's data gradient [n,h,w,cout];
 *   fsr_conv3x3_c3_wgrad(transposed): `dz` is the head's 64-channel INPUT, dw_oihw the head's [3][cout][3][3] gradient. */
int fsr_pack_conv3x3_c3(int dtype, const float* w_oihw, int cout, void* packed, int transposed, fsr_stream_t stream);
int fsr_conv3x3_c3_fwd(int dtype, const float* img, long long sn, long long sc, long long sh, long long sw, int n, int h,
                       int w, float scale0, float scale1, float scale2, float shift0, float shift1, float shift2,
                       const void* packed_w, const float* bias, int act, float slope, const float* prelu_weight, int cout,
                       void* out, void* preact, void* signs, fsr_stream_t stream);
size_t fsr_conv3x3_c3_wgrad_workspace(int n, int h, int w, int cout);
int fsr_conv3x3_c3_wgrad(int dtype, const float* img, long long sn, long long sc, long long sh, long long sw, int n, int h,
                         int w, float scale0, float scale1, float scale2, float shift0, float shift1, float shift2,
                         const void* dz, int cout, float* dw_oihw, float* dbias, void* workspace, int transposed,
                         fsr_stream_t stream);

/* ------------------------------------------------------------------ MaxPool2d(2,2) of vgg19.features (model.py:8)
 * x [n,h,w,c] -> y [n,h/2,w/2,c].  Backward routes g to the first maximum in window scan order; with
 * relu_mask != 0 it also applies the backward of the ReLU that produced x (dx = 0 where x <= 0). */
/* argmax (optional, uint8 [n,h/2,w/2,c]): bits 0..1 = which of the four inputs of the window (row-major, the first one equal to the
 * maximum) was taken, bit 2 = maximum > 0; fsr_maxpool2_bwd_argmax needs nothing else of the forward pass. */
int fsr_maxpool2_fwd(int dtype, const void* x, void* y, void* argmax, int n, int h, int w, int c, fsr_stream_t stream);
int fsr_maxpool2_bwd_argmax(int dtype, const void* g, const void* argmax, void* dx, int n, int h, int w, int c, int relu_mask,
                            fsr_stream_t stream);
int fsr_maxpool2_bwd(int dtype, const void* g, const void* x, const void* y, void* dx, int n, int h, int w, int c,
                     int relu_mask, fsr_stream_t stream);

/* ------------------------------------------------------------------ Discriminator head: Conv2d(512 -> 1, k=1) (model.py:184-186)
 * logits[p] = b + sum_c x[p][c]*w[c]; x [npix,c] `dtype`, logits float.
 * Backward: dx[p][c] = g[p]*w[c]; dw[c] = sum_p g[p]*x[p][c]; db[0] = sum_p g[p];
 * scratch: fsr_conv1x1_c1_bwd_scratch(c) bytes. */
int fsr_conv1x1_c1_fwd(int dtype, const void* x, const float* w, const float* b, float* logits, int npix, int c,
                       fsr_stream_t stream);
size_t fsr_conv1x1_c1_bwd_scratch(int c);
int fsr_conv1x1_c1_bwd(int dtype, const float* g, const void* x, const float* w, void* dx, float* dw, float* db,
                       void* scratch, int npix, int c, fsr_stream_t stream);

/* ------------------------------------------------------------------ losses (trainer.py:41,43,177,178,188,192,109)
 * BCEWithLogitsLoss (mean): loss[0] = mean(max(x,0) - x*t + log1p(exp(-|x|))); x, t float [count];
 * scratch (both loss forwards): fsr_loss_scratch() bytes.
 * Backward: dx = gscale[0] * (sigmoid(x) - t) / count   (gscale: device scalar, the upstream grad). */
size_t fsr_loss_scratch(void);
int fsr_bce_logits_fwd(const float* x, const float* t, float* loss, void* scratch, long long count, fsr_stream_t stream);
int fsr_bce_logits_bwd(const float* x, const float* t, const float* gscale, float* dx, long long count,
                       fsr_stream_t stream);
/* SmoothL1Loss (beta 1, mean) between `a` and `b` (`dtype` tensors, or float when dtype_is_f32_io):
 * loss[0] = mean(huber(a-b)).  Backward: da = gscale[0]*clamp(a-b,-1,1)/count (db = -da not produced:
 * the target branch of trainer.py:191/:109 needs no gradient). */
int fsr_smooth_l1_fwd(int dtype, const void* a, const void* b, float* loss, void* scratch, long long count,
                      fsr_stream_t stream);
int fsr_smooth_l1_bwd(int dtype, const void* a, const void* b, const float* gscale, void* da, long long count,
                      fsr_stream_t stream);

/* ------------------------------------------------------------------ validation metrics (trainer.py:46-69)
 * torchmetrics' SSIM (gaussian 11x11, sigma 1.5, data_range 1.0, k1 .01, k2 .03; reflect padding + crop = only windows
 * entirely inside the image) and the squared error PSNR needs, for images a, b given in [-1,1] (the kernel applies
 * trainer.py:63-65's (1 + x) / 2): float tensors of logical shape [n,3,h,w] with element strides (sn,sc,sh,sw) each --
 * the generator's NHWC head output and the NCHW HR batch are both read in place.
 * out[n][2] = (sum over channels and valid pixels of the SSIM map, sum of squared errors over all pixels):
 *   SSIM_i = out[i][0] / (3 (h-10)(w-10));  PSNR = 10 log10(1 / (sum_i out[i][1] / (3 n h w))).
 * scratch: fsr_ssim_sse_scratch(n,h,w) bytes.  h, w > 10. */
size_t fsr_ssim_sse_scratch(int n, int h, int w);
int fsr_ssim_sse(const float* a, long long asn, long long asc, long long ash, long long asw, const float* b,
                 long long bsn, long long bsc, long long bsh, long long bsw, int n, int h, int w, float* out,
                 void* scratch, fsr_stream_t stream);

/* ------------------------------------------------------------------ AdamW (trainer.py:33-38, torch defaults)
 * One fused step over a flat float parameter arena: g' = g*grad_scale (1/world_size after a SUM all-reduce);
 * p *= 1 - lr*wd; m, v updated; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).  `step_counter` is a DEVICE float
 * holding the number of steps taken so far: the call first increments it on the device, then derives the
 * bias corrections bc1 = 1 - beta1^t, bc2 = 1 - beta2^t from it (host-free, so a captured hipGraph replays). */
int fsr_adamw_step(float* p, const float* g, float* m, float* v, long long count, float lr, float beta1, float beta2,
                   float eps, float weight_decay, float* step_counter, float grad_scale, fsr_stream_t stream);

/* Dynamic loss scaling for the fp16 mode (no counterpart in the reference, which trains in fp32, trainer.py:33-43; the shape is
 * torch.cuda.amp.GradScaler's).  scale_state: DEVICE float[4] = {loss scale S, clean iterations, non-finite flag, skipped
 * iterations}; the backward passes are seeded with S (the caller reads scale_state[0] on the device).
 *   fsr_grad_nonfinite     raises the flag if any of the `count` gradients is inf / NaN (after the gradient all-reduce, so
 *                          every rank decides alike);
 *   fsr_adamw_step_scaled  fsr_adamw_step with g' = g*grad_scale/S -- and NO update at all (parameters, moments and step
 *                          counter untouched) while the flag is up;
 *   fsr_loss_scale_update  once per iteration, after the last optimizer step: flag up -> S *= backoff (>= 1), counters reset,
 *                          flag cleared; else after `growth_interval` clean iterations S *= growth.
 * All decisions are taken on the device: a captured hipGraph keeps adapting while it replays. */
int fsr_grad_nonfinite(const float* g, long long count, float* scale_state, fsr_stream_t stream);
int fsr_adamw_step_scaled(float* p, const float* g, float* m, float* v, long long count, float lr, float beta1, float beta2,
                          float eps, float weight_decay, float* step_counter, float grad_scale, const float* scale_state,
                          fsr_stream_t stream);
int fsr_loss_scale_update(float* scale_state, float growth_interval, float growth, float backoff, fsr_stream_t stream);

/* ------------------------------------------------------------------ crop + antialiased bicubic down-scale (dataloader.py:24-38)
 * For each of `n` samples: crop hr_size x hr_size at (crop_y[i], crop_x[i]) from the uint8 CHW image
 * image[i] (device pointers, heights/widths per image), hr = crop/127.5 - 1 (float NCHW [n,3,hr,hr]);
 * lr = antialiased bicubic (a = -0.5, support 2*scale) down-scale of the UNSCALED crop by `scale`,
 * then /127.5 - 1 (float NCHW [n,3,hr/scale,hr/scale]).  wtab: float [hr/scale][kmax] normalised
 * taps, xmin/xsize int [hr/scale] (identical for rows and columns; built on the host once). */
int fsr_crop_resize(const uint8_t* const* images, const int* img_h, const int* img_w, const int* crop_y,
                    const int* crop_x, int n, int hr_size, int scale, const float* wtab, const int* xmin,
                    const int* xsize, int kmax, float* hr_out, float* lr_out, float* tmp, fsr_stream_t stream);

/* ------------------------------------------------------------------ PNG -> .npy ingest (host code, no device work)
 * train.py:22-37 of the reference converts the data set once: PIL decode -> RGB -> uint8 CHW -> np.save, on 16 Python threads.
 * fsr_png_to_npy does the same natively on `threads` threads (zlib inflate, the five PNG row filters, grey / palette / alpha
 * handled as PIL's convert("RGB") does: replicated / looked up / dropped), writing .npy format 1.0 ('|u1', C order, (3, H, W)).
 * status[i] (optional): 0 converted; -1 not a PNG; -2 corrupt; -3 I/O error; -4 a PNG this decoder does not take (interlaced,
 * 16-bit, grey below 8 bits) -- the caller decodes exactly those files its own way.  Returns the number of files NOT converted.
 * fsr_png_decode_chw decodes one file into `out_chw` (3 * H * W bytes; null: only report the size). */
int fsr_png_to_npy(const char* const* png_paths, const char* const* npy_paths, int count, int threads, int* status);
int fsr_png_decode_chw(const char* png_path, unsigned char* out_chw, size_t capacity, int* height, int* width);

#ifdef __cplusplus
}
#endif
#endif /* FSR_HIP_H */
