// C-ABI entry points: library info, error reporting and the 3x3 convolution front-end that turns
// an fsr_conv_desc into tap tables for the implicit-GEMM kernel (conv_igemm.hip).
#include <string.h>

#include "fsr_common.h"
#include "fsr_conv_args.h"
#include "fsr_host.h"

static thread_local char g_err[512] = "";

int fsr_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int fsr_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fsr_fail(-3, "%s: launch failed: %s", what, hipGetErrorString(e));
  return 0;
}

static thread_local char g_kernel[160] = "";

void fsr_note_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}

extern "C" int fsr_version(void) { return FSR_ABI_VERSION; }
extern "C" const char* fsr_last_error(void) { return g_err; }
extern "C" const char* fsr_last_kernel(void) { return g_kernel; }

extern "C" int fsr_device_info(char* buf, size_t buflen) {
  if (!buf || buflen == 0) return fsr_fail(-1, "fsr_device_info: null buffer");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fsr_fail(-3, "fsr_device_info: no HIP device");
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return fsr_fail(-3, "fsr_device_info: query failed");
  snprintf(buf, buflen, "name=%s;arch=%s;cus=%d;hbm_bytes=%zu", p.name, p.gcnArchName, p.multiProcessorCount,
           (size_t)p.totalGlobalMem);
  return 0;
}

// Upper bound of the partial slots per image over every kernel configuration the dispatch may pick: one slot per
// 8x16-pixel tile (the smallest tile) or per 4-row wave group of a conv_tall3 tile (8 / 12 / 16-row tiles, two groups each:
// at most one per 4 rows), plus one (tile ranges of the persistent kernels straddle image borders).
static size_t stats_slots_bound(const fsr_conv_desc* d) {
  const int gh = d->oh, gw = d->ow;
  return (size_t)((gh + 7) / 4) * ((gw + 15) / 16) + 1;
}

extern "C" size_t fsr_conv3x3_scratch(const fsr_conv_desc* d) {
  if (!d || d->n <= 0 || d->oh <= 0 || d->ow <= 0 || d->cout <= 0) return 0;
  return (size_t)d->n * stats_slots_bound(d) * d->cout * 2 * sizeof(float);
}

static int conv3x3_enqueue(const fsr_conv_desc* d, ConvKArgs& a, hipStream_t stream);

static int conv3x3_impl(const fsr_conv_desc* d, const void* in, const void* packed_w, const float* bias,
                        const float* prelu_weight, const float* oscale, const void* dact_mask, float dact_slope,
                        void* out, void* preact, float* stats, void* scratch, hipStream_t stream, int* query_block);

extern "C" int fsr_conv3x3(const fsr_conv_desc* d, const void* in, const void* packed_w, const float* bias,
                           const float* prelu_weight, const float* oscale, const void* dact_mask, float dact_slope,
                           void* out, void* preact, float* stats, void* scratch, fsr_stream_t stream_) {
  return conv3x3_impl(d, in, packed_w, bias, prelu_weight, oscale, dact_mask, dact_slope, out, preact, stats, scratch,
                      (hipStream_t)stream_, nullptr);
}

// Which filter pack does the kernel that fsr_conv3x3 would pick for this call read?  Runs the SAME argument checks and
// dispatch chain with launches suppressed (ConvKArgs.query), so the answer cannot drift from the dispatch.
extern "C" int fsr_conv3x3_pack_block(const fsr_conv_desc* d, int optional_tensors) {
  if (!d) return fsr_fail(-1, "fsr_conv3x3_pack_block: null descriptor");
  static const float dummy_f[4] = {0.f, 0.f, 0.f, 0.f};
  const void* some = dummy_f;          // stands for "this optional tensor is given"; never dereferenced in query mode
  fsr_conv_desc q = *d;
  q.pack_lin = 0;
  int block = 0;
  const int rc = conv3x3_impl(&q, some, some, (optional_tensors & FSR_OPT_BIAS) ? dummy_f : nullptr,
                              (optional_tensors & FSR_OPT_PRELU) ? dummy_f : nullptr, (optional_tensors & FSR_OPT_OSCALE) ? dummy_f : nullptr,
                              (optional_tensors & FSR_OPT_MASK) ? some : nullptr, 0.f, (void*)some,
                              (optional_tensors & FSR_OPT_PREACT) ? (void*)some : nullptr,
                              (optional_tensors & FSR_OPT_STATS) ? (float*)some : nullptr,
                              (optional_tensors & FSR_OPT_STATS) ? (void*)some : nullptr, nullptr, &block);
  return rc < 0 ? rc : block;
}

static int conv3x3_impl(const fsr_conv_desc* d, const void* in, const void* packed_w, const float* bias,
                        const float* prelu_weight, const float* oscale, const void* dact_mask, float dact_slope,
                        void* out, void* preact, float* stats, void* scratch, hipStream_t stream, int* query_block) {
  if (!d || !in || !packed_w || !out) return fsr_fail(-1, "fsr_conv3x3: null argument");
  if (stats && !scratch) return fsr_fail(-1, "fsr_conv3x3: statistics need the scratch buffer (fsr_conv3x3_scratch bytes)");
  if (d->stride != 1 && d->stride != 2) return fsr_fail(-2, "fsr_conv3x3: stride must be 1 or 2");
  if (d->act == FSR_ACT_PRELU && !prelu_weight) return fsr_fail(-1, "fsr_conv3x3: PReLU needs its weight");
  if (d->n <= 0 || d->ih <= 0 || d->iw <= 0 || d->oh <= 0 || d->ow <= 0) return fsr_fail(-2, "fsr_conv3x3: bad dims");

  ConvKArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in;
  a.wpk = packed_w;
  a.out = out;
  a.bias = bias;
  a.prelu = prelu_weight;
  a.preact = preact;
  a.oscale = oscale;
  a.dmask = dact_mask;
  a.dmask_slope = dact_slope;
  a.dmask_add = (dact_mask && d->mask_is_addend == 1) ? 1 : 0;
  a.dmask_bits = (dact_mask && d->mask_is_addend == 2) ? 1 : 0;
  if (d->mask_is_addend && !dact_mask) return fsr_fail(-1, "fsr_conv3x3: mask_is_addend needs the dact_mask tensor");
  if (d->mask_is_addend < 0 || d->mask_is_addend > 2) return fsr_fail(-2, "fsr_conv3x3: mask_is_addend must be 0, 1 or 2");
  if (a.dmask_bits && (d->mode != FSR_CONV_DGRAD || d->stride != 2 || d->dtype == FSR_F32 || d->dtype == FSR_X3 || d->cout % 64 != 0))
    return fsr_fail(-2, "fsr_conv3x3: a sign-bit mask (mask_is_addend = 2) is read by the stride-2 data gradients of the 16-bit modes, cout %% 64 == 0");
  a.stats = stats ? (float*)scratch : nullptr;   // the kernels write per-workgroup partials; finished below
  a.stats_P_max = stats ? (int)stats_slots_bound(d) : 0;   // launchers compare their slot count with this BEFORE launching
  a.query = query_block ? 1 : 0;
  a.wlin = d->pack_lin;
  if (d->pack_lin != 0 && d->pack_lin != 64 && d->pack_lin != 128) return fsr_fail(-2, "fsr_conv3x3: pack_lin must be 0, 64 or 128");
  a.N = d->n;
  a.IH = d->ih;
  a.IW = d->iw;
  a.Cin = d->dtype == FSR_X3 ? 2 * d->cin : d->cin;   // x3: the kernels see a bf16 tensor of 2 x cin channels (hi / lo chunks of 32)
  a.Cout = d->cout;
  if (d->dtype == FSR_X3) {
    if (d->cin % 32 != 0) return fsr_fail(-2, "fsr_conv3x3: x3 tensors have a multiple of 32 channels (cin = %d)", d->cin);
    if (!d->out_f32 && d->cout % 32 != 0) return fsr_fail(-2, "fsr_conv3x3: x3 tensors have a multiple of 32 channels (cout = %d)", d->cout);
    if (d->pixel_shuffle && (d->cout / 4) % 32 != 0) return fsr_fail(-2, "fsr_conv3x3: x3 pixel shuffle needs cout / 4 %% 32 == 0");
    if (d->in_pixel_shuffled && (d->cin / 4) % 32 != 0) return fsr_fail(-2, "fsr_conv3x3: x3 in_pixel_shuffled needs cin / 4 %% 32 == 0");
    if (!query_block && ((((size_t)in | (d->out_f32 ? 0 : (size_t)out) | (size_t)preact | (size_t)dact_mask) & 127) != 0))
      return fsr_fail(-2, "fsr_conv3x3: x3 tensors must be 128-byte aligned");
  }
  a.CoutPad = (d->cout + 15) / 16 * 16;
  a.FOH = d->oh;
  a.FOW = d->ow;
  a.act = d->act;
  a.slope = d->slope;
  a.ps = d->pixel_shuffle;
  a.in_ps = d->in_pixel_shuffled;
  a.out_f32 = d->out_f32;
  a.pool2 = d->pool2;
  if (d->pool2 && (d->mode != FSR_CONV_FWD || d->dtype == FSR_F32 || d->stride != 1 || d->pixel_shuffle || d->out_f32 || stats || preact ||
                   dact_mask || oscale || (d->oh & 1) || (d->ow & 1) || d->cout % 16 != 0 || d->act == FSR_ACT_TANH))
    return fsr_fail(-2, "fsr_conv3x3: pool2 is for stride-1 forward launches of the 16-bit modes with even output extents "
                        "(no statistics / pre-activation / mask / scale tensors, no pixel shuffle)");
  if (d->out_f32 < 0 || d->out_f32 > FSR_OUT_U8) return fsr_fail(-2, "fsr_conv3x3: unknown output kind %d", d->out_f32);
  if (d->out_f32 == FSR_OUT_U8 && (d->act != FSR_ACT_TANH || d->cout > 16 || d->pixel_shuffle || d->mode != FSR_CONV_FWD))
    return fsr_fail(-2, "fsr_conv3x3: uint8 image output is for tanh heads (forward, cout <= 16)");
  if (a.ps && stats) return fsr_fail(-2, "fsr_conv3x3: statistics are not available together with pixel shuffle");
  if ((stats || preact || dact_mask) && (d->cout % 16 != 0 || (d->out_f32 && d->dtype != FSR_F32)))
    return fsr_fail(-2, "fsr_conv3x3: statistics / pre-activation / mask tensors need cout %% 16 == 0 and a `dtype` output");
  if (a.ps && (d->cout % 16 != 0)) return fsr_fail(-2, "fsr_conv3x3: pixel shuffle needs cout %% 16 == 0");
  if (a.in_ps && (d->cin % 4 != 0)) return fsr_fail(-2, "fsr_conv3x3: in_pixel_shuffled needs cin %% 4 == 0");
  if (stats && d->mode == FSR_CONV_DGRAD && d->stride != 1)
    return fsr_fail(-2, "fsr_conv3x3: statistics are not available for stride-2 data gradients");

  if (int rc = conv3x3_enqueue(d, a, stream)) return rc;
  if (query_block) {
    *query_block = a.wlin_want;
    return 0;
  }
  if (stats) {
    // second level: the image's slots added in a fixed order into stats[n][cout][2]
    return fsr_launch_reduce_partials((const float*)scratch, stats, d->n, a.stats_P, d->cout * 2, d->cout * 2, a.stats_tpi, a.stats_per, 1.f, 0, stream);
  }
  return 0;
}

static int conv3x3_enqueue(const fsr_conv_desc* d, ConvKArgs& a, hipStream_t stream) {
  if (d->mode == FSR_CONV_FWD) {
    if (d->oh != (d->ih - 1) / d->stride + 1 || d->ow != (d->iw - 1) / d->stride + 1)
      return fsr_fail(-2, "fsr_conv3x3: output dims do not match k=3,p=1,stride=%d", d->stride);
    a.GH = d->oh;
    a.GW = d->ow;
    a.org_y = a.org_x = -1;
    a.ntaps = 9;
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        const int t = ky * 3 + kx;
        a.tdy[t] = ky;
        a.tdx[t] = kx;
        a.tw[t] = t;
      }
    a.osy = a.osx = 1;
    a.ooy = a.oox = 0;
    return fsr_conv_igemm_dispatch(d->dtype, a, d->stride, stream);
  }
  if (d->mode != FSR_CONV_DGRAD) return fsr_fail(-2, "fsr_conv3x3: unknown mode %d", d->mode);
  if (d->ih != (d->oh - 1) / d->stride + 1 || d->iw != (d->ow - 1) / d->stride + 1)
    return fsr_fail(-2, "fsr_conv3x3: dgrad dims do not match k=3,p=1,stride=%d", d->stride);

  if (d->stride == 1) {
    // dx[y,x] = sum_{ky,kx} dy[y+1-ky, x+1-kx] * w[:, :, ky, kx]
    a.GH = d->oh;
    a.GW = d->ow;
    a.org_y = a.org_x = -1;
    a.ntaps = 9;
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        const int t = ky * 3 + kx;
        a.tdy[t] = 2 - ky;
        a.tdx[t] = 2 - kx;
        a.tw[t] = t;
      }
    a.osy = a.osx = 1;
    a.ooy = a.oox = 0;
    return fsr_conv_igemm_dispatch(d->dtype, a, 1, stream);
  }
  // stride 2: split dx into its four parity classes.  For dx row y = 2i+py the contributing
  // filter rows are ky = 1 (py = 0; dy row i) or ky in {0, 2} (py = 1; dy rows i+1, i).
  {   // 64..512 channels: persistent kernel, all four classes per tile, dy read once (conv_s2d3.hip)
    ConvKArgs p = a;
    if (const int rc = fsr_conv_s2d3_try(d->dtype, p, stream)) {
      a.wlin_want = p.wlin_want;
      return rc < 0 ? rc : 0;
    }
  }
  if (a.dmask_bits) return fsr_fail(-2, "fsr_conv3x3: no kernel with a sign-bit mask takes this launch");
  // The four classes go out as ONE launch, the 4-tap class first (longest workgroups first).
  ConvKArgs cls[4];
  int ncls = 0;
  for (int py = 1; py >= 0; --py)
    for (int px = 1; px >= 0; --px) {
      ConvKArgs b = a;
      b.GH = (d->oh - py + 1) / 2;
      b.GW = (d->ow - px + 1) / 2;
      if (b.GH <= 0 || b.GW <= 0) continue;
      b.org_y = b.org_x = 0;
      b.ntaps = 0;
      for (int ky = 0; ky < 3; ++ky) {
        if (((py + 1 - ky) & 1) != 0) continue;
        for (int kx = 0; kx < 3; ++kx) {
          if (((px + 1 - kx) & 1) != 0) continue;
          const int t = b.ntaps++;
          b.tdy[t] = (py + 1 - ky) / 2;
          b.tdx[t] = (px + 1 - kx) / 2;
          b.tw[t] = ky * 3 + kx;
        }
      }
      b.osy = b.osx = 2;
      b.ooy = py;
      b.oox = px;
      cls[ncls++] = b;
    }
  if (fsr_conv_stage_mode() & 16) return fsr_conv_igemm_dispatch_classes(d->dtype, cls, ncls, stream);
  for (int k = 0; k < ncls; ++k) {
    if (int rc = fsr_conv_igemm_dispatch(d->dtype, cls[k], 1, stream)) return rc;
    if (a.query) break;
  }
  return 0;
}
