// conv_tall3.hip -- host side of the all-DMA 32x32x16 3x3 convolution: shape checks and dispatch.  The kernel template lives in
// conv_tall3_body.h (its header comment describes the design); its instantiations are compiled in four translation units --
// conv_tall3_bf16.hip, conv_tall3_f16.hip, conv_tall3_x3n.hip (x3, 64-channel blocks), conv_tall3_x3w.hip (x3, 128-channel blocks) --
// which the build compiles in parallel (as one file the 50-odd kernels took 3.5 minutes).
#include "fsr_common.h"
#include "fsr_conv_args.h"
#include "fsr_host.h"

#include <stdlib.h>

// the instantiation families (each returns what t3_launch returns: 1 launched, 0 not this kernel's shape, < 0 error)
int fsr_t3_run_bf16(ConvKArgs& b, bool narrow, int mb, int S, hipStream_t stream);
int fsr_t3_run_f16(ConvKArgs& b, bool narrow, int mb, int S, hipStream_t stream);
int fsr_t3_run_x3_narrow(ConvKArgs& b, int mb, int S, bool g3, hipStream_t stream);
int fsr_t3_run_x3_wide(ConvKArgs& b, int mb, int S, bool up_form, hipStream_t stream);

int fsr_t3_cus() {
  // FSR_PERSIST_CUS=<n> (tests): number of CUs the persistent walk is sized for -- few, so that every workgroup walks several
  // tiles and the DMA stream runs on across tile boundaries.  Read per launch (the tests change it between launches).
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


// 1 = launched, 0 = not this kernel's shape (the caller falls through to conv_igemm.hip), < 0 = error.
int fsr_conv_tall3_try(int dtype, ConvKArgs& a, int S, hipStream_t stream) {
  if ((dtype != FSR_BF16 && dtype != FSR_F16 && dtype != FSR_X3) || (S != 1 && S != 2) || a.ntaps != 9) return 0;
  if (dtype == FSR_X3 && ((a.Cin & 63) != 0 || (a.Cin >> 6) * 3 >= (1 << 15))) return 0;   // x3: a.Cin = physical channels, (hi, lo) pairs of chunks
#ifdef FSR_NO_T3S2     // A/B builds (tools/build_variant.sh): stride-2 forwards stay on conv_igemm.hip
  if (S == 2) return 0;
#endif
  if (a.Cin < 128 || a.Cin % 32 != 0 || a.Cout % 64 != 0 || a.CoutPad != a.Cout) return 0;
  // Cout = 64 (the data gradient of a 64 -> 128 layer): 64-channel blocks, a wave = 32 MB pixels x 32 channels (one filter
  // fragment, 5 reads per 4 MFMAs)
  const bool narrow = a.Cout % 128 != 0;
  // (statistics on the 64-channel block: x3 only -- the generator's 64 -> 64 forwards, which the 16-bit modes give to conv64_v2)
  if (narrow && ((a.stats && dtype != FSR_X3) || a.pool2 || (S == 2 && dtype != FSR_X3))) return 0;
  if (a.oscale || a.out_f32) return 0;
  // PixelShuffle epilogue, PReLU and the pre-activation copy (the generator's up-sampling convolutions, model.py:30-40): the x3
  // form of the 128-channel block only (the 16-bit modes run these layers on conv64_v2)
  const bool x3up = dtype == FSR_X3 && S == 1 && a.Cout % 128 == 0 && !a.stats && !a.pool2 && !a.dmask;
  if ((a.preact || a.ps || a.act == FSR_ACT_PRELU) && !x3up) return 0;
  const bool x3up_form = a.preact || a.ps || a.act == FSR_ACT_PRELU;     // (x3up holds: the PSM = 2 instantiation)
  if (x3up_form && a.in_ps) return 0;
  int ps_out_shift = 0;
  if (a.ps && a.in_ps) return 0;
  if (a.ps) {
    const int cps = a.Cout >> 2;
    if (a.Cout > 256 || cps % 32 != 0 || (cps & (cps - 1)) != 0) return 0;
    while ((1 << ps_out_shift) < cps) ++ps_out_shift;
  }
  // depth-to-space input (the data gradient of a PixelShuffle convolution): x3 only (the 16-bit modes keep their measured
  // dispatch), stride 1, whole power-of-two runs of 32-channel chunks per quadrant
  int ps_shift = 0;
  if (a.in_ps) {
    const int cpq = a.Cin >> 7;          // 32-channel chunks per quadrant
    if (dtype != FSR_X3 || S != 1 || cpq < 1 || (a.Cin & 127) != 0 || (cpq & (cpq - 1)) != 0) return 0;
    while ((1 << ps_shift) < cpq) ++ps_shift;
  }
  if (a.stats && (a.pool2 || a.dmask)) return 0;   // statistics: forward launches
  if (a.act != FSR_ACT_NONE && a.act != FSR_ACT_RELU && a.act != FSR_ACT_LEAKY && !(a.act == FSR_ACT_PRELU && x3up)) return 0;
  if (a.act == FSR_ACT_LEAKY && !(a.slope >= 0.f && a.slope <= 1.f)) return 0;   // the epilogue's max(v, slope * v) form
  if (a.osy != 1 || a.osx != 1 || a.ooy != 0 || a.oox != 0 || a.org_y != -1 || a.org_x != -1) return 0;
  if (a.pool2 && (a.dmask || (a.GH & 1) || (a.GW & 1) || S == 2)) return 0;
  if ((long long)a.N * a.IH * a.IW * a.Cin >= (1LL << 31) || (long long)a.N * a.FOH * a.FOW * a.Cout >= (1LL << 31)) return 0;
  // canonical tap order (ky, kx): which filter slice serves the tap that reads halo offset (ky, kx)
  int slice[9];
  unsigned woff[9];
  for (int t = 0; t < 9; ++t) slice[t] = -1;
  for (int t = 0; t < 9; ++t) {
    if (a.tdy[t] < 0 || a.tdy[t] > 2 || a.tdx[t] < 0 || a.tdx[t] > 2) return 0;
    slice[a.tdy[t] * 3 + a.tdx[t]] = a.tw[t];
  }
  for (int t = 0; t < 9; ++t) {
    if (slice[t] < 0) return 0;
    woff[t] = (unsigned)((size_t)slice[t] * a.CoutPad * a.Cin * 2);
  }
  // Stage-contiguous filter pack (fsr_pack_conv3x3_lin, block = this launch's channel block): the slices of a (block, chunk)
  // lie back to back, rows already in LDS order and units swizzled, so a DMA piece reads 1 KB of contiguous memory (8 whole
  // lines) instead of 16 half lines 2 Cin bytes apart.  Measured (profiles/r04_filter_pack.txt): stride 1 -3..5 %, stride-2 forward -9..14 %.
  const int bn_here = (a.Cout % 128 != 0) ? 64 : 128;
  if (a.wlin != 0 && a.wlin != bn_here)
    return fsr_fail(-2, "conv_tall3: the filter pack is stage-contiguous in blocks of %d channels, this launch needs %d", a.wlin, bn_here);
  const int wlin = a.wlin;
  if (wlin)
    for (int t = 0; t < 9; ++t) woff[t] = (unsigned)(slice[t] * bn_here * 64);
  // Tile height (16, 12 or 8 rows): the tallest one that pads the map least -- 16 rows everywhere except the 24-row maps
  // (two 12-row tiles; 16-row tiles would compute 32 rows: measured 906 against 752 TFLOP/s on 512 -> 512 @ 24^2).  Whole
  // tile rounds do NOT decide: 12-row tiles give 512 -> 512 @ 48^2 exactly 3 rounds instead of 2.25 and still measured
  // 0..4 % slower -- a workgroup whose partner has finished runs faster alone, and the shorter tile pays more staging per
  // MFMA (profiles/r03_conv_tall3_ab.txt).  FSR_T3_ROWS forces a height (tests).  Stride 2: 8-row tiles (four plane buffers).
  int best_mb = 4, best_rows = 1 << 30;
  const char* rows_env = getenv("FSR_T3_ROWS");
  const int forced = rows_env ? atoi(rows_env) : 0;
  for (int mb = 4; mb >= 2; --mb) {
    const int th = 4 * mb;
    if (forced && forced != th) continue;
    const int rows = (a.GH + th - 1) / th * th;
    if (rows < best_rows) { best_rows = rows; best_mb = mb; }
  }
  if (S == 2) best_mb = 2;
  ConvKArgs b = a;          // `a` stays untouched unless a launch is taken
  for (int t = 0; t < 9; ++t) b.t3_woff[t] = woff[t];
  b.t3_ps_shift = a.ps ? ps_out_shift : ps_shift;      // (a launch has a depth-to-space INPUT or OUTPUT, never both)
  b.tiles_x = (b.GW + 15) / 16;
  b.tiles_y = (b.GH + 4 * best_mb - 1) / (4 * best_mb);
  // x3, 64-channel block: three-tap stages (G = 3, two slots: one barrier per 12 MFMAs and wave instead of per 4).  Same-box A/B,
  // interleaved: the generator's 64 -> 64 launches 6.98 -> 6.73 ms per iteration, the x3 iteration 448.6 -> 451.1 images/s (+0.6 %);
  // what bounds the block is its 5 fragment reads per 4 MFMAs, not the barriers.  FSR_T3N_G3=0: the one-tap form (A/B).
  static const bool g3 = !(getenv("FSR_T3N_G3") && atoi(getenv("FSR_T3N_G3")) == 0);
  int rc = 0;
  if (dtype == FSR_X3) rc = narrow ? fsr_t3_run_x3_narrow(b, best_mb, S, g3, stream) : fsr_t3_run_x3_wide(b, best_mb, S, x3up_form, stream);
  else if (dtype == FSR_F16) rc = fsr_t3_run_f16(b, narrow, best_mb, S, stream);
  else rc = fsr_t3_run_bf16(b, narrow, best_mb, S, stream);
  if (rc == 1) a = b;
  return rc;
}
