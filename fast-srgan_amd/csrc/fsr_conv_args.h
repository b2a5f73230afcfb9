// Kernel-argument blocks for the 3x3 convolution family (host + device view).
#pragma once
#include <stddef.h>

// One launch of the generic implicit-GEMM kernel.  A launch produces an output *grid* of
// GH x GW pixels per image; grid pixel (gy, gx) reads input pixels
//     (gy*S + org_y + tdy[t], gx*S + org_x + tdx[t])  for t in [0, ntaps)
// multiplies them with weight slice tw[t] ([CoutPad][Cin], Cin contiguous) and is written to
// output pixel (gy*osy + ooy, gx*osx + oox).  Forward 3x3 (stride S), data-gradient of a
// stride-1 conv (flipped taps) and the four parity classes of a stride-2 data-gradient are
// all instances of this one form.
struct ConvKArgs {
  const void* in;
  const void* wpk;
  void* out;
  const float* bias;    // [Cout] or null
  const float* prelu;   // device scalar, used when act == FSR_ACT_PRELU
  void* preact;         // optional tensor like `out`: receives the pre-activation
  const float* oscale;  // optional [Cout] scale applied to the accumulator before the bias
  const void* dmask;    // optional tensor like `out`: result *= (dmask > 0 ? 1 : dmask_slope) (fused activation backward)
  float dmask_slope;
  int dmask_add;        // the dmask tensor is an addend (result += dmask) instead of a gate
  int dmask_bits;       // dmask is a packed sign-bit tensor [N][FOH][FOW][Cout / 8] (bit = the producing layer's output > 0): conv_s2d3.hip
  float* stats;         // optional per-workgroup partials [N][stats_P][Cout][2] (sum, sum of squares of the pre-activation)
  int stats_P;          // partial slots per image (set by the launcher: tiles per image, or tile ranges per image)
  int stats_P_max;      // slots per image the caller's scratch buffer holds (fsr_conv3x3_scratch): checked BEFORE a launch
  int stats_tpi, stats_per;   // persistent kernels: tiles per image / tiles per workgroup (0 for one-tile workgroups)
  int N, IH, IW, Cin;
  int GH, GW;
  int Cout, CoutPad;
  int org_y, org_x;
  int ntaps;
  int tdy[9], tdx[9], tw[9];   // host-side tap table; the kernel reads the packed form below
  unsigned long long taps_lo;  // taps 0..7, one byte each: tdy | tdx<<2 | tw<<4
  unsigned taps_hi;            // tap 8
  int HH, HW;           // halo extent (rows, cols) this launch needs
  int FOH, FOW;         // spatial dims of the output tensor (before pixel shuffle)
  int osy, osx, ooy, oox;
  int act;
  float slope;
  int ps;               // epilogue stores depth-to-space(2); weight rows packed [q][c]
  int in_ps;            // `in` is stored depth-to-space(2) (gradient of a pixel-shuffle conv)
  int out_f32;          // 1: store float regardless of T;  2 (FSR_OUT_U8): store the uint8 image of a tanh head
  int pool2;            // epilogue stores MaxPool2d(2,2) of the activated result ([N][FOH/2][FOW/2][Cout]) instead of the result
  int tiles_x, tiles_y, nblk_n;
  int premask;          // epilogue loads every mask value before its first store
  // conv_tall3.hip: byte offset of the filter slice that serves canonical tap ky*3+kx, and the launch's tile count
  unsigned t3_woff[9];
  int t3_ntiles;
  int t3_ps_shift;      // depth-to-space input: log2 of the 32-channel chunks per quadrant
  int wlin;             // 0: standard pack [9][rows][K]; 64 / 128: stage-contiguous pack of that channel-block size
                        // ([block][32-channel chunk][slice][row][32]: a DMA piece = 1 KB of contiguous memory; fsr_pack_conv3x3_lin)
  int query, wlin_want; // query = 1 (fsr_conv3x3_pack_block): nothing is launched; the kernel that WOULD run records the block
                        // size of the stage-contiguous pack it takes (0: it reads the standard pack)
};

// A launch may cover up to four "classes" that differ only in their output grid, tap table and output offset (the
// parity classes of a stride-2 data gradient: one launch, workgroups [wg_end[k-1], wg_end[k]) belong to class k, longest
// classes first).  n <= 1: the ConvKArgs fields are used as they are.
struct ConvKClass {
  int GH, GW, ntaps, ooy, oox, tiles_x, tiles_y, wg_end;
  unsigned long long taps_lo;
};
struct ConvKClasses {
  int n;
  ConvKClass c[4];
};
