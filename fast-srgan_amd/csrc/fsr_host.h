// Host-side helpers shared by the C-ABI entry points (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

struct ConvKArgs;

// Records a message retrievable through fsr_last_error() and returns `code` (negative).
int fsr_fail(int code, const char* fmt, ...);
// Returns 0 if the most recent launch was accepted by the runtime, else records and returns -3.
int fsr_check_launch(const char* what);
// Records the name of the kernel configuration a dispatch picked (fsr_last_kernel(): per-kernel attribution in bench.py).
void fsr_note_kernel(const char* fmt, ...);

int fsr_conv_igemm_dispatch(int dtype, ConvKArgs& a, int S, hipStream_t stream);
// `n` launches that differ only in output grid / taps / output offset (stride-2 data-gradient classes) as ONE launch
int fsr_conv_igemm_dispatch_classes(int dtype, ConvKArgs* cls, int n, hipStream_t stream);
int fsr_conv_stage_mode();   // FSR_CONV_STAGE tuning / test switch, see conv_igemm.hip
// 1 = launched, 0 = shape not handled by the LDS-resident-filter kernel, < 0 = error
int fsr_conv64_persistent_try(int dtype, ConvKArgs& a, int S, hipStream_t stream);
int fsr_conv_tall3_try(int dtype, ConvKArgs& a, int S, hipStream_t stream);        // 128..512 channels, stride 1: 32x32x16 MFMA, all-DMA (conv_tall3.hip)
int fsr_conv64_s2fwd_try(int dtype, ConvKArgs& a, int S, hipStream_t stream);     // stride-2 forward, 64 -> 64, persistent streaming
int fsr_conv_s2d3_try(int dtype, ConvKArgs& a, hipStream_t stream);        // stride-2 data gradient, 64..512 channels, all parity classes (conv_s2d3.hip)
// Second level of every cross-workgroup reduction (reduce.hip): out[b][i] (+)= scale * sum_p part[(b*P + p)*stride + i]
// for i < len, p in a fixed order.  per > 0: batch b owns only the slots of the tile ranges [w*per, (w+1)*per) that
// intersect its tpi tiles.
int fsr_launch_reduce_partials(const float* part, float* out, int batches, int P, int len, int stride, int tpi, int per,
                               float scale, int accumulate, hipStream_t stream);
