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

int fsr_conv_igemm_dispatch(int dtype, ConvKArgs& a, int S, hipStream_t stream);
// 1 = launched, 0 = shape not handled by the LDS-resident-filter kernel, < 0 = error
int fsr_conv64_persistent_try(int dtype, ConvKArgs& a, int S, hipStream_t stream);
