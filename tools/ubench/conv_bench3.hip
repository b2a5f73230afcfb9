// A/B bench of the 128..512-channel stride-1 convolutions through the C ABI of libfsr_hip.so, without torch:
// FSR_TALL3=0 (conv_igemm.hip, the round-2 tall configuration) against FSR_TALL3=1 / 3 (conv_tall3.hip, 128- / 256-channel
// tiles), same tensors, outputs compared element by element, HIP-event timing of back-to-back launches.
//   hipcc --offload-arch=gfx950 -O2 -I include tools/ubench/conv_bench3.hip -L fast-srgan_amd -lfsr_hip
//         -Wl,-rpath,'$ORIGIN/../../fast-srgan_amd' -o tools/ubench/conv_bench3
//   ./conv_bench3 [reps] [modes, comma separated: default 0,1,3]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "fsr_hip.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

static unsigned short f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static unsigned rng_state = 12345u;
static float frand() {   // uniform (-1, 1)
  rng_state = rng_state * 1664525u + 1013904223u;
  return ((rng_state >> 8) * (1.0f / 8388608.0f)) - 1.0f;
}

struct Shape { const char* name; int n, h, w, cin, cout, dgrad, relu, mask; };

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  std::vector<int> modes;
  {
    std::string m = argc > 2 ? argv[2] : "0,1,3";
    size_t p = 0;
    while (p < m.size()) {
      modes.push_back(atoi(m.c_str() + p));
      p = m.find(',', p);
      if (p == std::string::npos) break;
      ++p;
    }
  }
  Shape custom = {"custom", 0, 0, 0, 0, 0, 0, 1, 0};
  if (argc > 7) {
    custom.n = atoi(argv[3]); custom.h = atoi(argv[4]); custom.w = atoi(argv[5]); custom.cin = atoi(argv[6]); custom.cout = atoi(argv[7]);
    if (argc > 8) { custom.dgrad = atoi(argv[8]); custom.relu = !custom.dgrad; custom.mask = custom.dgrad; }
  }
  const Shape all_shapes[] = {
      {"vgg 128->128 @192 b32 fwd", 32, 192, 192, 128, 128, 0, 1, 0},
      {"vgg 128->256 @96  b32 fwd", 32, 96, 96, 128, 256, 0, 1, 0},
      {"vgg 256->256 @96  b32 fwd", 32, 96, 96, 256, 256, 0, 1, 0},
      {"vgg 256->256 @96  b32 dgrad+mask", 32, 96, 96, 256, 256, 1, 0, 1},
      {"vgg 256->512 @48  b32 fwd", 32, 48, 48, 256, 512, 0, 1, 0},
      {"vgg 512->512 @48  b32 fwd", 32, 48, 48, 512, 512, 0, 1, 0},
      {"vgg 512->512 @48  b32 dgrad+mask", 32, 48, 48, 512, 512, 1, 0, 1},
      {"D   256->128 @96  b64 dgrad+mask", 64, 96, 96, 256, 128, 1, 0, 1},
      {"D   512->256 @48  b64 dgrad+mask", 64, 48, 48, 512, 256, 1, 0, 1},
      {"vgg 512->512 @24  b32 fwd", 32, 24, 24, 512, 512, 0, 1, 0},
  };
  std::vector<Shape> shapes;
  if (custom.n > 0) shapes.push_back(custom);
  else for (const Shape& s : all_shapes) shapes.push_back(s);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%-36s %6s %10s %10s  %s\n", "layer", "mode", "us", "TFLOP/s", "kernel / max|diff| vs mode 0");
  for (const Shape& s : shapes) {
    const size_t nin = (size_t)s.n * s.h * s.w * s.cin, nout = (size_t)s.n * s.h * s.w * s.cout;
    std::vector<unsigned short> hin(nin), hmask(nout);
    // BENCH_RELU=1: forward inputs look like ReLU outputs (half of them zero) -- what the layers see inside the network; the
    // chip clocks higher on such data than on dense uniform values
    const bool relu_in = getenv("BENCH_RELU") && atoi(getenv("BENCH_RELU")) && !s.dgrad;
    for (auto& v : hin) { const float x = frand(); v = f2bf(relu_in && x < 0.f ? 0.f : x); }
    for (auto& v : hmask) v = f2bf(frand());
    // OIHW weights of the FORWARD convolution this launch belongs to: forward cout x cin; data gradient: the forward conv maps
    // cout_l -> cin_l (in = dL/dy has cin_l = s.cin channels, out = dL/dx has s.cout channels), weights [s.cin][s.cout][3][3]
    const int wco = s.dgrad ? s.cin : s.cout, wci = s.dgrad ? s.cout : s.cin;
    std::vector<float> hw((size_t)wco * wci * 9), hb(s.cout);
    const float sc = sqrtf(2.0f / (9.0f * s.cin));
    for (auto& v : hw) v = frand() * sc * 1.7f;
    for (auto& v : hb) v = frand() * 0.1f;
    unsigned short *din, *dmask, *dw, *dout[2];
    float *dwf, *db;
    CK(hipMalloc(&din, nin * 2));
    CK(hipMalloc(&dmask, nout * 2));
    CK(hipMalloc(&dw, (size_t)9 * s.cout * s.cin * 2));
    CK(hipMalloc(&dout[0], nout * 2));
    CK(hipMalloc(&dout[1], nout * 2));
    CK(hipMalloc(&dwf, hw.size() * 4));
    CK(hipMalloc(&db, hb.size() * 4));
    CK(hipMemcpy(din, hin.data(), nin * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dmask, hmask.data(), nout * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwf, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    if (fsr_pack_conv3x3(FSR_BF16, s.dgrad ? FSR_PACK_DGRAD : FSR_PACK_FWD, dwf, wco, wci, s.cin, dw, st) != 0) {
      fprintf(stderr, "pack: %s\n", fsr_last_error());
      return 1;
    }
    fsr_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.dtype = FSR_BF16;
    d.mode = s.dgrad ? FSR_CONV_DGRAD : FSR_CONV_FWD;
    d.n = s.n; d.ih = s.h; d.iw = s.w; d.cin = s.cin;
    d.oh = s.h; d.ow = s.w; d.cout = s.cout;
    d.stride = 1;
    d.act = s.relu ? FSR_ACT_RELU : FSR_ACT_NONE;
    const double flop = 2.0 * s.n * s.h * s.w * (double)s.cout * s.cin * 9;
    std::vector<unsigned short> ref(nout), got(nout);
    for (size_t mi = 0; mi < modes.size(); ++mi) {
      char mv[16];
      snprintf(mv, sizeof(mv), "%d", modes[mi]);
      setenv("FSR_TALL3", mv, 1);
      unsigned short* o = dout[mi ? 1 : 0];
      CK(hipMemsetAsync(o, 0xff, nout * 2, st));
      auto launch = [&]() {
        const int rc = fsr_conv3x3(&d, din, dw, s.dgrad ? nullptr : db, nullptr, nullptr, s.mask ? dmask : nullptr, 0.2f, o, nullptr, nullptr,
                                   nullptr, st);
        if (rc != 0) {
          fprintf(stderr, "fsr_conv3x3: %s\n", fsr_last_error());
          exit(1);
        }
      };
      launch();
      launch();
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) launch();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps;
      char note[200];
      if (mi == 0) {
        CK(hipMemcpy(ref.data(), o, nout * 2, hipMemcpyDeviceToHost));
        snprintf(note, sizeof(note), "%s", fsr_last_kernel());
      } else {
        CK(hipMemcpy(got.data(), o, nout * 2, hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        size_t bad = 0;
        for (size_t i = 0; i < nout; ++i) {
          const double a = bf2f(got[i]), b = bf2f(ref[i]);
          const double df = fabs(a - b);
          if (!(df <= 0.02 * fabs(b) + 0.02)) ++bad;
          if (df > md || df != df) md = df;
          if (fabs(b) > mx) mx = fabs(b);
        }
        snprintf(note, sizeof(note), "%s  max|diff| %.4g (max|ref| %.3g)  bad %zu", fsr_last_kernel(), md, mx, bad);
      }
      printf("%-36s %6d %10.1f %10.1f  %s\n", s.name, modes[mi], us, flop / us * 1e-6, note);
      fflush(stdout);
    }
    CK(hipFree(din)); CK(hipFree(dmask)); CK(hipFree(dw)); CK(hipFree(dout[0])); CK(hipFree(dout[1])); CK(hipFree(dwf)); CK(hipFree(db));
  }
  return 0;
}
