// Probe: HBM write bandwidth of the conv epilogue's store patterns (NHWC bf16, 64 or 128 channels per pixel).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/store_pattern tools/probes/store_pattern.hip ; run it on the GPU box.
//   mode 0: lane (l15, lg) stores 8 B at pixel l15, channel n*16 + lg*4     (generic epilogue: 32-B pieces per pixel)
//   mode 1: lane stores 16 B at pixel l15, byte p*64 + lg*16                  (first-layer kernel: 64-B pieces)
//   mode 2: lane stores 16 B, 64 lanes contiguous (1 KB per instruction)      (what a transposing epilogue would do)
//   mode 3: as 0 but output pixels strided by 2 in x and y (one parity class of a stride-2 data gradient)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void k(unsigned short* out, int H, int W, int C, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int tiles_x = W / 16, tiles_y = H / 16;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; const int img = bid / tiles_y;
  const int NT = C / 16;
  uint2 v8 = {0x3f803f80u, 0x3f803f80u};
  uint4 v16 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  for (int m = 0; m < 4; ++m) {
    const int y = ty * 16 + wave * 4 + m, x = tx * 16 + l15;
    if (mode == 0 || mode == 3) {
      size_t pix = mode == 0 ? ((size_t)img * H + y) * W + x : ((size_t)img * 2 * H + 2 * y) * (2 * W) + 2 * x;
      for (int n = 0; n < NT; ++n) *(uint2*)(out + pix * C + n * 16 + lg * 4) = v8;
    } else if (mode == 1) {
      size_t pix = ((size_t)img * H + y) * W + x;
      for (int p = 0; p < NT / 2; ++p) *(uint4*)(out + pix * C + p * 32 + lg * 8) = v16;
    } else {
      // the wave's 4 rows x 16 pixels x C channels as contiguous 1 KB pieces
      size_t base = (((size_t)img * H + ty * 16 + wave * 4 + m) * W + tx * 16) * C;   // 16 pixels * C * 2 B contiguous
      for (int i = 0; i < 16 * C * 2 / 1024; ++i) *(uint4*)(out + base + (size_t)i * 512 + lane * 8) = v16;
    }
  }
}

int main() {
  const int N = 32, H = 384, W = 384;
  for (int C : {64, 128}) {
    const int h = C == 64 ? H : H / 2, w = C == 64 ? W : W / 2;
    size_t bytes = (size_t)N * h * w * C * 2;
    unsigned short* out;
    hipMalloc(&out, bytes * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9;
      for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(N * (h / 16) * (w / 16)), dim3(256), 0, 0, out, h, w, C, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("C=%d mode %d: %.1f us  %.2f TB/s\n", C, mode, best * 1e3, bytes / best / 1e9);
    }
    hipFree(out);
  }
  return 0;
}
