// x3 instantiations of conv_tall3 (conv_tall3_body.h) on 128-channel blocks (NA = 2): VGG / discriminator forwards and data
// gradients, with the InstanceNorm statistics, with a depth-to-space input (PSM = 1), the generator's up-sampling forwards
// (PSM = 2: PixelShuffle store, PReLU, pre-activation copy; model.py:30-40), and the stride-2 forward.
#include "conv_tall3_body.h"

int fsr_t3_run_x3_wide(ConvKArgs& b, int mb, int S, bool up_form, hipStream_t stream) {
  if (S == 2) return b.stats ? t3_launch<bf16_t, 128, 4, 1, 4, 2, 2, true, 2, true>(b, 2, stream) : t3_launch<bf16_t, 128, 4, 1, 4, 2, 2, false, 2, true>(b, 2, stream);
#define T3_GO3(MBV)                                                                                          \
  do {                                                                                                       \
    if (b.stats) return (b.in_ps || up_form) ? 0 : t3_launch<bf16_t, 128, 4, 1, 4, MBV, 2, true, 1, true>(b, 2, stream);   \
    if (b.in_ps) return t3_launch<bf16_t, 128, 4, 1, 4, MBV, 2, false, 1, true, 1>(b, 2, stream);            \
    if (up_form) return t3_launch<bf16_t, 128, 4, 1, 4, MBV, 2, false, 1, true, 2>(b, 2, stream);            \
    return t3_launch<bf16_t, 128, 4, 1, 4, MBV, 2, false, 1, true>(b, 2, stream);                            \
  } while (0)
  if (mb == 4) T3_GO3(4);
  else if (mb == 3) T3_GO3(3);
  else T3_GO3(2);
#undef T3_GO3
}
