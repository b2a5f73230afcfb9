// x3 instantiations of conv_tall3 (conv_tall3_body.h) on 64-channel blocks (NA = 1): the generator's 64 -> 64 forwards and data
// gradients (three-tap stages, G = 3; FSR_T3N_G3=0 selects the one-tap form for A/B), the data gradients into 64 channels -- with a
// depth-to-space input (PSM = 1) for the up-sampling convolutions, model.py:30-35 --, and the discriminator's 64 -> 64 stride-2 forward.
#include "conv_tall3_body.h"

int fsr_t3_run_x3_narrow(ConvKArgs& b, int mb, int S, bool g3, hipStream_t stream) {
  if (S == 2) return b.stats ? t3_launch<bf16_t, 64, 4, 1, 4, 2, 1, true, 2, true>(b, 2, stream) : t3_launch<bf16_t, 64, 4, 1, 4, 2, 1, false, 2, true>(b, 2, stream);
#define T3_GO3(MBV)                                                                                          \
  do {                                                                                                       \
    if (b.in_ps) return b.stats ? 0 : t3_launch<bf16_t, 64, 4, 3, 2, MBV, 1, false, 1, true, 1>(b, 2, stream);   \
    if (g3 && b.stats) return t3_launch<bf16_t, 64, 4, 3, 2, MBV, 1, true, 1, true>(b, 2, stream);           \
    if (g3) return t3_launch<bf16_t, 64, 4, 3, 2, MBV, 1, false, 1, true>(b, 2, stream);                     \
    if (b.stats) return t3_launch<bf16_t, 64, 4, 1, 4, MBV, 1, true, 1, true>(b, 2, stream);                 \
    return t3_launch<bf16_t, 64, 4, 1, 4, MBV, 1, false, 1, true>(b, 2, stream);                             \
  } while (0)
  if (mb == 4) T3_GO3(4);
  else if (mb == 3) T3_GO3(3);
  else T3_GO3(2);
#undef T3_GO3
}
