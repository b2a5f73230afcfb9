// 16-bit instantiations of conv_tall3 (conv_tall3_body.h) for bf16_t: 64- and 128-channel blocks, 8 / 12 / 16-row tiles, with and
// without the InstanceNorm statistics, stride 1 and the stride-2 forward.
#include "conv_tall3_body.h"

int fsr_t3_run_bf16(ConvKArgs& b, bool narrow, int mb, int S, hipStream_t stream) {
  typedef bf16_t TT;
  if (S == 2) return b.stats ? t3_launch<TT, 128, 4, 1, 4, 2, 2, true, 2>(b, 2, stream) : t3_launch<TT, 128, 4, 1, 4, 2, 2, false, 2>(b, 2, stream);
#define T3_GO(MBV)                                                                         \
  do {                                                                                     \
    if (narrow) return t3_launch<TT, 64, 4, 1, 4, MBV, 1>(b, 2, stream);                   \
    if (b.stats) return t3_launch<TT, 128, 4, 1, 4, MBV, 2, true>(b, 2, stream);           \
    return t3_launch<TT, 128, 4, 1, 4, MBV>(b, 2, stream);                                 \
  } while (0)
  if (mb == 4) T3_GO(4);
  else if (mb == 3) T3_GO(3);
  else T3_GO(2);
#undef T3_GO
}
