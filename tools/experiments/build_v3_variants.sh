#!/bin/bash
# Experiment builds of tools/experiments/conv64_v3.hip: libfsr_hip_v3<tag>.so = the in-tree objects + conv64_v3.o (compiled with the
# given -D flags) + conv_igemm.o rebuilt with the dispatch hook (-DFSR_EXPERIMENT_C64V3).  Run `python fast-srgan_amd/build.py` first.
#   usage: build_v3_variants.sh <tag>[:-Dflag[,-Dflag...]] ...     e.g.  base  a1:-DFSR_ABLV3=1  spread:-DFSR_V3_SPREAD
R=$(cd $(dirname $0)/../.. && pwd)
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -I $R/include -I $R/fast-srgan_amd/csrc"
$CC -DFSR_EXPERIMENT_C64V3 -c $R/fast-srgan_amd/csrc/conv_igemm.hip -o /tmp/conv_igemm_v3hook.o &
for V in "$@"; do
  TAG=${V%%:*}; FL=""; [ "$V" != "$TAG" ] && FL=$(echo ${V#*:} | tr ',' ' ')
  $CC $FL -x hip -c $R/tools/experiments/conv64_v3.hip -o /tmp/conv64_v3_$TAG.o &
done
wait
for V in "$@"; do
  TAG=${V%%:*}
  OBJS=$(ls $R/fast-srgan_amd/_obj/*.o | grep -v conv_igemm.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/fast-srgan_amd/libfsr_hip_v3$TAG.so $OBJS /tmp/conv_igemm_v3hook.o /tmp/conv64_v3_$TAG.o -lz
done
ls -la $R/fast-srgan_amd/*.so
