#!/bin/bash
# ablation builds of conv64_v3.hip (results wrong on purpose): libfsr_hip_v3a<mask>.so = the in-tree objects with conv64_v3.o replaced
R=$(cd $(dirname $0)/.. && pwd)
for M in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -I $R/include -I $R/fast-srgan_amd/csrc -DFSR_ABLV3=$M -c $R/fast-srgan_amd/csrc/conv64_v3.hip -o /tmp/conv64_v3_a$M.o &
done
wait
for M in "$@"; do
  OBJS=$(ls $R/fast-srgan_amd/_obj/*.o | grep -v conv64_v3.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/fast-srgan_amd/libfsr_hip_v3a$M.so $OBJS /tmp/conv64_v3_a$M.o -lz
done
ls -la $R/fast-srgan_amd/*.so
