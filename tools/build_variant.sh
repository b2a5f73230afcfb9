#!/bin/bash
# Builds a variant of the kernel library with extra -D flags: tools/build_variant.sh <suffix> -DFSR_ABL=1 ...
set -e
R=$(cd $(dirname $0)/.. && pwd)
SUF=$1; shift
O=/tmp/fsr_var_$SUF; mkdir -p $O
for f in $R/fast-srgan_amd/csrc/*.hip $R/fast-srgan_amd/csrc/*.cpp; do
  b=$(basename ${f%.*})
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -I $R/include -I $R/fast-srgan_amd/csrc -x hip "$@" -c $f -o $O/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/fast-srgan_amd/libfsr_hip_$SUF.so $O/*.o -lz
echo built $R/fast-srgan_amd/libfsr_hip_$SUF.so
