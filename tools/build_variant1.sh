#!/bin/bash
# Builds a variant of the kernel library that differs in ONE source file: tools/build_variant1.sh <suffix> <file.hip> -DFSR_ABLS=1 ...
# (the other objects come from the product build under fast-srgan_amd/_obj)
set -e
R=$(cd $(dirname $0)/.. && pwd)
SUF=$1; F=$2; shift; shift
O=/tmp/fsr_var1_$SUF; mkdir -p $O
b=$(basename $F .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -I $R/include -I $R/fast-srgan_amd/csrc "$@" -c $R/fast-srgan_amd/csrc/$b.hip -o $O/$b.o
OBJS=$(ls $R/fast-srgan_amd/_obj/*.o | grep -v "/$b.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/fast-srgan_amd/libfsr_hip_$SUF.so $OBJS $O/$b.o -lz
echo built $R/fast-srgan_amd/libfsr_hip_$SUF.so
