#!/bin/bash
# Ablation of the weight-gradient kernel (variant libraries built by tools/build_variant.sh ablwN -DFSR_ABLW=N)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
{
for v in "" ablw1 ablw2 ablw4 ablw5 ablw6; do
  lib=fast-srgan_amd/libfsr_hip${v:+_$v}.so
  echo "== ${v:-shipped}"
  FSR_HIP_LIB=$PWD/$lib timeout 300 python tools/conv_bench.py --only wgrad --batch 32 --filter "VGG" 2>&1 | grep -v amdgpu.ids | grep -v "first\|head" | awk -F'|' '{print $1 "|" $4}'
done
} > gpurun_out/r3/wgrad_abl.txt 2>&1
cat gpurun_out/r3/wgrad_abl.txt
