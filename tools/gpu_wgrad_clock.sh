#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
{
for v in "" ablw7 ablw1; do
  lib=fast-srgan_amd/libfsr_hip${v:+_$v}.so
  FSR_HIP_LIB=$PWD/$lib timeout 120 python tools/wgrad_clock.py 2>&1 | grep -v amdgpu.ids
done
} | tee gpurun_out/r3/wgrad_clock.txt
