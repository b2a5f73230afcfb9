#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
{
timeout 600 python tests/probes/bf16_trajectory_probe.py bf16 0 0 3 4 2>&1 | grep -v amdgpu.ids
FSR_WGRAD_BM=64 timeout 600 python tests/probes/bf16_trajectory_probe.py bf16 0 2>&1 | grep -v amdgpu.ids
FSR_WGRAD_S2=8 timeout 600 python tests/probes/bf16_trajectory_probe.py bf16 0 2>&1 | grep -v amdgpu.ids
FSR_HIP_LIB=$PWD/fast-srgan_amd/libfsr_hip_oldw.so timeout 600 python tests/probes/bf16_trajectory_probe.py bf16 3 4 2>&1 | grep -v amdgpu.ids
} | tee -a gpurun_out/r3/bf16_trajectories.txt
