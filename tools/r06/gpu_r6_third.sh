#!/bin/bash
# round 6, third GPU pass: what bounds conv64_v3 -- ablation builds (tools/build_v3_variants.sh) and SQ counters on three layers
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out; mkdir -p $O
export FSR_C64V3=2
: > $O/r6_v3_ablation.txt
for L in "" v3a1 v3a2 v3a3 v3a4 v3a8 v3a16; do
  if [ -z "$L" ]; then unset FSR_HIP_LIB; else export FSR_HIP_LIB=$PWD/fast-srgan_amd/libfsr_hip_$L.so; fi
  for F in "VGG 64->64" "G up1" "D 64->128" "G stem"; do
    echo "== ${L:-base} $F" >> $O/r6_v3_ablation.txt
    timeout 120 python tools/conv_bench.py --batch 32 --dtype f16 --filter "$F" --only fwd 2>&1 | grep -v -e amdgpu.ids -e "^layer" >> $O/r6_v3_ablation.txt
  done
done
unset FSR_HIP_LIB
PMC_KERNEL_DTYPE=f16 bash tools/pmc_kernel.sh "VGG 64->64" v3vgg libfsr_hip.so fwd > $O/r6_pmc_sq_v3_vgg.txt 2>&1
FSR_C64V3=0 PMC_KERNEL_DTYPE=f16 bash tools/pmc_kernel.sh "VGG 64->64" v2vgg libfsr_hip.so fwd > $O/r6_pmc_sq_v2_vgg.txt 2>&1
rm -rf $O/pmck_*/k_kernel_trace.csv
echo done
