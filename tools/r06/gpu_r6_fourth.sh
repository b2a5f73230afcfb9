#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out; mkdir -p $O
: > $O/r6_v3_spread.txt
export FSR_C64V3=2
for L in v3base v3spread; do
  export FSR_HIP_LIB=$PWD/fast-srgan_amd/libfsr_hip_$L.so
  echo "== $L check" >> $O/r6_v3_spread.txt; python tools/experiments/v3_check.py 2>&1 | grep -v amdgpu >> $O/r6_v3_spread.txt
done
for r in 1 2; do for L in product v3base v3spread; do
  if [ $L = product ]; then unset FSR_HIP_LIB; else export FSR_HIP_LIB=$PWD/fast-srgan_amd/libfsr_hip_$L.so; fi
  echo "== $L" >> $O/r6_v3_spread.txt
  timeout 200 python tools/conv_bench.py --batch 32 --dtype f16 --filter "64->" --only fwd 2>&1 | grep -E "VGG 64|G up|D 64->128|G stem" >> $O/r6_v3_spread.txt
done; done
unset FSR_HIP_LIB FSR_C64V3
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/r6_tests.log 2>&1
cp $O/parity_errors.log $O/r6_parity_errors.log 2>/dev/null
echo done
