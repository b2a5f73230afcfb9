#!/bin/bash
# Weight-gradient kernel: parity tests, then per-layer timing of the stride-2 forms (8-row register-staged / 4-row LDS-DMA)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_ops.py -x -q -m gpu -k "wgrad" 2>&1 | tail -3
for v in 8 4 8 4; do
  echo "== FSR_WGRAD_S2=$v"
  FSR_WGRAD_S2=$v timeout 300 python tools/conv_bench.py --only wgrad --batch 64 --filter "s2" 2>&1 | grep -v "amdgpu.ids\|first\|head" | awk -F'|' '{print $1 "|" $4}'
done > gpurun_out/r3/wgrad_s2_ab.txt 2>&1
cat gpurun_out/r3/wgrad_s2_ab.txt
