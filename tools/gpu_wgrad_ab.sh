#!/bin/bash
# Weight-gradient kernel: parity tests, then per-layer timing with the 64-row block forced / the default 128-row block
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_ops.py -x -q -m gpu -k "wgrad" 2>&1 | tail -4
for bm in 64 128 64 128; do
  echo "== FSR_WGRAD_BM=$bm"
  FSR_WGRAD_BM=$bm timeout 300 python tools/conv_bench.py --only wgrad --batch 32 2>&1 | grep -v "amdgpu.ids\|first\|head" | awk -F'|' '{print $1 "|" $4}'
done > gpurun_out/r3/wgrad_ab.txt 2>&1
cat gpurun_out/r3/wgrad_ab.txt
