#!/bin/bash
# quick GPU check: the op-level parity tests, then the bench line (no sustained leg)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests/test_ops.py tests/test_parity_bench.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py --steps 60 --warmup 10 --no-sustained 2>&1 | tail -1 | tee gpurun_out/r3/bench_quick.json | cut -c1-600
