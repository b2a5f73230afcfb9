#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3
export TMPDIR=/tmp
cd $R
rm -f gpurun_out/parity_errors.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/r3/pytest_gpu.log 2>&1
tail -15 gpurun_out/r3/pytest_gpu.log
cp gpurun_out/parity_errors.log gpurun_out/r3/parity_errors.log 2>/dev/null
cp gpurun_out/convergence.txt gpurun_out/r3/convergence_test.txt 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-f32 --no-sustained > gpurun_out/r3/bench_quick.log 2>&1
tail -1 gpurun_out/r3/bench_quick.log | cut -c1-400
