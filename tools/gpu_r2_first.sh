#!/bin/bash
# Round 2, first GPU session: the whole -m gpu suite (new parity tests log their measured errors), smoke, the bench line,
# accurate per-layer kernel times, kernel trace of one eager iteration.
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rm -f $R/gpurun_out/parity_errors.log
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit: $?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_full.log 2>&1
echo "bench exit: $?" >> gpurun_out/bench_full.log
timeout 400 python tools/conv_bench.py > gpurun_out/conv_bench.log 2>&1
cd /tmp
FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inference --no-graph --no-f32 > $R/gpurun_out/rocprof_stats.log 2>&1
cd $R
tail -15 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -2 gpurun_out/bench_full.log | cut -c1-3000; cat gpurun_out/conv_bench.log
