#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/pytest_gpu.log 2>&1
timeout 300 python tools/conv_bench.py --only fwd,dgrad > gpurun_out/conv_bench.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_full.log 2>&1
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcstep_$C -o step -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-inference --no-graph > $R/gpurun_out/pmcstep_$C.log 2>&1
done
cd $R
tail -4 gpurun_out/pytest_gpu.log; cat gpurun_out/conv_bench.log; tail -1 gpurun_out/bench_full.log | cut -c1-2200
