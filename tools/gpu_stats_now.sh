#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_now -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inference --no-graph --no-f32 --no-sustained > $O/rocprof_now.log 2>&1
tail -1 $O/rocprof_now.log | cut -c1-300
rm -f $O/prof_now/bench_kernel_trace.csv.bak
