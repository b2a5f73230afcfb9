#!/bin/bash
# round 4: rocprofv3 kernel statistics of the bench iteration (single stream, eager: exclusive durations)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_stats; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inference --no-graph --no-f32 --no-sustained --no-cfg5 > $O/rocprof_stats.log 2>&1
grep '^{' $O/rocprof_stats.log | tail -1 | cut -c1-300
