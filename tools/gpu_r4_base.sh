#!/bin/bash
# round 4, baseline call: the driver-style bench line, rocprofv3 kernel statistics of the same iteration (single stream, eager:
# exclusive durations), per-layer conv timings (forward / data gradient / weight gradient at batch 32 and 64)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 600 python bench.py --steps 100 --warmup 10 > $O/bench_n1.json.log 2>&1
cd /tmp
FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inference --no-graph --no-f32 --no-sustained --no-cfg5 > $O/rocprof_stats.log 2>&1
cd $R
for n in 32 64; do
  echo "== batch $n" >> $O/conv_bench.txt
  timeout 300 python tools/conv_bench.py --batch $n 2>&1 | grep -v "amdgpu.ids" >> $O/conv_bench.txt
done
grep '^{' $O/bench_n1.json.log | tail -1 | cut -c1-600; ls $O/prof_stats | head; tail -40 $O/conv_bench.txt
