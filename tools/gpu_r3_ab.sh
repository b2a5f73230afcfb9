#!/bin/bash
# in-situ A/B: the bench with the round-2 tall kernel (FSR_TALL3=0) and with conv_tall3 (1), alternating
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3
export TMPDIR=/tmp
cd $R
for m in 0 1 0 1; do
  FSR_TALL3=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-f32 > gpurun_out/r3/bench_tall3_$m.log 2>&1
  tail -1 gpurun_out/r3/bench_tall3_$m.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FSR_TALL3=$m', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline'].get('kernel'))"
done
