#!/bin/bash
# same-box, interleaved A/B of a trainer switch given as $1 (e.g. FSR_EARLY_CONTENT_BWD): bench line without the f32 / CPU / inference legs
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
V=${1:-FSR_EARLY_CONTENT_BWD}
for i in 1 2 3; do
  for v in 0 1; do
    env $V=$v timeout 300 python bench.py --steps 100 --warmup 10 --no-f32 --no-cpu-baseline --no-inference --no-sustained 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$V=$v %8.1f images/s  %.3f ms  clock %s  roofline %.3f' % (d['value'], d['ms_per_step'], d['clock']['sclk_mhz_mean'], d['roofline']['frac']))"
  done
done | tee gpurun_out/r3/ab_$V.txt
