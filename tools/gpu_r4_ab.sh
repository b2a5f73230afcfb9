#!/bin/bash
# round 4: same-box A/B of a library switch given as $1 (e.g. FSR_T3_SPLIT): per-layer conv timings for the layers matching $2,
# then the bench line without the f32 / CPU / inference / cfg5 legs, interleaved; $3 = a pytest -k expression run first
set -u
R=$GRAFT_REPO_ROOT; cd $R
V=${1:-FSR_T3_SPLIT}; F=${2:-VGG}; K=${3:-}
O=gpurun_out/r04_ab_$V; mkdir -p $O; export TMPDIR=/tmp
if [ -n "$K" ]; then timeout 900 python -m pytest tests/test_ops.py -x -q -m gpu -k "$K" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -3 $O/tests.log; fi
for v in 0 1; do
  for n in 32 64; do
    echo "== $V=$v batch $n" >> $O/conv.txt
    env $V=$v timeout 300 python tools/conv_bench.py --batch $n --filter "$F" --only fwd,dgrad 2>&1 | grep -v "amdgpu.ids" >> $O/conv.txt
  done
done
for i in 1 2; do
  for v in 0 1; do
    env $V=$v timeout 300 python bench.py --steps 100 --warmup 10 --no-f32 --no-cpu-baseline --no-inference --no-sustained --no-cfg5 2>&1 | tail -1 > $O/bench_${v}_$i.json
    python - <<PY | tee -a $O/ab.txt
import json
d=json.load(open("$O/bench_${v}_$i.json"))
print("$V=$v %8.1f images/s  %.3f ms  clock %s  roofline %.4f  us %.1f" % (d["value"], d["ms_per_step"], d["clock"]["sclk_mhz_mean"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"]))
PY
  done
done
cat $O/conv.txt
