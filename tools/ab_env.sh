#!/bin/bash
# A/B of one environment switch on the default bench step: tools/ab_env.sh VAR [tests...]
V=$1; shift
mkdir -p gpurun_out/ab
if [ $# -gt 0 ]; then timeout 900 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -3; fi
for rep in 1 2; do
  for g in 0 1; do
    env $V=$g timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-inference --no-f32 2>/dev/null | grep '^{' > gpurun_out/ab/bench_${V}${g}_r${rep}.json
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab/bench_${V}${g}_r${rep}.json").read().strip().splitlines()[-1]); print("$V=$g", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
  done
done
