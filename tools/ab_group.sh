#!/bin/bash
# A/B of the grouped generator-stem weight gradient (FSR_WGRAD_GROUP=0 / 1) on the default bench step.
mkdir -p gpurun_out/ab
timeout 600 python -m pytest tests/test_ops.py -x -q -m gpu -k "wgrad_grouped or conv_fwd_dgrad_wgrad" 2>&1 | tail -3 > gpurun_out/ab/tests.log
timeout 600 python -m pytest tests/test_trainer.py tests/test_parity_bench.py -x -q -m gpu 2>&1 | tail -3 >> gpurun_out/ab/tests.log
for rep in 1 2; do
  for g in 0 1; do
    FSR_WGRAD_GROUP=$g timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-inference --no-f32 2>/dev/null | grep '^{' > gpurun_out/ab/bench_g${g}_r${rep}.json
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab/bench_g*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
cat gpurun_out/ab/tests.log
