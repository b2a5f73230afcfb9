#!/bin/bash
# conv64 second form vs first form (FSR_CONV64_V1=1): tests, per-layer sweep, step A/B
mkdir -p gpurun_out/ab64
timeout 900 python -m pytest tests/test_ops.py tests/test_modules.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/ab64/tests.log
for v in 0 1; do
  echo "== FSR_CONV64_V1=$v" >> gpurun_out/ab64/sweep.log
  for n in 32 256; do FSR_CONV64_V1=$v python tools/conv_bench.py --batch $n --filter "G stem" --only fwd,dgrad 2>&1 | tail -1 >> gpurun_out/ab64/sweep.log; done
  FSR_CONV64_V1=$v python tools/conv_bench.py --batch 32 --filter "64->" --only fwd,dgrad 2>&1 | tail -6 >> gpurun_out/ab64/sweep.log
done
for rep in 1 2; do
  for v in 1 0; do
    FSR_CONV64_V1=$v timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-inference --no-f32 2>/dev/null | grep '^{' > gpurun_out/ab64/bench_v${v}_r${rep}.json
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab64/bench_v*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
cat gpurun_out/ab64/sweep.log gpurun_out/ab64/tests.log
