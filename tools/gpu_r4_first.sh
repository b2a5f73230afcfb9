#!/bin/bash
# round 4, first GPU call: the stride-2 forward of conv_tall3 (tests at full size, per-layer and per-step A/B against the
# build without it), the new parity tests at the timed configurations
R=$(cd $(dirname $0)/.. && pwd); cd $R
O=gpurun_out/r4a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops.py -x -q -m gpu -k "tall3 or timed_batch" > $O/tests_ops.log 2>&1; echo "ops rc=$?" >> $O/tests_ops.log
for L in libfsr_hip.so libfsr_hip_nos2.so; do
  for n in 32 64; do
    echo "== $L batch $n" >> $O/conv_s2.log
    FSR_HIP_LIB=$R/fast-srgan_amd/$L timeout 200 python tools/conv_bench.py --batch $n --filter "s2" --only fwd 2>&1 | grep -v "amdgpu.ids" >> $O/conv_s2.log
  done
done
for rep in 1 2; do
  for L in libfsr_hip.so libfsr_hip_nos2.so; do
    FSR_HIP_LIB=$R/fast-srgan_amd/$L timeout 300 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-inference --no-f32 --no-cfg5 --no-sustained 2>/dev/null | grep '^{' > $O/bench_${L}_$rep.json
    python - <<PY >> $O/ab.log
import json
d=json.load(open("$O/bench_${L}_$rep.json"))
print("$L", d["value"], d["ms_per_step"], d["roofline"]["frac"], [(k["kernel"][:52], k["ms_per_step"], k["tflops"]) for k in d["roofline"]["kernels"]])
PY
  done
done
timeout 1200 python -m pytest tests/test_parity_bench.py -x -q -m gpu > $O/tests_parity.log 2>&1; echo "parity rc=$?" >> $O/tests_parity.log
cp gpurun_out/parity_errors.log $O/parity_errors.log 2>/dev/null
tail -3 $O/tests_ops.log; cat $O/conv_s2.log; cat $O/ab.log; tail -15 $O/tests_parity.log
