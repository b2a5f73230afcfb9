#!/bin/bash
# round 6, second GPU pass: the GPU suite with durations, conv64_v3 against conv64_v2 (and the FSR_C64T3 experiment) per layer and on
# the inference legs, a finer CPU thread sweep, the 1000-iteration x 8-seed convergence study of the x3 mode
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=70 ) > $O/r6_tests.log 2>&1
cp $O/parity_errors.log $O/r6_parity_errors.log 2>/dev/null
for V in v2 v3 t3; do
  case $V in v2) export FSR_C64V3=0; unset FSR_C64T3;; v3) unset FSR_C64V3 FSR_C64T3;; t3) export FSR_C64V3=0 FSR_C64T3=1;; esac
  for M in "" "--mask"; do
    echo "== $V f16 batch 32 $M" >> $O/r6_conv64.txt
    timeout 300 python tools/conv_bench.py --batch 32 --dtype f16 --filter "64->" --only fwd,dgrad $M 2>&1 | grep -v amdgpu.ids >> $O/r6_conv64.txt
  done
done
unset FSR_C64V3 FSR_C64T3
for V in v2 v3; do
  if [ $V = v2 ]; then export FSR_C64V3=0; else unset FSR_C64V3; fi
  python bench.py --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-cfg5 --no-f32 --no-x3 --no-bf16 --no-sustained --inference-dtypes f16,bf16 --inference-seconds 3 --detail $O/r6_inf_$V.json 2>/dev/null | tail -1 > $O/r6_inf_$V.log
done
unset FSR_C64V3
python tools/cpu_threads.py 4,8,12,16,24 > $O/r6_cpu_threads_fine.txt 2>&1
( time timeout 2400 python tools/convergence.py 1000 $O/r6_convergence_x3.json --seeds 8 --modes x3 ) > $O/r6_convergence_x3.txt 2>&1
echo done
