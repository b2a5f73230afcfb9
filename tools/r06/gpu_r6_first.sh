#!/bin/bash
# round 6, first GPU pass: the GPU suite with durations, the lean bench line, r5-library A/B of the x3 / f16 iterations, CPU thread sweep
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=70 ) > $O/r6_tests.log 2>&1
cp $O/parity_errors.log $O/r6_parity_errors.log 2>/dev/null
( time python bench.py --steps 20 --warmup 5 ) > $O/r6_bench.log 2> $O/r6_bench.err
cp $O/bench_detail.json $O/r6_bench_detail.json 2>/dev/null
: > $O/r6_ab.log
for r in 1 2; do for lib in r5 new; do for dt in x3 f16; do
  if [ $lib = r5 ]; then export FSR_HIP_LIB=$PWD/fast-srgan_amd/libfsr_hip_r5.so; else unset FSR_HIP_LIB; fi
  python bench.py --dtype $dt --steps 20 --warmup 5 --no-inference --no-cpu-baseline --no-cfg5 --no-f32 --no-f16 --no-bf16 --no-x3 --no-sustained --detail /tmp/d.json 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$lib $dt', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['frac'])" >> $O/r6_ab.log 2>&1
done; done; done
unset FSR_HIP_LIB
python tools/cpu_threads.py > $O/r6_cpu_threads.txt 2>&1
echo done
