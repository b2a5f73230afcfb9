#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out; : > $O/r6_nt_ab.txt
for r in 1 2; do for L in base nt21 nt28 nt29; do for dt in x3v x3; do
  if [ $L = base ]; then unset FSR_HIP_LIB; else export FSR_HIP_LIB=$PWD/fast-srgan_amd/libfsr_hip_$L.so; fi
  python bench.py --dtype $dt --steps 20 --warmup 5 --no-inference --no-cpu-baseline --no-cfg5 --no-f32 --no-f16 --no-bf16 --no-x3 --no-sustained --detail /tmp/d.json 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$L $dt', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['frac'])" >> $O/r6_nt_ab.txt 2>&1
done; done; done
cat $O/r6_nt_ab.txt
