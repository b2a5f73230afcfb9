#!/bin/bash
# interleaved A/B of two builds of the kernel library on the default bench step: tools/ab_lib.sh libA.so libB.so [reps]
A=$1; B=$2; N=${3:-2}
for rep in $(seq $N); do
  for L in $A $B; do
    FSR_HIP_LIB=$PWD/fast-srgan_amd/$L timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-inference --no-f32 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['family']['frac'])"
  done
done
