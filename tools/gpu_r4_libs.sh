#!/bin/bash
# round 4: same-box comparison of library builds (tools/build_variant1.sh): $1 = conv_bench layer filter ("-" = skip), $2 = fwd,dgrad,..,
# $3.. = library suffixes ("base" = libfsr_hip.so); per-layer timings at batch 32, then the bench step twice, interleaved
set -u
R=$GRAFT_REPO_ROOT; cd $R
F=$1; W=$2; shift; shift
O=gpurun_out/r04_libs; mkdir -p $O; export TMPDIR=/tmp
lib() { [ $1 = base ] && echo $R/fast-srgan_amd/libfsr_hip.so || echo $R/fast-srgan_amd/libfsr_hip_$1.so; }
if [ "$F" != "-" ]; then
  for s in "$@"; do
    echo "== $s batch 32" >> $O/conv.txt
    FSR_HIP_LIB=$(lib $s) timeout 300 python tools/conv_bench.py --batch 32 --filter "$F" --only $W 2>&1 | grep -v "amdgpu.ids\|^layer" >> $O/conv.txt
  done
  cat $O/conv.txt
fi
for i in 1 2; do
  for s in "$@"; do
    FSR_HIP_LIB=$(lib $s) timeout 300 python bench.py --steps 100 --warmup 10 --no-f32 --no-cpu-baseline --no-inference --no-sustained --no-cfg5 2>&1 | tail -1 > $O/bench_${s}_$i.json
    python - <<PY | tee -a $O/ab.txt
import json
d=json.load(open("$O/bench_${s}_$i.json"))
print("%-8s %8.1f images/s  %.3f ms  clock %s  roofline %.4f  us %.1f" % ("$s", d["value"], d["ms_per_step"], d["clock"]["sclk_mhz_mean"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"]))
PY
  done
done
