#!/bin/bash
# round 4: per-layer timings of ablation builds (tools/build_variant1.sh) of one kernel: $1 = library suffix prefix (e.g. s2d_),
# $2 = conv_bench layer filter, $3 = fwd|dgrad, $4.. = the masks
set -u
R=$GRAFT_REPO_ROOT; cd $R
P=$1; F=$2; W=$3; shift; shift; shift
O=gpurun_out/r04_abl_$P; mkdir -p $O; export TMPDIR=/tmp
for m in base "$@"; do
  L=libfsr_hip_$P$m.so; [ $m = base ] && L=libfsr_hip.so
  for n in 32 64; do
    echo "== $L batch $n" >> $O/abl.txt
    FSR_HIP_LIB=$R/fast-srgan_amd/$L timeout 200 python tools/conv_bench.py --batch $n --filter "$F" --only $W 2>&1 | grep -v "amdgpu.ids\|^layer" >> $O/abl.txt
  done
done
cat $O/abl.txt
