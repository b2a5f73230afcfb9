#!/bin/bash
# round 4: per-layer conv timings under several values of one environment variable: $1 = variable, $2 = layer filter,
# $3 = fwd|dgrad|fwd,dgrad, $4.. = values ("-" = unset)
set -u
R=$GRAFT_REPO_ROOT; cd $R
V=$1; F=$2; W=$3; shift; shift; shift
O=gpurun_out/r04_env_$V; mkdir -p $O; export TMPDIR=/tmp
for val in "$@"; do
  for n in 32 64; do
    echo "== $V=$val batch $n" >> $O/env.txt
    if [ "$val" = "-" ]; then timeout 200 python tools/conv_bench.py --batch $n --filter "$F" --only $W 2>&1 | grep -v "amdgpu.ids\|^layer" >> $O/env.txt
    else env $V=$val timeout 200 python tools/conv_bench.py --batch $n --filter "$F" --only $W 2>&1 | grep -v "amdgpu.ids\|^layer" >> $O/env.txt; fi
  done
done
cat $O/env.txt
