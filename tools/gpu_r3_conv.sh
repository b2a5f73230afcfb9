#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3
cd $R
timeout 1200 python tools/convergence.py 300 gpurun_out/r3/convergence.json > gpurun_out/r3/convergence.txt 2>&1
tail -70 gpurun_out/r3/convergence.txt
true
true
