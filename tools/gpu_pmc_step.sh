#!/bin/bash
# PMC passes (kernel-trace only) over one eager training iteration: HBM bytes of the conv kernels.
set -u
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcstep_$C -o step -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-inference --no-graph > $R/gpurun_out/pmcstep_$C.log 2>&1
done
cd $R
python tools/pmc_traffic.py gpurun_out/pmcstep_FETCH_SIZE/step_counter_collection.csv gpurun_out/pmcstep_WRITE_SIZE/step_counter_collection.csv 4
tail -1 gpurun_out/pmcstep_WRITE_SIZE.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('launches_per_step', d['roofline']['launches_per_step'], 'alg bytes', d['roofline']['algorithmic_bytes_per_launch'])"
