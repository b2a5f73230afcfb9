#!/bin/bash
# PMC passes + profiles/conv_traffic.json + the bench line for the CURRENT kernel sources (the part of gpu_r2_artifacts.sh that is
# tied to the source hash); everything lands under gpurun_out/r02
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmcstep_$C -o step -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-inference --no-graph --no-f32 > $O/pmcstep_$C.log 2>&1
done
cd $R
L=$(grep '^{' $O/pmcstep_WRITE_SIZE.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['family']['launches_per_step'])")
python tools/pmc_traffic.py $O/pmcstep_FETCH_SIZE/step_counter_collection.csv $O/pmcstep_WRITE_SIZE/step_counter_collection.csv 4 $L $R/profiles/conv_traffic.json > $O/pmc_traffic.txt 2>&1
cp $R/profiles/conv_traffic.json $O/conv_traffic.json
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_n1.json.log 2>&1
tail -4 $O/pmc_traffic.txt | cut -c1-300; grep '^{' $O/bench_n1.json.log | cut -c1-400
