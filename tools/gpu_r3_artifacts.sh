#!/bin/bash
# Round-3 artefacts for profiles/: the default bench line (both precisions, sustained legs, clocks), rocprofv3 kernel statistics of
# the same iteration (single stream, eager: exclusive durations), PMC passes (FETCH_SIZE / WRITE_SIZE over eager iterations ->
# profiles/conv_traffic.json, bound to the kernel sources' hash), SQ / TCP counters of conv_tall3 on two layers, the per-layer A/B
# of conv_tall3 against the round-2 tall configuration.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmcstep_$C -o step -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-inference --no-graph --no-f32 --no-sustained > $O/pmcstep_$C.log 2>&1
done
cd $R
L=$(grep '^{' $O/pmcstep_WRITE_SIZE.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['family']['launches_per_step'])")
python tools/pmc_traffic.py $O/pmcstep_FETCH_SIZE/step_counter_collection.csv $O/pmcstep_WRITE_SIZE/step_counter_collection.csv 4 $L $R/profiles/conv_traffic.json > $O/pmc_traffic.txt 2>&1
cp $R/profiles/conv_traffic.json $O/conv_traffic.json
timeout 900 python bench.py > $O/bench_n1.json.log 2>&1
cd /tmp
FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inference --no-graph --no-f32 --no-sustained > $O/rocprof_stats.log 2>&1
cd $R
./tools/ubench/conv_bench3 10 0,1,3 > $O/conv_tall3_ab.txt 2>&1
bash tools/ubench/pmc3.sh > $O/pmc3.log 2>&1
timeout 300 python bench.py --workload cfg5 --dtype f16 --steps 10 --warmup 3 > $O/bench_cfg5_f16.json.log 2>&1
timeout 300 python bench.py --workload cfg5 --steps 10 --warmup 3 > $O/bench_cfg5.json.log 2>&1
tail -4 $O/pmc_traffic.txt; tail -1 $O/bench_n1.json.log | cut -c1-1500; ls $O $O/prof_stats | head -30
