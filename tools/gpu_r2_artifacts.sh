#!/bin/bash
# Round-2 artefacts for profiles/: the bench line, rocprofv3 kernel statistics of the same iteration (single stream: exclusive
# durations), PMC passes (FETCH_SIZE / WRITE_SIZE over one eager iteration -> profiles/conv_traffic.json; SQ counters of the
# dominant kernels), per-layer kernel times next to MIOpen, the parity-error log of the -m gpu suite.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
export TMPDIR=/tmp
cd $R
rm -f gpurun_out/parity_errors.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1
cp gpurun_out/parity_errors.log $O/parity_errors.log
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmcstep_$C -o step -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-inference --no-graph --no-f32 > $O/pmcstep_$C.log 2>&1
done
cd $R
L=$(grep '^{' $O/pmcstep_WRITE_SIZE.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['family']['launches_per_step'])")
python tools/pmc_traffic.py $O/pmcstep_FETCH_SIZE/step_counter_collection.csv $O/pmcstep_WRITE_SIZE/step_counter_collection.csv 4 $L $R/profiles/conv_traffic.json > $O/pmc_traffic.txt 2>&1
cp $R/profiles/conv_traffic.json $O/conv_traffic.json
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_n1.json.log 2>&1
cd /tmp
FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inference --no-graph --no-f32 > $O/rocprof_stats.log 2>&1
cd $R
timeout 1200 python tools/conv_bench.py --miopen > $O/conv_bench.txt 2>&1
bash tools/pmc_kernel.sh "VGG 256" r02tall > $O/pmc_sq_tall.txt 2>&1
bash tools/pmc_kernel.sh "G stem" r02stem > $O/pmc_sq_stem.txt 2>&1
bash tools/pmc_kernel.sh "D 64->128" r02d64 > $O/pmc_sq_d64.txt 2>&1
timeout 300 python bench.py --workload cfg5 --steps 5 --warmup 2 > $O/bench_cfg5.json.log 2>&1
tail -3 $O/pytest_gpu.log; cat $O/pmc_traffic.txt | tail -4; tail -1 $O/bench_n1.json.log | cut -c1-2500
