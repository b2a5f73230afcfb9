#!/bin/bash
# Round-4 artefacts for profiles/: PMC passes (FETCH_SIZE / WRITE_SIZE over eager iterations -> profiles/conv_traffic.json, bound to
# the kernel sources' hash), the default bench line (both precisions, cfg5, inference, sustained legs, CPU baseline), rocprofv3 kernel
# statistics of the same iteration (single stream, eager: exclusive durations), SQ / cache counter tables of the stride-2 kernels
# (forward: conv_tall3 S = 2, conv64_s2fwd; data gradient: conv_s2d3) and of conv_tall3 on a 256-channel layer, per-layer conv timings.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmcstep_$C -o step -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-inference --no-graph --no-f32 --no-sustained --no-cfg5 > $O/pmcstep_$C.log 2>&1
done
cd $R
L=$(grep '^{' $O/pmcstep_WRITE_SIZE.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['family']['launches_per_step'])")
python tools/pmc_traffic.py $O/pmcstep_FETCH_SIZE/step_counter_collection.csv $O/pmcstep_WRITE_SIZE/step_counter_collection.csv 4 $L $R/profiles/conv_traffic.json > $O/pmc_traffic.txt 2>&1
cp $R/profiles/conv_traffic.json $O/conv_traffic.json
timeout 900 python bench.py > $O/bench_n1.json.log 2>&1
cd /tmp
FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inference --no-graph --no-f32 --no-sustained --no-cfg5 > $O/rocprof_stats.log 2>&1
cd $R
bash tools/pmc_kernel.sh "D s2 " r04_s2fwd libfsr_hip.so fwd > $O/pmc_sq_stride2_fwd.txt 2>&1
bash tools/pmc_kernel.sh "D s2 " r04_s2dgrad libfsr_hip.so dgrad > $O/pmc_sq_stride2_dgrad.txt 2>&1
bash tools/pmc_kernel.sh "VGG 256" r04_tall3 libfsr_hip.so fwd > $O/pmc_sq_tall3.txt 2>&1
for n in 32 64; do
  echo "== batch $n" >> $O/conv_bench.txt
  timeout 300 python tools/conv_bench.py --batch $n 2>&1 | grep -v "amdgpu.ids" >> $O/conv_bench.txt
done
rm -rf $R/gpurun_out/pmck_*/k_kernel_trace.csv
tail -4 $O/pmc_traffic.txt; tail -1 $O/bench_n1.json.log | cut -c1-1200; tail -12 $O/pmc_sq_stride2_dgrad.txt
