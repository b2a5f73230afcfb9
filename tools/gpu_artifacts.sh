#!/bin/bash
# Artefacts for profiles/ (copy what is to be judged from gpurun_out/<tag>/ into profiles/<tag>_*): rocprofv3 kernel statistics
# (single stream, eager: exclusive durations) of the timed iteration in the x3, f16, bf16 and exact-f32 modes and of the cfg5 workload,
# of generator inference in f16 and x3, the PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate passes, kernel-trace only) bound
# to the kernel sources' hash for the x3 and the f16 iteration, and the per-layer convolution timings.
#   usage: bash tools/gpu_artifacts.sh [stats|pmc|inf|conv|all] [tag, default r06] ["modes of the stats / pmc passes", default all]
set -u
R=$GRAFT_REPO_ROOT
WHAT=${1:-all}
TAG=${2:-r06}
SMODES=${3:-"x3v x3 f16 bf16 f32"}
PMODES=${3:-"x3v x3 f16"}
O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
COMMON="--steps 5 --warmup 2 --no-cpu-baseline --no-inference --no-graph --no-f32 --no-x3 --no-f16 --no-bf16 --no-sustained --no-cfg5"
cd /tmp
if [ "$WHAT" = stats ] || [ "$WHAT" = all ]; then
  for M in $SMODES; do
    FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$M -o bench -- python $R/bench.py --dtype $M $COMMON --detail $O/bench_detail_$M.json > $O/rocprof_$M.log 2>&1
    cp $O/prof_$M/bench_kernel_stats.csv $O/bench_kernel_stats_$M.csv; rm -f $O/prof_$M/bench_kernel_trace.csv
  done
  FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -o bench -- python $R/bench.py --workload cfg5 --steps 5 --warmup 2 --no-graph --no-sustained --detail $O/bench_detail_cfg5.json > $O/rocprof_cfg5.log 2>&1
  cp $O/prof_cfg5/bench_kernel_stats.csv $O/bench_kernel_stats_cfg5.csv; rm -f $O/prof_cfg5/bench_kernel_trace.csv
fi
if [ "$WHAT" = pmc ] || [ "$WHAT" = all ]; then
  for M in $PMODES; do
    for C in FETCH_SIZE WRITE_SIZE; do
      FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 500 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_${M}_$C -o step -- python $R/bench.py --dtype $M --steps 1 --warmup 1 --no-cpu-baseline --no-inference --no-graph --no-f32 --no-x3 --no-f16 --no-bf16 --no-sustained --no-cfg5 --detail $O/pmc_detail_${M}.json > $O/pmc_${M}_$C.log 2>&1
    done
    L=$(python -c "import sys,json; print(json.load(open('$O/pmc_detail_${M}.json'))['roofline']['family']['launches_per_step'])")
    OUT=$R/profiles/conv_traffic.json; [ $M != f16 ] && OUT=$R/profiles/conv_traffic_$M.json
    (cd $R && PMC_DTYPE=$M python tools/pmc_traffic.py $O/pmc_${M}_FETCH_SIZE/step_counter_collection.csv $O/pmc_${M}_WRITE_SIZE/step_counter_collection.csv 4 $L $OUT) > $O/pmc_traffic_$M.txt 2>&1
    cp $OUT $O/
    rm -f $O/pmc_${M}_*/step_kernel_trace.csv
  done
fi
if [ "$WHAT" = inf ] || [ "$WHAT" = all ]; then
  for M in f16 x3; do
    INF_DTYPE=$M timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/inf_$M -o inf -- python $R/tools/infer_kernel_profile.py > $O/inf_$M.log 2>&1
    cp $O/inf_$M/inf_kernel_stats.csv $O/inference_kernel_stats_$M.csv; rm -f $O/inf_$M/inf_kernel_trace.csv
  done
fi
if [ "$WHAT" = conv ] || [ "$WHAT" = all ]; then
  for M in f16 x3; do
    echo "== $M, batch 32" >> $O/conv_bench_$M.txt
    (cd $R && timeout 300 python tools/conv_bench.py --batch 32 --dtype $M 2>&1 | grep -v "amdgpu.ids") >> $O/conv_bench_$M.txt
  done
fi
ls $O; tail -3 $O/pmc_traffic_*.txt 2>/dev/null; tail -2 $O/inf_*.log 2>/dev/null
