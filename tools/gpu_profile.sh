#!/bin/bash
# Round artefacts: bench line (with cpu_baseline), rocprofv3 kernel stats of the same command, PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only) for the dominant conv kernel on one layer.
# The kernel-stats run uses ONE stream (FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0) so that per-kernel durations are exclusive.
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 600 python $R/bench.py --steps 10 --warmup 3 > $R/gpurun_out/bench_full.log 2>&1
FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inference --no-graph > $R/gpurun_out/rocprof_stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -o conv -- python $R/tools/conv_bench.py --only fwd --filter "VGG 256" > $R/gpurun_out/pmc_$C.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_sq -o conv -- python $R/tools/conv_bench.py --only fwd --filter "VGG" > $R/gpurun_out/pmc_sq.log 2>&1
cd $R
tail -1 gpurun_out/bench_full.log | cut -c1-2000
ls gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_sq
