#!/bin/bash
# End-of-round check at HEAD: the full -m gpu suite, smoke(), the source-hash-bound artefacts (PMC record, bench line) and the
# rocprofv3 kernel statistics of the same command.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
cd $R
rm -f gpurun_out/parity_errors.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1
cp gpurun_out/parity_errors.log $O/parity_errors.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
bash tools/gpu_pmc_refresh.sh > $O/refresh.log 2>&1
cd /tmp
FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inference --no-graph --no-f32 > $O/rocprof_stats.log 2>&1
cd $R
tail -2 $O/pytest_gpu.log; tail -1 $O/smoke.log; tail -3 $O/refresh.log | cut -c1-300
