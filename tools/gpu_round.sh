#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof kernel stats.  Logs land in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit: $?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps ${STEPS:-5} --warmup 2 > gpurun_out/bench.log 2>&1
echo "bench exit: $?" >> gpurun_out/bench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inference > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
echo "rocprof exit: $?" >> $GRAFT_REPO_ROOT/gpurun_out/rocprof.log
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof | head -30
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -3 gpurun_out/bench.log
