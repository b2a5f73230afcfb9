#!/bin/bash
# round 6, final pass for the CURRENT kernel sources: GPU suite, PMC traffic (bound to the source hash), rocprofv3 kernel statistics,
# inference profiles, per-layer timings, then the bench line the way the driver runs it
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 ) > $O/r6_tests.log 2>&1
cp $O/parity_errors.log $O/r6_parity_errors.log 2>/dev/null
bash tools/gpu_artifacts.sh pmc r06 > $O/r6_artifacts_pmc.log 2>&1
( time python bench.py --steps 20 --warmup 5 ) > $O/r6_bench.log 2> $O/r6_bench.err
cp $O/bench_detail.json $O/r6_bench_detail.json 2>/dev/null
bash tools/gpu_artifacts.sh stats r06 > $O/r6_artifacts_stats.log 2>&1
bash tools/gpu_artifacts.sh inf r06 > $O/r6_artifacts_inf.log 2>&1
bash tools/gpu_artifacts.sh conv r06 > $O/r6_artifacts_conv.log 2>&1
rm -rf $O/r06/prof_*/ $O/r06/pmc_*_SIZE $O/r06/inf_x3 $O/r06/inf_f16 2>/dev/null
echo done
