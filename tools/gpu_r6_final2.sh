#!/bin/bash
# round 6, the x3v pass (kernel sources unchanged since gpu_r6_final.sh): GPU suite with the x3v cases, PMC traffic + rocprofv3 statistics of
# the x3v iteration, the bench line with x3v on top, the 1000-iteration x 8-seed convergence study of x3v
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 ) > $O/r6_tests.log 2>&1
cp $O/parity_errors.log $O/r6_parity_errors.log 2>/dev/null
bash tools/gpu_artifacts.sh pmc r06 "x3v" > $O/r6_artifacts_pmc.log 2>&1
( time python bench.py --steps 20 --warmup 5 ) > $O/r6_bench.log 2> $O/r6_bench.err
cp $O/bench_detail.json $O/r6_bench_detail.json 2>/dev/null
bash tools/gpu_artifacts.sh stats r06 "x3v" > $O/r6_artifacts_stats.log 2>&1
rm -rf $O/r06/prof_*/ $O/r06/pmc_*_SIZE 2>/dev/null
( time timeout 2400 python tools/convergence.py 1000 $O/r6_convergence_x3v.json --seeds 8 --modes x3v ) > $O/r6_convergence_x3v.txt 2>&1
echo done
