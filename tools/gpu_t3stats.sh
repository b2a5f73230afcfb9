#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3
for m in 1 0; do
  FSR_TALL3=$m FSR_BENCH_EAGER=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3/t3stats_$m -o t -- python $R/tools/conv_bench.py --only fwd --batch 32 --filter "D 128" > /dev/null 2>&1
  echo "== FSR_TALL3=$m"; python - <<P
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/r3/t3stats_$m/t_kernel_stats.csv')))
for r in rows[:4]: print("%-100s %5s %8.1f us"%(r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3))
P
done
