#!/bin/bash
# per-kernel totals of one eager single-stream bench run for a given library: tools/kstats.sh <lib.so> <tag>
set -u
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
FSR_HIP_LIB=$R/fast-srgan_amd/$1 FSR_SIDE_STREAM=0 FSR_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$2 -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inference --no-graph --no-f32 > $R/gpurun_out/ks_$2.log 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/ks_$2/b_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
it = 6.0
print("$2 total ms/iter %.3f" % (tot / it / 1e6))
for r in rows[:14]:
    print("  %-70s calls/it %5.1f  us/it %8.1f" % (r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:70], int(r["Calls"]) / it, float(r["TotalDurationNs"]) / it / 1e3))
PY
