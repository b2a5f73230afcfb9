#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the convolution kernels of one conv_bench filter, split per kernel (round-4 verdict 6b: is conv64_v2's
# traffic above its algorithmic bytes halo re-reads or output copies?).   tools/pmc_fetch_write.sh "<filter>" <tag> [dtype] [fwd|dgrad]
set -u
R=$GRAFT_REPO_ROOT
F="$1"; TAG="$2"; DT="${3:-f16}"; W="${4:-fwd}"
export TMPDIR=/tmp FSR_BENCH_EAGER=1
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcfw_${TAG}_$C -o k -- python $R/tools/conv_bench.py --only $W --filter "$F" --dtype $DT > $R/gpurun_out/pmcfw_${TAG}_$C.log 2>&1
done
python - "$R/gpurun_out/pmcfw_${TAG}_FETCH_SIZE/k_counter_collection.csv" "$R/gpurun_out/pmcfw_${TAG}_WRITE_SIZE/k_counter_collection.csv" <<'EOP'
import csv, sys, collections
def per(path, counter):
    out = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "conv" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
            e = out[r["Kernel_Name"][:110]]
            e[0] += float(r["Counter_Value"]); e[1] += 1
    return out
f, w = per(sys.argv[1], "FETCH_SIZE"), per(sys.argv[2], "WRITE_SIZE")
for k in f:
    if k in w and f[k][1] and w[k][1]:
        fb, wb = 2 * f[k][0] / f[k][1] * 1024, w[k][0] / w[k][1] * 1024
        print("%-112s dispatches %3d  fetch %8.1f MB (2 x FETCH_SIZE)  write %8.1f MB  total %8.1f MB" % (k, f[k][1], fb / 1e6, wb / 1e6, (fb + wb) / 1e6))
EOP
grep -A3 "^layer" $R/gpurun_out/pmcfw_${TAG}_WRITE_SIZE.log | cut -c1-110
rm -f $R/gpurun_out/pmcfw_${TAG}_*/k_kernel_trace.csv
