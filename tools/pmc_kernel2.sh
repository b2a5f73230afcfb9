#!/bin/bash
# Vector-memory path counters (TA / TCP / SQ VMEM) of one conv_bench filter.
set -u
R=$GRAFT_REPO_ROOT
F="$1"; TAG="$2"; LIB="${3:-libfsr_hip.so}"
export TMPDIR=/tmp FSR_BENCH_EAGER=1 FSR_HIP_LIB=$R/fast-srgan_amd/$LIB
cd /tmp
i=0
for C in "TA_TA_BUSY TA_TOTAL_WAVEFRONTS TA_FLAT_READ_WAVEFRONTS TA_FLAT_READ_LDS_WAVEFRONTS TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" \
         "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_TCP_TA_ADDR_STALL_CYCLES" \
         "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcv_${TAG}_$i -o k -- python $R/tools/conv_bench.py --only fwd --filter "$F" > $R/gpurun_out/pmcv_${TAG}_$i.log 2>&1
done
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for path in glob.glob("$R/gpurun_out/pmcv_${TAG}_*/k_counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(ConvK")[0].replace("void ", "")[:64]
        if "conv_igemm" not in k and "conv64" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add((path, r["Dispatch_Id"]))
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()): print("    %-34s %.4g" % (c, x))
PY
