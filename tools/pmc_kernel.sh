#!/bin/bash
# SQ / cache counters of the convolution kernels of one conv_bench filter (separate --pmc passes, kernel-trace only).
#   tools/pmc_kernel.sh "<conv_bench filter>" <tag> [lib.so] [fwd|dgrad|wgrad]
set -u
R=$GRAFT_REPO_ROOT
F="$1"; TAG="$2"; LIB="${3:-libfsr_hip.so}"; W="${4:-fwd}"
export TMPDIR=/tmp FSR_BENCH_EAGER=1 FSR_HIP_LIB=$R/fast-srgan_amd/$LIB
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM" \
         "TCC_HIT TCC_MISS TCC_REQ" "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmck_${TAG}_$i -o k -- python $R/tools/conv_bench.py --only $W --filter "$F" --dtype ${PMC_KERNEL_DTYPE:-bf16} > $R/gpurun_out/pmck_${TAG}_$i.log 2>&1
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmck_${TAG}_*/k_counter_collection.csv
