#!/bin/bash
# PMC counters of the weight-gradient kernel (four passes, kernel-trace only) on one bench layer
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03
cat > /tmp/pmcw.txt <<'EOP'
pmc: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES
pmc: GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
pmc: SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS
pmc: TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum
EOP
FSR_BENCH_EAGER=1 rocprofv3 -i /tmp/pmcw.txt --kernel-trace -d $R/gpurun_out/r03/pmc_wgrad -o pmc --output-format csv -- python $R/tools/conv_bench.py --only wgrad --batch 32 --filter "${1:-VGG 256}" > $R/gpurun_out/r03/pmc_wgrad.log 2>&1
tail -3 $R/gpurun_out/r03/pmc_wgrad.log
ls -R $R/gpurun_out/r03/pmc_wgrad | head -30
