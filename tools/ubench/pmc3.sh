cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03
cat > /tmp/pmc1.txt <<'EOP'
pmc: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES
pmc: GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
pmc: TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum
pmc: TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
EOP
for shape in "32 96 96 256 256" "32 48 48 512 512"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 -i /tmp/pmc1.txt --kernel-trace -d $R/gpurun_out/r03/pmc_tall3_$tag -o pmc --output-format csv -- $R/tools/ubench/conv_bench3 3 1 $shape > $R/gpurun_out/r03/pmc_tall3_$tag.log 2>&1
  tail -2 $R/gpurun_out/r03/pmc_tall3_$tag.log
done
