# ablation of conv_tall3 on two big layers: FSR_T3_DBG 0 full, 1 no stores, 2 no DMA (garbage operands), 3 neither
mkdir -p gpurun_out/r3
B=./tools/ubench/conv_bench3
for shape in "32 96 96 256 256" "32 48 48 512 512"; do
  for dbg in 0 1 2 3; do
    echo "== shape $shape FSR_T3_DBG=$dbg"
    FSR_T3_DBG=$dbg timeout 60 $B 10 1,3 $shape 2>&1 | tail -2 | cut -c1-110
  done
done
cd /tmp && export TMPDIR=/tmp
cat > /tmp/pmc1.txt <<'EOP'
pmc: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES
pmc: SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_VALU
pmc: GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
EOP
for m in 1 3; do
  rocprofv3 -i /tmp/pmc1.txt --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r3/pmc_m$m -o pmc --output-format csv -- $GRAFT_REPO_ROOT/tools/ubench/conv_bench3 3 $m 32 96 96 256 256 > $GRAFT_REPO_ROOT/gpurun_out/r3/pmc_m$m.log 2>&1
done
ls -R $GRAFT_REPO_ROOT/gpurun_out/r3 | head -40
