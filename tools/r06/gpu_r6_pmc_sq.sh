#!/bin/bash
# SQ / cache counter tables of the dominant kernel after the round-6 code-generation fix: VGG 256->256 @96^2, batch 32, forward, x3 and fp16
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
PMC_KERNEL_DTYPE=x3 bash tools/pmc_kernel.sh "VGG 256->256" t3x3 libfsr_hip.so fwd > $O/r6_pmc_sq_tall3_x3.txt 2>&1
PMC_KERNEL_DTYPE=f16 bash tools/pmc_kernel.sh "VGG 256->256" t3f16 libfsr_hip.so fwd > $O/r6_pmc_sq_tall3_f16.txt 2>&1
rm -rf $O/pmck_*/k_kernel_trace.csv
tail -8 $O/r6_pmc_sq_tall3_x3.txt; tail -8 $O/r6_pmc_sq_tall3_f16.txt
