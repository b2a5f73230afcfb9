mkdir -p gpurun_out/r3
B=./tools/ubench/conv_bench3
for dbg in 7 6 5 3 1 2 4 0; do
  echo "== FSR_T3_DBG=$dbg (1 no stores, 2 no DMA, 4 no main loop)"
  FSR_T3_DBG=$dbg timeout 60 $B 2 0,1 1 32 32 128 256 2>&1 | tail -3
done
