B=./tools/ubench/conv_bench3
echo "=== continuous persistent pipeline (default), uniform data"; $B 10 0,1,3 2>&1 | cut -c1-135
echo "=== one workgroup per tile (FSR_T3_PERSIST=0)"; FSR_T3_PERSIST=0 $B 10 1,3 2>&1 | grep -v "conv_igemm" | cut -c1-135
echo "=== ReLU-like forward inputs"; BENCH_RELU=1 $B 10 0,1,3 2>&1 | grep fwd | cut -c1-135
