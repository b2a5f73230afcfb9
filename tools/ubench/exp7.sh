B=./tools/ubench/conv_bench3
echo "=== automatic tile height"; $B 10 0,1 2>&1 | cut -c1-135
for r in 16 12 8; do echo "=== FSR_T3_ROWS=$r"; FSR_T3_ROWS=$r $B 10 1 2>&1 | grep -v "^layer" | cut -c1-135; done
