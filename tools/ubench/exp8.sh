B=./tools/ubench/conv_bench3
echo "=== mode 1 (4 waves, 128x64 per wave, 2 WG/CU) vs 3 (8 waves, 256-wide) vs 5 (4 waves, 128x128 per wave, ONE wave per SIMD)"
$B 10 0,1,3,5 2>&1 | grep -v "128->128\|@24\|256->128" | cut -c1-140
