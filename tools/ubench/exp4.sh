B=./tools/ubench/conv_bench3
run() { echo "== $1"; shift; env "$@" timeout 60 $B 10 1,3 32 96 96 256 256 2>&1 | tail -2 | cut -c1-120; env "$@" timeout 60 $B 10 1,3 32 48 48 512 512 2>&1 | tail -2 | cut -c1-120; }
run "baseline" FSR_T3_DBG=0
run "no barriers in the loop (wrong results)" FSR_T3_DBG=32
run "no DMA waits (wrong results)" FSR_T3_DBG=64
run "no barriers, no DMA waits" FSR_T3_DBG=96
run "no barriers, no DMA waits, no DMA, no stores" FSR_T3_DBG=99
run "setprio 1 around each substep's MFMAs" FSR_T3_DBG=128
run "persistent walk for the 4-wave variant" FSR_T3_PERSIST=1
run "baseline again" FSR_T3_DBG=0
