B=./tools/ubench/conv_bench3
run() { echo "== $1"; shift; env "$@" timeout 60 $B 10 1,3 32 96 96 256 256 2>&1 | tail -2 | cut -c1-120; env "$@" timeout 60 $B 10 1,3 32 48 48 512 512 2>&1 | tail -2 | cut -c1-120; }
run "baseline" FSR_T3_DBG=0
run "tile-start wait vmcnt(16) [stores left in flight]" FSR_T3_DBG=8
run "no lgkmcnt(0) before barriers" FSR_T3_DBG=16
run "one tile per workgroup (hardware scheduling)" FSR_PERSIST_CUS=100000
run "baseline again" FSR_T3_DBG=0
