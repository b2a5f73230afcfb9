cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops.py -x -q -m gpu -k "s2d3 or fused_activation_mask" 2>&1 | tail -2
python tools/conv_bench.py --batch 32 --filter "D s2" --only dgrad 2>&1 | grep "D s2"
python tools/conv_bench.py --batch 64 --filter "D s2" --only dgrad 2>&1 | grep "D s2"
bash tools/pmc_kernel.sh "D s2 " r04_s2dgrad_b libfsr_hip.so dgrad 2>&1 | grep -A5 "^conv_s2d3"
