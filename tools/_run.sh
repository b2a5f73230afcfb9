cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do echo "== WLIN_EXP=$v"; if [ $v = 1 ]; then export FSR_WLIN_EXP=1; else unset FSR_WLIN_EXP; fi; python tools/conv_bench.py --batch 32 --filter "2" --only fwd,dgrad 2>&1 | grep "D s2 \|VGG 1\|VGG 2\|VGG 5\|D 1\|D 2"; done
