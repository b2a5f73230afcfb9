#!/bin/bash
# weight-gradient tuning switches in the x3v iteration (same box, interleaved)
cd ${GRAFT_REPO_ROOT:-.}
for r in 1 2; do for cfg in "base" "FSR_WGRAD_BM=64" "FSR_WGRAD_S2=8" "FSR_WGRAD_GROUP=0"; do
  env $( [ "$cfg" = base ] && echo "FSR_DUMMY=1" || echo "$cfg" ) python bench.py --dtype x3v --steps 20 --warmup 5 --no-inference --no-cpu-baseline --no-cfg5 --no-f32 --no-f16 --no-bf16 --no-x3 --no-sustained --detail /tmp/d_$r.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'])"
  python -c "import json; d=json.load(open('/tmp/d_$r.json')); print('   wgrad', d['roofline']['weight_gradient'])"
done; done
