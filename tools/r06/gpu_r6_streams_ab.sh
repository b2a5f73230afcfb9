cd ${GRAFT_REPO_ROOT:-.}
for r in 1 2; do for cfg in "1 1" "0 1" "1 0" "0 0"; do set -- $cfg
  FSR_SIDE_STREAM=$1 FSR_WGRAD_STREAM=$2 python bench.py --dtype x3v --steps 20 --warmup 5 --no-inference --no-cpu-baseline --no-cfg5 --no-f32 --no-f16 --no-bf16 --no-x3 --no-sustained --detail /tmp/d.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side=$1 wgrad=$2', d['value'], d['ms_per_step'])"
done; done
