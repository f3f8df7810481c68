#!/bin/bash
# scratch: headline step of several trees under tools/ on one box: bash tools/head_ab.sh _head _v1 ...
set -u
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
for t in "$@"; do
  tree=$R/tools/$t
  timeout -k 10 200 python $tree/bench.py --steps 60 --warmup 5 --no-variants --no-cpu-baseline > /tmp/o.json 2> /tmp/o.err || { echo failed $t; tail -3 /tmp/o.err; exit 1; }
  python - <<PY
import json
o=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
b=o['step_breakdown_ms']
print("tree $t: %.3f ms  gram %.3f solve %.3f other %.3f kernel %.3f"%(o['ms_per_step'],b['gram_ms'],b['solve_ms'],b['other_ms'],o['roofline']['avg_launch_ms']), flush=True)
PY
done; done
