#!/bin/bash
# scratch: headline step at several (steps, warmup) settings on one box
set -u
R=$GRAFT_REPO_ROOT
for cfg in "20 5" "20 14" "100 5" "40 26" "20 5"; do
  set -- $cfg
  timeout -k 10 200 python $R/bench.py --steps $1 --warmup $2 --no-variants --no-cpu-baseline > /tmp/o.json 2> /tmp/o.err || { echo failed; tail -3 /tmp/o.err; exit 1; }
  python - <<PY
import json
o=json.load(open('/tmp/o.json'))
b=o['step_breakdown_ms']
print("steps $1 warmup $2: %.3f ms  gram %.3f solve %.3f other %.3f kernel %.3f"%(o['ms_per_step'],b['gram_ms'],b['solve_ms'],b['other_ms'],o['roofline']['avg_launch_ms']), flush=True)
PY
done
