#!/bin/bash
# rocprofv3 kernel statistics of the bench and of the sweeps (run on the GPU box from the repo root)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
for w in "$@"; do
  rm -rf $OUT/prof_$w
  if [ "$w" = bench ]; then
    timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o $w --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/prof_$w.log 2>&1 || { echo "prof $w failed"; tail -5 $OUT/prof_$w.log; exit 1; }
  else
    timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o $w --output-format csv -- python3 $R/tools/prof_sweeps.py $w > $OUT/prof_$w.log 2>&1 || { echo "prof $w failed"; tail -5 $OUT/prof_$w.log; exit 1; }
  fi
  f=$(find $OUT/prof_$w -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/r03_${w}_kernel_stats.csv
  find $OUT/prof_$w -name "*kernel_trace.csv" -size +20M -delete
  echo "== $w"; head -12 "$f" | cut -c1-150
done
