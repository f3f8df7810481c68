#!/bin/bash
# kernel statistics of the headline loop alone (no variants): bash tools/prof_headline.sh
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
rm -rf $OUT/prof_head
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/prof_head -o head --output-format csv -- python3 $R/bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-variants > $OUT/prof_head.log 2>&1 || { echo failed; tail -5 $OUT/prof_head.log; exit 1; }
f=$(find $OUT/prof_head -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/r03_headline_kernel_stats.csv
find $OUT/prof_head -name "*kernel_trace.csv" -size +20M -delete
