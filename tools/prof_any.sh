#!/bin/bash
# rocprofv3 kernel statistics of any script: bash tools/prof_any.sh <tag> <script> [args]
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
tag=$1; shift
rm -rf $OUT/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$tag -o $tag --output-format csv -- python3 $R/"$@" > $OUT/prof_$tag.log 2>&1 || { echo failed; tail -5 $OUT/prof_$tag.log; exit 1; }
f=$(find $OUT/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/r03_${tag}_kernel_stats.csv
find $OUT/prof_$tag -name "*kernel_trace.csv" -size +20M -delete
