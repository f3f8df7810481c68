#!/bin/bash
# kernel trace of the headline loop on the multi-rank path (one rank under RCCL, every collective issued)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
rm -rf $OUT/prof_head_rccl
export TTK_BENCH_REHEARSE=rccl
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/prof_head_rccl -o head --output-format csv -- python3 $R/bench.py --steps 30 --warmup 14 --no-cpu-baseline --no-variants > $OUT/prof_head_rccl.log 2>&1 || { echo failed; tail -5 $OUT/prof_head_rccl.log; exit 1; }
find $OUT/prof_head_rccl -name "*kernel_trace.csv" -size +20M -delete
