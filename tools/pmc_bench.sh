#!/bin/bash
# one rocprofv3 counter pass over a short bench run; per-kernel averages of the counters (run on the GPU box from the repo root)
#   bash tools/pmc_bench.sh <tag> <counters...>
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
tag=$1; shift
rm -rf $OUT/pmcb_$tag
timeout -s KILL 400 rocprofv3 --pmc "$@" -d $OUT/pmcb_$tag -o $tag --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${TTK_PMC_BENCH_FLAGS:-} > $OUT/pmcb_$tag.log 2>&1 || { echo "pmc $tag failed"; tail -5 $OUT/pmcb_$tag.log; exit 1; }
f=$(find $OUT/pmcb_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" "$OUT/r03_pmc_$tag.csv" <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
with open(sys.argv[2], "w") as o:
    o.write("kernel,counter,launches,average_per_launch\n")
    for k, d in sorted(acc.items()):
        for c, v in sorted(d.items()):
            o.write('"%s",%s,%d,%.1f\n' % (k, c, cnt[(k, c)], v / cnt[(k, c)]))
            if any(t in k for t in ("gram_kernel", "matvec_fused", "chol_dag")):
                print("%-44s %-28s %6d %18.1f" % (k[:44], c, cnt[(k, c)], v / cnt[(k, c)]))
P
rm -rf $OUT/pmcb_$tag
