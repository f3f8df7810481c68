#!/bin/bash
# one rocprofv3 counter pass over tools/time_bond.py <which> (run on the GPU box from the repo root)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
w=$1; shift
rm -rf $OUT/pmc_$w
timeout -s KILL 150 rocprofv3 --pmc "$@" -d $OUT/pmc_$w -o $w --output-format csv -- python3 $R/tools/time_bond.py $w > $OUT/pmc_$w.log 2>&1 || { echo "pmc $w failed"; tail -5 $OUT/pmc_$w.log; exit 1; }
f=$(find $OUT/pmc_$w -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    if "bond" in k or "filter" in k or "patch" in k:
        print(k)
        for c, v in d.items():
            print("   %-32s %16.0f per launch" % (c, v / cnt[(k, c)]))
P
find $OUT/pmc_$w -name "*.csv" -size +5M -delete
