"""Scratch: sum PMC counter values per kernel-name pattern from a rocprofv3 counter_collection CSV."""
import csv, sys, glob, collections
path = glob.glob(sys.argv[1] + "/*counter_collection.csv")[0]
pat = sys.argv[2]
tot = collections.defaultdict(float)
n = collections.defaultdict(int)
for r in csv.DictReader(open(path)):
    if pat in r["Kernel_Name"]:
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(k, n[k], f"{tot[k]:.4g}", f"per-dispatch {tot[k] / n[k]:.4g}")
