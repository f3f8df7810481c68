"""Scratch: list the kernels of the last core update in a rocprofv3 kernel trace, in time order."""
import csv, sys, glob
path = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# a step starts at a gram_pack_kernel
starts = [i for i, r in enumerate(rows) if "gram_pack_kernel" in r[2]]
a, b = starts[-2], starts[-1]
seg = rows[a:b]
t0 = seg[0][0]
agg = []
for s, e, n in seg:
    short = n.split("(")[0].replace("void ", "")[:48]
    if agg and agg[-1][0] == short:
        agg[-1][1] += 1; agg[-1][2] += e - s; agg[-1][4] = e
    else:
        agg.append([short, 1, e - s, s, e])
prev_end = t0
for name, cnt, dur, s, e in agg:
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  {name:48s} x{cnt:<4d} busy {dur / 1e3:9.1f} us  span {(e - s) / 1e3:9.1f}")
    prev_end = e
print("step span us", (rows[b][0] - t0) / 1e3)
