"""Scratch: per-launch durations of selected kernels from a rocprofv3 kernel trace CSV (last N launches)."""
import csv, sys, glob
path = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
pat, last = sys.argv[2], int(sys.argv[3])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        if pat in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
rows.sort()
rows = rows[-last:]
print(" ".join(f"{(e - s) / 1000:.1f}" for s, e, g in rows))
print("grids", " ".join(str(g) for s, e, g in rows))
print("gaps", " ".join(f"{(rows[i + 1][0] - rows[i][1]) / 1000:.1f}" for i in range(len(rows) - 1)))
