"""Scratch: idle gaps between kernels inside one core update of a rocprofv3 kernel trace of bench.py."""
import csv, sys, glob, collections
rows = []
with open(glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:34]))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith('void gram_kernel')]
k = int(sys.argv[2]) if len(sys.argv) > 2 else -3
seg = rows[idx[k]:idx[k + 1]]
busy = sum(e - s for s, e, n in seg) / 1e3
print("span us", (seg[-1][1] - seg[0][0]) / 1e3, "busy", busy, "kernels", len(seg))
gap = collections.defaultdict(float); cnt = collections.Counter()
for j in range(1, len(seg)):
    g = (seg[j][0] - seg[j - 1][1]) / 1e3
    key = seg[j - 1][2][:22] + " -> " + seg[j][2][:22]
    gap[key] += g; cnt[key] += 1
for key in sorted(gap, key=lambda x: -gap[x])[:10]:
    print(key.ljust(52), cnt[key], round(gap[key], 1))
print([n[:14] for s, e, n in seg if not n.startswith('chol_') and 'chol_syrk' not in n])
last_back = max(j for j, (s, e, n) in enumerate(seg) if n.startswith('chol_back'))
t0 = seg[last_back][1]
for j in range(last_back, len(seg)):
    s, e, n = seg[j]
    print(round((s - t0) / 1e3, 1), round((e - s) / 1e3, 1), n[:40])
