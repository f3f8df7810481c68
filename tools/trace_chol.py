"""Timeline of the diagonal chain of the single-launch Cholesky (library built with -DTTK_DAG_TRACE,
path in TTK_LIB_PATH).  usage: TTK_LIB_PATH=tools/libttk_trace.so python tools/trace_chol.py [n]"""
import sys

import torch

sys.path.insert(0, ".")
from tensornetworksfork_amd import ops  # noqa: E402
from tools.time_chol import problem  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
dev = torch.device("cuda:0")
A, b = problem(n, dev)
ops.chol_mode(1)
for _ in range(3):
    x, info = ops.chol_solve_dense(A, b)
torch.cuda.synchronize()
nb = (n + 63) // 64
cw = ops._chol_ws[(A.device, n)]
off = ((nb + 1) * 64 * nb * 64 + nb * 64 * 64) * 8
off = (off + 255) // 256 * 256 + (nb + 2) * 128 + 4096 * 4
tr = cw.view(torch.int64)[off // 8: off // 8 + nb * 16].reshape(nb, 16).cpu().double() / 100.0   # microseconds
names = ["acc->lds", "F loop", "invert", "tile2 last term", "finish+publish L", "publish W"]
pairs = [(1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7)]
lo, hi = min(8, nb - 2), nb - 2
print(f"n={n} nb={nb} total {tr[nb - 1, 7] - tr[0, 1]:.1f} us; per column {(tr[hi, 1] - tr[lo, 1]) / (hi - lo):.2f} us (columns {lo}..{hi})")
for name, (a, c) in zip(names, pairs):
    d = tr[lo:hi, c] - tr[lo:hi, a]
    print(f"  {name:22s} mean {d.mean():6.2f}  min {d.min():6.2f}  max {d.max():6.2f}")
for name, (a, c) in [("  phase A0 (F0 alone)", (2, 12)), ("  phase B0", (12, 13)), ("  phase A1", (13, 14)), ("  B1+A2", (14, 15))]:
    d = tr[lo:hi, c] - tr[lo:hi, a]
    print(f"  {name:22s} mean {d.mean():6.2f}  min {d.min():6.2f}  max {d.max():6.2f}")
d = tr[lo + 1:hi + 1, 1] - tr[lo:hi, 6]
print(f"  {'L out -> next acc done':22s} mean {d.mean():6.2f}  min {d.min():6.2f}  max {d.max():6.2f}")

import numpy as np
per = (tr[1:, 1] - tr[:-1, 1]).numpy()
nxt = (tr[1:, 1] - tr[:-1, 6]).numpy()
t2 = (tr[:, 6] - tr[:, 5]).numpy()
for a in range(0, nb - 1, 8):
    b = min(nb - 1, a + 8)
    print(f"  columns {a:2d}..{b:2d}: period {per[a:b].mean():6.2f}  L out->next acc {nxt[a:b].mean():6.2f}  tile2 last {t2[a:b].mean():5.2f}")

start = tr[:, 0].numpy(); a8 = tr[:, 8].numpy(); a9 = tr[:, 9].numpy(); a1 = tr[:, 1].numpy(); lout = tr[:, 7].numpy()
t0 = start[0]
for jj in list(range(1, nb, 6)) + [nb - 1]:
    print(f"  col {jj:2d}: start {start[jj]-t0:8.1f}  bulk acc done {a8[jj]-t0:8.1f}  L(j,j-1) seen {a9[jj]-t0:8.1f}  acc done {a1[jj]-t0:8.1f}   (prev L out {lout[jj-1]-t0:8.1f})")

# second tile's last term: quarters applied beside the factorisation / inversion, and when the worker's tile was seen
raw = cw.view(torch.int64)[off // 8: off // 8 + nb * 16].reshape(nb, 16).cpu()
early = raw[1:, 11].numpy()
print("  quarters applied early, histogram 0..4:", [int((early == k).sum()) for k in range(5)])
seen = raw[1:, 10].double().numpy() / 100.0
ok = raw[1:, 10].numpy() != 0
f0 = tr[1:, 2].numpy()
print(f"  release seen in {ok.sum()} of {nb - 1} cols; mean time after the start of the F loop {np.mean(seen[ok] - f0[ok]):.2f} us")
