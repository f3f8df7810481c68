"""Scratch: right-hand-side kernel at shapes with one row tile (library chosen by TTK_LIB_PATH)."""
import sys, torch
sys.path.insert(0, ".")
from tensornetworksfork_amd import ops
def timed(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n
dev = torch.device("cuda:0")
for (S, C, rl, p, rr) in [(54000,9,9,50,24),(54000,1,9,50,24),(54000,1,1,50,24),(54000,1,16,65,16),(54000,1,10,50,10),(50000,1,9,65,48),(54000,1,16,50,16),(54000,1,4,50,24)]:
    L = torch.rand(S, C, rl, dtype=torch.float64, device=dev); X = torch.rand(S, p, dtype=torch.float64, device=dev)
    R = torch.rand(S, rr, dtype=torch.float64, device=dev); g = torch.rand(S, C, dtype=torch.float64, device=dev)
    print((S, C, rl, p, rr), "%.1f us" % timed(lambda: ops.rhs(L, X, R, g)), flush=True)
