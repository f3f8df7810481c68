"""Time ops.rhs at given shapes: python tools/time_rhs.py S C rl p rr [S C rl p rr ...]"""
import sys
import torch
sys.path.insert(0, ".")
from tensornetworksfork_amd import ops

a = [int(v) for v in sys.argv[1:]]
dev = torch.device("cuda:0")
for k in range(0, len(a), 5):
    S, C, rl, p, rr = a[k:k + 5]
    L = torch.rand(S, C, rl, dtype=torch.float64, device=dev)
    X = torch.rand(S, p, dtype=torch.float64, device=dev)
    R = torch.rand(S, rr, dtype=torch.float64, device=dev)
    g = torch.rand(S, C, dtype=torch.float64, device=dev)
    for _ in range(5):
        ops.rhs(L, X, R, g)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.rhs(L, X, R, g)
    e1.record()
    torch.cuda.synchronize()
    print((S, C, rl, p, rr), "%.1f us per call" % (e0.elapsed_time(e1) * 20))
