"""Scratch: the fused matrix-free product at the headline shape, timed with events; TTK_LIB_PATH selects a
diagnostic build (phases switched off)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd import ops
dev = torch.device("cuda:0")
S, rl, p = 54000, 32, 6
g = torch.Generator().manual_seed(0)
L = torch.rand(S, 1, rl, generator=g, dtype=torch.float64).to(dev)
X = torch.rand(S, p, generator=g, dtype=torch.float64).to(dev)
R = torch.rand(S, rl, generator=g, dtype=torch.float64).to(dev)
H = torch.ones(S, 1, 1, dtype=torch.float64, device=dev)
v = torch.rand(rl, p, rl, generator=g, dtype=torch.float64).to(dev)
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print(os.environ.get("TTK_LIB_PATH", "default"), "fused matvec (+reduce) us:", round(timeit(lambda: ops.matvec(L, X, R, H, v)), 1),
      " jvp:", round(timeit(lambda: ops.jvp(L, X, R, H, v)), 1), " rhs:", round(timeit(lambda: ops.rhs(L, X, R, H.reshape(S, 1))), 1))
