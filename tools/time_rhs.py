"""Time J^T g of one Khatri-Rao term on both routes (three-factor kernel, features-first kernel):
python tools/time_rhs.py S C rl p rr [S C rl p rr ...]"""
import sys
import torch
sys.path.insert(0, ".")
from tensornetworksfork_amd import ops


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


a = [int(v) for v in sys.argv[1:]]
dev = torch.device("cuda:0")
for k in range(0, len(a), 5):
    S, C, rl, p, rr = a[k:k + 5]
    L = torch.rand(S, C, rl, dtype=torch.float64, device=dev)
    X = torch.rand(S, p, dtype=torch.float64, device=dev)
    R = torch.rand(S, rr, dtype=torch.float64, device=dev)
    g = torch.rand(S, C, dtype=torch.float64, device=dev)
    line = "%s three-factor %.1f us" % ((S, C, rl, p, rr), timed(lambda: ops.rhs(L, X, R, g)))
    if 16 <= p <= 256 and rr <= 128:
        try:
            line += ", features-first %.1f us" % timed(lambda: ops.rhs_feature(L, X, R, g))
        except Exception as e:     # shapes that route does not take
            line += ", features-first n/a (%s)" % type(e).__name__
    print(line)
