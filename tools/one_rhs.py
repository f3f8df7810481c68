"""Five right-hand-side launches of one shape, for counter collection: rocprofv3 --pmc ... -- python tools/one_rhs.py S C rl p rr"""
import sys, torch
sys.path.insert(0, ".")
from tensornetworksfork_amd import ops
S, C, rl, p, rr = [int(v) for v in sys.argv[1:6]]
dev = torch.device("cuda:0")
L = torch.rand(S, C, rl, dtype=torch.float64, device=dev); X = torch.rand(S, p, dtype=torch.float64, device=dev)
R = torch.rand(S, rr, dtype=torch.float64, device=dev); g = torch.rand(S, C, dtype=torch.float64, device=dev)
for _ in range(5):
    ops.rhs(L, X, R, g)
torch.cuda.synchronize()
