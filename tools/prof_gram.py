"""Scratch: run the Gram kernel a few times for rocprofv3 (headline shape, polynomial features
as in the benchmark, so the feature-pair classes are active)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd import ops
from tensornetworksfork_amd.tensor.engine import pair_classes
dev = torch.device("cuda:0")
S, C, rl, p, rr = 54000, 1, 32, 6, 32
torch.manual_seed(0)
L = torch.rand(S, C, rl, dtype=torch.float64, device=dev) - 0.5
x = torch.rand(S, dtype=torch.float64, device=dev) * 2 - 1
X = torch.stack([x ** d for d in range(p)], dim=-1).contiguous()
R = torch.rand(S, rr, dtype=torch.float64, device=dev) - 0.5
H = torch.full((S, 1, 1), 2.0, dtype=torch.float64, device=dev)
rep = pair_classes(X)
ws = None
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(n):
    ws = ops.gram(L, H, X, R, sym=True, ws=ws, pair_rep=rep)
torch.cuda.synchronize()
print(len(set(rep)), ws.compressed[:4].tolist())
