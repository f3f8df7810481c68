"""Scratch: run the Gram kernel a few times for rocprofv3 (headline shape)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd import ops
dev = torch.device("cuda:0")
S, C, rl, p, rr = 54000, 1, 32, 6, 32
torch.manual_seed(0)
L = torch.rand(S, C, rl, dtype=torch.float64, device=dev) - 0.5
X = torch.rand(S, p, dtype=torch.float64, device=dev) - 0.5
R = torch.rand(S, rr, dtype=torch.float64, device=dev) - 0.5
H = torch.full((S, 1, 1), 2.0, dtype=torch.float64, device=dev)
ws = None
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(n):
    ws = ops.gram(L, H, X, R, sym=True, ws=ws)
torch.cuda.synchronize()
print(ws.compressed[:4].tolist())
