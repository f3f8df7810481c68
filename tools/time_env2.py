import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd import ops
dev = torch.device("cuda:0")
rl, p, rr, C = 32, 6, 32, 1
core = torch.rand(rl, p, rr, dtype=torch.float64, device=dev)
for S in (1000, 6000, 16000, 32768, 54000, 108000):
    E = torch.rand(S, C, rl, dtype=torch.float64, device=dev); X = torch.rand(S, p, dtype=torch.float64, device=dev)
    o = ops.env_step_left(E, X, core)
    for _ in range(3): ops.env_step_left(E, X, core, out=o)
    torch.cuda.synchronize()
