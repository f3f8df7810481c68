"""One many-feature site, right-environment step only (for counter runs): python tools/time_env_right_one.py [f32]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd import ops
dev = torch.device("cuda:0")
mm32 = len(sys.argv) > 1 and sys.argv[1] == "f32"
S, rl, p, rr = 50000, 48, 65, 48
g = torch.Generator().manual_seed(1)
R = torch.rand((S, rr), generator=g, dtype=torch.float64).to(dev)
X = torch.rand((S, p), generator=g, dtype=torch.float64).to(dev)
core = torch.rand((rl, p, rr), generator=g, dtype=torch.float64).to(dev)
for _ in range(5):
    ops.env_step_right(R, X, core, mm32=mm32)
torch.cuda.synchronize()
