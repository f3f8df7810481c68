"""Scratch: time the Gram kernel (HIP events inside the library) for a given build of the library."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensornetworksfork_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
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
for name, r in (("classed", rep), ("full", None)):
    ws = ops.gram(L, H, X, R, sym=True, pair_rep=r)
    ops.gram_timing(True)
    for _ in range(5):
        ws = ops.gram(L, H, X, R, sym=True, ws=ws, pair_rep=r)
    torch.cuda.synchronize()
    ms = ops.gram_timing_read()
    ops.gram_timing(False)
    nx = 21 if r is None else len(set(r))
    fl = 2.0 * S * 528 * nx * 528
    print(name, "ms", [round(m, 3) for m in ms], "TF/s", round(fl / (sum(ms) / len(ms)) * 1e-9, 1), flush=True)
