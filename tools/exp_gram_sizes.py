"""Scratch: Gram kernel time at the per-rank sample counts of 1, 2, 4, 8-way sharding of the headline problem."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd import ops
from tensornetworksfork_amd.tensor.engine import pair_classes
dev = torch.device("cuda:0")
C, rl, p, rr = 1, 32, 6, 32
torch.manual_seed(0)
for S in (54000, 27000, 13500, 6750):
    L = torch.rand(S, C, rl, dtype=torch.float64, device=dev) - 0.5
    x = torch.rand(S, dtype=torch.float64, device=dev) * 2 - 1
    X = torch.stack([x ** d for d in range(p)], dim=-1).contiguous()
    R = torch.rand(S, rr, dtype=torch.float64, device=dev) - 0.5
    H = torch.full((S, 1, 1), 2.0, dtype=torch.float64, device=dev)
    rep = pair_classes(X)
    ws = ops.gram(L, H, X, R, sym=True, pair_rep=rep)
    ops.gram_timing(True)
    for _ in range(6):
        ws = ops.gram(L, H, X, R, sym=True, ws=ws, pair_rep=rep)
    torch.cuda.synchronize()
    ms = ops.gram_timing_read()
    ops.gram_timing(False)
    best = min(ms)
    print("S", S, "ms", [round(m, 3) for m in ms], "TF/s", round(2.0 * S * 528 * 11 * 528 / best * 1e-9, 1), "x ideal", round(best / (5.49 * S / 54000), 2), flush=True)
