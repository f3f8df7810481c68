"""Scratch: single-launch Cholesky at order 6144 against the number of compute units kept for the diagonal chain."""
import sys
import torch
sys.path.insert(0, ".")
from tensornetworksfork_amd import ops
from tools.time_chol import problem
dev = torch.device("cuda:0")
A, b = problem(6144, dev)
ops.chol_mode(1)
for u in [0, 12, 16, 20, 24, 28, 32, 40, 48]:
    ops.chol_mode(100 + u)
    for _ in range(3):
        ops.chol_solve_dense(A, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.chol_solve_dense(A, b)
    e1.record()
    torch.cuda.synchronize()
    print(f"chain units {u or 'auto'}: {e0.elapsed_time(e1) / 20:.3f} ms per solve", flush=True)
ops.chol_mode(100)
ops.chol_mode(-1)
