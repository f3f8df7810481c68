"""Scratch: the solve at order n with the backward substitution as one launch per group (200) / one launch (201)."""
import sys
import torch
sys.path.insert(0, ".")
from tensornetworksfork_amd import ops  # noqa: E402
from tools.time_chol import problem  # noqa: E402
dev = torch.device("cuda:0")
for n in [int(a) for a in sys.argv[1:]] or [2048, 4096, 6144]:
    A, b = problem(n, dev)
    ops.chol_mode(1)
    for mode in (200, 201, 200, 201):
        ops.chol_mode(mode)
        ops.chol_solve_dense(A, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.chol_solve_dense(A, b)
        e1.record()
        torch.cuda.synchronize()
        print(f"n={n} back mode {mode}: {e0.elapsed_time(e1) / 10:.3f} ms per solve", flush=True)
    ops.chol_mode(201)
    ops.chol_mode(-1)
