"""Cholesky solve: single-launch factorisation against the multi-launch one and torch, with timings.
usage: python tools/time_chol.py [n ...]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from tensornetworksfork_amd import ops  # noqa: E402


def problem(n, dev, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    M = torch.randn(n, n + 64, generator=g, dtype=torch.float64, device=dev)
    A = M @ M.T / (n + 64) + 0.05 * torch.eye(n, dtype=torch.float64, device=dev)
    b = torch.randn(n, generator=g, dtype=torch.float64, device=dev)
    return A, b


def run(n, dev):
    A, b = problem(n, dev)
    scale = A.diagonal().abs().mean()
    ref = torch.linalg.solve(A / scale, -b / scale)
    out = {}
    for mode in (0, 1):
        ops.chol_mode(mode)
        x, info = ops.chol_solve_dense(A, b)
        torch.cuda.synchronize()
        err = ((x - ref).norm() / ref.norm()).item()
        res = (((A / scale) @ x + b / scale).norm() / (b / scale).norm()).item()
        xs = [ops.chol_solve_dense(A, b)[0] for _ in range(5)]
        same = all(torch.equal(xs[0], y) for y in xs)
        torch.cuda.synchronize()
        reps = 10
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            ops.chol_solve_dense(A, b)
        ev1.record()
        torch.cuda.synchronize()
        out[mode] = (err, res, int(info.item()), same, ev0.elapsed_time(ev1) / reps)
    ops.chol_mode(-1)
    print(f"n={n:5d}  multi: err {out[0][0]:.1e} res {out[0][1]:.1e} info {out[0][2]} {out[0][4]:.3f} ms | "
          f"single: err {out[1][0]:.1e} res {out[1][1]:.1e} info {out[1][2]} same {out[1][3]} {out[1][4]:.3f} ms", flush=True)


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    for n in [int(a) for a in sys.argv[1:]] or [64, 100, 128, 200, 512, 1000, 2048, 4096, 6144]:
        t = time.time()
        run(n, dev)
