"""Scratch timing of individual kernels at headline sizes (not the bench contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd import ops

dev = torch.device("cuda:0")
S, C, rl, p, rr = 54000, 1, 32, 6, 32
D = rl * p * rr
torch.manual_seed(0)
L = torch.rand(S, C, rl, dtype=torch.float64, device=dev) - 0.5
X = torch.rand(S, p, dtype=torch.float64, device=dev) - 0.5
R = torch.rand(S, rr, dtype=torch.float64, device=dev) - 0.5
H = torch.full((S, 1, 1), 2.0, dtype=torch.float64, device=dev)
g = torch.rand(S, C, dtype=torch.float64, device=dev)
core = torch.rand(rl, p, rr, dtype=torch.float64, device=dev)


def timeit(name, fn, n=10, flops=None, bytes_=None):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    extra = ""
    if flops: extra += f"  {flops / ms * 1e-9:.2f} TFLOP/s"
    if bytes_: extra += f"  {bytes_ / ms * 1e-6:.1f} GB/s"
    print(f"{name:28s} {ms:9.4f} ms{extra}", flush=True)
    return ms

ws = ops.gram(L, H, X, R, sym=True)
gd_flops = 2.0 * S * 528 * 21 * 528
timeit("hess_left", lambda: ops.hess_left(H, L))
timeit("gram (compressed)", lambda: ops.gram(L, H, X, R, sym=True, ws=ws), n=5, flops=gd_flops)
scale = ops.gram_scale(ws)
A = torch.empty(D, D, dtype=torch.float64, device=dev)
timeit("gram_scale", lambda: ops.gram_scale(ws))
timeit("gram_expand", lambda: ops.gram_expand(ws, scale, 1.0, out=A), bytes_=D * D * 8.0)
timeit("rhs", lambda: ops.rhs(L, X, R, g), flops=2.0 * S * D)
timeit("env_step_left", lambda: ops.env_step_left(L, X, core), flops=2.0 * S * D, bytes_=S * (rl + p + rr) * 8.0)
timeit("env_step_right", lambda: ops.env_step_right(R, X, core), flops=2.0 * S * D)
v = torch.rand(rl, p, rr, dtype=torch.float64, device=dev)
timeit("matvec (jvp+rhs)", lambda: ops.matvec(L, X, R, H, v), flops=4.0 * S * D)
timeit("predict", lambda: ops.predict(L, R))
w = torch.rand(rl, p, rr, dtype=torch.float64, device=dev)
bb = ops.rhs(L, X, R, g)
timeit("chol_solve_gram (own)", lambda: ops.chol_solve_gram(ws, bb, w, scale, 1.0), n=5, flops=D ** 3 / 3)
timeit("torch cholesky 6144", lambda: torch.linalg.cholesky(A), n=3, flops=D ** 3 / 3)
Lc = torch.linalg.cholesky(A)
b = torch.rand(D, 1, dtype=torch.float64, device=dev)
timeit("torch cholesky_solve", lambda: torch.cholesky_solve(b, Lc), n=3)
timeit("torch cholesky_ex", lambda: torch.linalg.cholesky_ex(A), n=3, flops=D ** 3 / 3)
timeit("torch linalg.solve (LU)", lambda: torch.linalg.solve(A, b), n=2, flops=2 * D ** 3 / 3)
M = torch.rand(192, 32, dtype=torch.float64, device=dev)
timeit("torch qr 192x32", lambda: torch.linalg.qr(M), n=5)
B1 = torch.rand(D, D, dtype=torch.float64, device=dev)
timeit("torch dgemm 6144^3", lambda: B1 @ A, n=3, flops=2.0 * D ** 3)
