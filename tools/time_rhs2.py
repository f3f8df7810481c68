"""Scratch: J^T g of a many-feature site (cfg-4 carriage: C=9, rl=24, p=50, rr=24): two-stage (g.L, outer rows, GEMM)
against the right-hand-side kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd import ops
from tensornetworksfork_amd.tensor import engine as E
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (S, C, rl, p, rr) in [(54000, 9, 24, 50, 24), (54000, 9, 1, 50, 24), (54000, 9, 24, 50, 1)]:
    L = torch.rand(S, C, rl, dtype=torch.float64, device=dev) - 0.5
    X = torch.rand(S, p, dtype=torch.float64, device=dev) - 0.5
    R = torch.rand(S, rr, dtype=torch.float64, device=dev) - 0.5
    g = torch.rand(S, C, dtype=torch.float64, device=dev) - 0.5
    def two():
        GL = ops.gl(g, L)
        W = ops.kr_rows(GL.unsqueeze(1), E._ones_col(S, dev), R).reshape(S, rl * rr)
        return E._xt_times(X, W).reshape(p, rl, rr).permute(1, 0, 2).contiguous()
    def one():
        return ops.rhs(L, X, R, g)
    for name, f in (("two-stage", two), ("kernel", one)):
        try:
            ref = f()
        except Exception as ex:
            print((S, C, rl, p, rr), name, "unsupported:", str(ex)[:60]); continue
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): out = f()
        torch.cuda.synchronize()
        print((S, C, rl, p, rr), name, "ms", round((time.perf_counter() - t0) / 10 * 1e3, 3), "norm", float(out.norm()), flush=True)
