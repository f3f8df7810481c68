"""Scratch: right-hand side of a site with a narrow left bond and many columns, three-factor kernel against
GL x R rows + one transposed product over the samples."""
import sys, torch
sys.path.insert(0, ".")
from tensornetworksfork_amd import ops
def timed(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n
dev = torch.device("cuda:0")
for (S, C, rl, p, rr) in [(54000,9,9,50,24),(54000,1,10,50,10),(54000,1,16,65,16),(50000,9,9,65,48),(54000,1,16,50,16)]:
    L = torch.rand(S, C, rl, dtype=torch.float64, device=dev); X = torch.rand(S, p, dtype=torch.float64, device=dev)
    R = torch.rand(S, rr, dtype=torch.float64, device=dev); g = torch.rand(S, C, dtype=torch.float64, device=dev)
    one = torch.ones(S, 1, dtype=torch.float64, device=dev)
    def alt():
        GL = ops.gl(g, L)
        W = ops.kr_rows(GL.unsqueeze(1), one, R).reshape(S, rl * rr)
        return ops.gemm_tn(X, W).reshape(p, rl, rr).permute(1, 0, 2).contiguous()
    a, b = ops.rhs(L, X, R, g), alt()
    err = ((a - b).norm() / a.norm()).item()
    print((S, C, rl, p, rr), "three-factor %.1f us, rows + transposed product %.1f us, feature %.1f us, diff %.1e" % (
        timed(lambda: ops.rhs(L, X, R, g)), timed(alt), timed(lambda: ops.rhs_feature(L, X, R, g)), err), flush=True)
