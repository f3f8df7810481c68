"""Scratch: old vs new Cholesky panel kernel - result agreement and time."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get('TTK_DBG', 'libttk_dbg.so'))
from tensornetworksfork_amd import ops
dev = torch.device("cuda:0")
lib = _lib.lib()
for (S, C, rl, p, rr) in [(3000, 1, 8, 5, 8), (4000, 2, 7, 5, 7), (54000, 1, 32, 6, 32)]:
    torch.manual_seed(0)
    L = torch.rand(S, C, rl, dtype=torch.float64, device=dev) - 0.5
    X = torch.rand(S, p, dtype=torch.float64, device=dev) - 0.5
    R = torch.rand(S, rr, dtype=torch.float64, device=dev) - 0.5
    H = (2.0 * torch.eye(C, dtype=torch.float64, device=dev)).expand(S, C, C).contiguous()
    g = torch.rand(S, C, dtype=torch.float64, device=dev)
    w = torch.rand(rl, p, rr, dtype=torch.float64, device=dev)
    ws = ops.gram(L, H, X, R, sym=True)
    b = ops.rhs(L, X, R, g)
    scale = ops.gram_scale(ws)
    A = ops.gram_expand(ws, scale=scale, ridge=1e-3)
    ref = torch.linalg.solve(A, ops.solve_rhs(b, w, scale, 1e-3))
    for variant in (2048,):
        lib.ttk_debug_group(variant)
        x, info = ops.chol_solve_gram(ws, b, w, scale, 1e-3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            x, info = ops.chol_solve_gram(ws, b, w, scale, 1e-3)
        e1.record()
        torch.cuda.synchronize()
        print((S, C, rl, p, rr), "variant", variant, "info", int(info.item()), "rel err vs LU",
              ((x - ref).norm() / ref.norm()).item(), "ms", e0.elapsed_time(e1) / 5, flush=True)
