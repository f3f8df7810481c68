"""Scratch: right-hand-side kernel, correctness against einsum and timing under rocprofv3 (slab counts via a debug build)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd import _lib, ops
dbg = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libttk_dbg.so")
if os.path.exists(dbg):
    _lib.LIB_PATH = dbg
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (S, C, rl, p, rr) in [(54000, 1, 32, 6, 32), (6000, 1, 32, 6, 32), (1000, 3, 5, 4, 7), (777, 2, 6, 6, 33), (54000, 1, 6, 6, 32),
                          (257, 1, 48, 5, 70), (5000, 10, 24, 12, 24)]:
    L = torch.rand(S, C, rl, dtype=torch.float64, device=dev) - 0.5
    X = torch.rand(S, p, dtype=torch.float64, device=dev) - 0.5
    R = torch.rand(S, rr, dtype=torch.float64, device=dev) - 0.5
    g = torch.rand(S, C, dtype=torch.float64, device=dev) - 0.5
    ref = torch.einsum('sc,scl,sp,sr->lpr', g, L, X, R)
    out = ops.rhs(L, X, R, g)
    err = ((out - ref).norm() / ref.norm()).item()
    out2 = ops.rhs(L, X, R, g, out=out.clone())
    err2 = ((out2 - 2 * ref).norm() / ref.norm()).item()
    print((S, C, rl, p, rr), "err", err, err2)
if os.path.exists(dbg):
    S, C, rl, p, rr = 54000, 1, 32, 6, 32
    L = torch.rand(S, C, rl, dtype=torch.float64, device=dev); X = torch.rand(S, p, dtype=torch.float64, device=dev)
    R = torch.rand(S, rr, dtype=torch.float64, device=dev); g = torch.rand(S, C, dtype=torch.float64, device=dev)
    for ns in (128, 256, 384, 512, 768, 1024):
        _lib.lib().ttk_debug_rhs_slabs(ns)
        for _ in range(4):
            ops.rhs(L, X, R, g)
        torch.cuda.synchronize()
