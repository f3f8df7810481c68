"""Scratch: times the conv-bond kernels at the shapes of the bond-4 MNIST sweep (54000 samples, 50 patches of 17
pixels, bond 10, 4 channels, 10 outputs).  python tools/time_bond.py [filter|jv|rhs|patch ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd import ops

dev = torch.device("cuda:0")
S, C, rl, p, rr, ca, cb, px = 54000, 10, 10, 50, 10, 4, 4, 17
g = torch.Generator().manual_seed(0)
mk = lambda *sh: torch.randn(*sh, dtype=torch.float64, generator=g).to(dev)   # noqa: E731
which = sys.argv[1:] or ["filter", "jv", "rhs", "patch"]


def timeit(name, fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1) / reps * 1e3:9.1f} us", flush=True)


L = mk(ca, S, C, rl)
if "filter" in which:
    n = px * cb
    Z, H, v, gg = mk(S, rl, n), mk(S, C, C), mk(ca * n), mk(S, C)
    timeit("filter_matvec n=68 ca=4", lambda: ops.filter_matvec(L, Z, H, v))
    timeit("filter_jv", lambda: ops.filter_jv(L, Z, v))
    timeit("filter_rhs", lambda: ops.filter_rhs(L, Z, gg))
    Z1, v1 = mk(S, rl, px), mk(ca * px)
    timeit("filter_matvec n=17 ca=4", lambda: ops.filter_matvec(L, Z1, H, v1))
if "jv" in which or "rhs" in which:
    G, R, v, gg = mk(ca, cb, S, p), mk(cb, S, rr), mk(rl, p, rr), mk(S, C)
    if "jv" in which:
        timeit("bond_jv ca=cb=4", lambda: ops.bond_jv(L, G, R, v))
    if "jv" in which and "rhs" in which:
        Hm = mk(S, C, C)
        timeit("bond_matvec ca=cb=4", lambda: ops.bond_matvec(L, G, R, Hm, v))
    if "rhs" in which:
        timeit("bond_rhs ca=cb=4", lambda: ops.bond_rhs(L, G, R, gg))
if "patch" in which:
    U, X = mk(S, rl, p), mk(S, p, px)
    Zt = torch.empty((S, rl, 1, px, cb), dtype=torch.float64, device=dev)
    timeit("patch_contract (one b)", lambda: ops.patch_contract(U, X, Zt, 0, 1))
