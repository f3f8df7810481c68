"""Scan of the slab count of the right-hand-side kernel (needs a -DTTK_DEBUG_HOOKS build: TTK_LIB_PATH=tools/libttk_dbg.so)."""
import sys, ctypes, torch
sys.path.insert(0, ".")
from tensornetworksfork_amd import _lib, ops
L_ = _lib.lib()
def timed(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n
dev = torch.device("cuda:0")
for (S, C, rl, p, rr) in [(54000,1,10,50,10),(54000,1,16,50,16),(54000,1,32,6,32),(54000,1,8,20,8),(54000,1,1,6,32),(6000,1,32,6,32)]:
    L = torch.rand(S, C, rl, dtype=torch.float64, device=dev); X = torch.rand(S, p, dtype=torch.float64, device=dev)
    R = torch.rand(S, rr, dtype=torch.float64, device=dev); g = torch.rand(S, C, dtype=torch.float64, device=dev)
    out = []
    for ns in (64, 128, 192, 256, 384, 512, 768):
        L_.ttk_debug_rhs_slabs(ns)
        out.append("%d: %.1f" % (ns, timed(lambda: ops.rhs(L, X, R, g))))
    print((S, C, rl, p, rr), "  ".join(out))
