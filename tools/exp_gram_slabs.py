"""Scratch: Gram kernel time against the forced slab count (debug build of the library)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensornetworksfork_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libttk_dbg.so")
import torch
from tensornetworksfork_amd import ops
from tensornetworksfork_amd.tensor.engine import pair_classes
dev = torch.device("cuda:0")
S, C, rl, p, rr = 54000, 1, 32, 6, 32
torch.manual_seed(0)
L = torch.rand(S, C, rl, dtype=torch.float64, device=dev) - 0.5
x = torch.rand(S, dtype=torch.float64, device=dev) * 2 - 1
X = torch.stack([x ** d for d in range(p)], dim=-1).contiguous()
R = torch.rand(S, rr, dtype=torch.float64, device=dev) - 0.5
H = torch.full((S, 1, 1), 2.0, dtype=torch.float64, device=dev)
rep = pair_classes(X)
ref = None
for ns in [int(a) for a in sys.argv[1:]]:
    _lib.lib().ttk_debug_gram_slabs(ns)
    ws = ops.gram(L, H, X, R, sym=True, pair_rep=rep)
    ops.gram_timing(True)
    for _ in range(4):
        ws = ops.gram(L, H, X, R, sym=True, ws=ws, pair_rep=rep)
    torch.cuda.synchronize()
    ms = ops.gram_timing_read()
    ops.gram_timing(False)
    A = ws.compressed.clone()
    if ref is None:
        ref = A
    print("slabs", ns, "ms", [round(m, 3) for m in ms], "rel diff", float((A - ref).norm() / ref.norm()), flush=True)
    del ws
