import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libttk_dbg.so')
from tensornetworksfork_amd import ops
dev = torch.device("cuda:0")
lib = _lib.lib()
S, C, rl, p, rr = 54000, 1, 32, 6, 32
torch.manual_seed(0)
L = torch.rand(S, C, rl, dtype=torch.float64, device=dev) - 0.5
X = torch.rand(S, p, dtype=torch.float64, device=dev) - 0.5
R = torch.rand(S, rr, dtype=torch.float64, device=dev) - 0.5
H = torch.full((S, 1, 1), 2.0, dtype=torch.float64, device=dev)
g = torch.rand(S, C, dtype=torch.float64, device=dev)
w = torch.rand(rl, p, rr, dtype=torch.float64, device=dev)
ws = ops.gram(L, H, X, R, sym=True)
b = ops.rhs(L, X, R, g)
scale = ops.gram_scale(ws)
for _ in range(3):
    x, info = ops.chol_solve_gram(ws, b, w, scale, 1.0)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
lib.ttk_debug_read.argtypes = [ctypes.c_void_p]
print(lib.ttk_debug_read(ctypes.cast(buf, ctypes.c_void_p)))
for o in (0, 16):
    t = [buf[o + i] for i in range(12)]
    print("j =", 40 + o // 16, "phase deltas (10 ns ticks):", [t[i + 1] - t[i] for i in range(11)], "total", t[11] - t[0])
