"""Scratch: time the trailing-update kernel alone at several trailing sizes."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get('TTK_DBG', 'libttk_dbg.so'))
lib = _lib.lib()
dev = torch.device("cuda:0")
nb = 96
ld = nb * 64
nrows = (nb + 1) * 64
aug = torch.rand(nrows, ld, dtype=torch.float64, device=dev) * 1e-3
st = torch.cuda.current_stream().cuda_stream
vp = ctypes.c_void_p
lib.ttk_debug_syrk.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for TS in (64,):
  lib.ttk_debug_syrk_tile(TS)
  for K in (128,):
    for o0 in (K, 1024, 2048, 3072, 4096, 5120, 5888):
        for _ in range(2):
            lib.ttk_debug_syrk(aug.data_ptr(), ld, 0, K, o0, nrows, ld, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.ttk_debug_syrk(aug.data_ptr(), ld, 0, K, o0, nrows, ld, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        n = ld - o0
        flops = (n * (n + 128) / 2 + 64 * n) * K * 2     # lower tiles incl. diagonal tiles + the extra block row
        print(f"TS={TS} K={K} o0={o0} n={n} {ms * 1e3:.1f} us  {flops / ms * 1e-9:.1f} TF/s", flush=True)
