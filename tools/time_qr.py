"""Time ops.qr at given shapes: python tools/time_qr.py m n [m n ...]"""
import sys
import torch
sys.path.insert(0, ".")
from tensornetworksfork_amd import ops

a = [int(v) for v in sys.argv[1:]]
dev = torch.device("cuda:0")
for k in range(0, len(a), 2):
    m, n = a[k:k + 2]
    A = torch.rand(m, n, dtype=torch.float64, device=dev)
    for _ in range(5):
        ops.qr(A)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.qr(A)
    e1.record()
    torch.cuda.synchronize()
    print((m, n), "%.1f us" % (e0.elapsed_time(e1) * 20))
