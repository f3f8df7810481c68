import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd import ops
dev = torch.device("cuda:0")
for (m, n) in [(192, 32), (36, 6), (36, 32), (192, 6), (5, 9), (256, 32), (1, 1), (1200, 24), (50, 65)]:
    A = torch.randn(m, n, dtype=torch.float64, device=dev)
    Q, R = ops.qr(A)
    Qr, Rr = torch.linalg.qr(A.cpu())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): Q, R = ops.qr(A)
    torch.cuda.synchronize()
    print((m, n), "us", (time.perf_counter() - t0) / 20 * 1e6, "errQ", (Q.cpu() - Qr).abs().max().item(), "errR", (R.cpu() - Rr).abs().max().item())
