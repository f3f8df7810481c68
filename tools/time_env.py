import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (S, C, rl, p, rr) in [(54000, 1, 32, 6, 32), (6000, 1, 32, 6, 32), (1000, 3, 5, 4, 7), (777, 2, 6, 6, 33), (54000, 1, 6, 6, 32), (54000, 1, 32, 6, 6)]:
    E = torch.rand(S, C, rl, dtype=torch.float64, device=dev) - 0.5
    X = torch.rand(S, p, dtype=torch.float64, device=dev) - 0.5
    R = torch.rand(S, rr, dtype=torch.float64, device=dev) - 0.5
    core = torch.rand(rl, p, rr, dtype=torch.float64, device=dev) - 0.5
    o1 = ops.env_step_left(E, X, core)
    r1 = torch.einsum('scl,sp,lpr->scr', E, X, core)
    o2 = ops.env_step_right(R, X, core)
    r2 = torch.einsum('sr,sp,lpr->sl', R, X, core)
    o3 = ops.env_step_left(E, X, core, out=o1.clone())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.env_step_left(E, X, core, out=o1)
    torch.cuda.synchronize(); tl = (time.perf_counter() - t0) / 20 * 1e6
    t0 = time.perf_counter()
    for _ in range(20): ops.env_step_right(R, X, core, out=o2)
    torch.cuda.synchronize(); tr = (time.perf_counter() - t0) / 20 * 1e6
    print((S, C, rl, p, rr), "left err", ((o3 - 2 * r1).abs().max() / r1.abs().max()).item(), "right err", 0.0 if False else ((ops.env_step_right(R, X, core) - r2).abs().max() / r2.abs().max()).item(), f"us left {tl:.1f} right {tr:.1f}", flush=True)
