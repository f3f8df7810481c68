"""Time the right-environment step and the features-first right-hand side of many-feature sites, fp64 and
float32 matrix arithmetic (GPU).  usage: python tools/time_env_right.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd import ops

dev = torch.device("cuda:0")


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (S, C, rl, p, rr) in [(50000, 10, 48, 65, 48), (54000, 9, 24, 50, 24), (50000, 10, 48, 65, 1), (50000, 10, 16, 40, 16),
                          (50000, 10, 64, 65, 64), (50000, 10, 1, 65, 48), (54000, 9, 1, 50, 24), (54000, 9, 24, 50, 1),
                          (54000, 9, 8, 20, 8), (54000, 9, 32, 16, 32), (20000, 3, 12, 30, 12)]:
    g = torch.Generator().manual_seed(1)
    R = torch.rand((S, rr), generator=g, dtype=torch.float64).to(dev)
    X = torch.rand((S, p), generator=g, dtype=torch.float64).to(dev)
    core = torch.rand((rl, p, rr), generator=g, dtype=torch.float64).to(dev)
    L = torch.rand((S, C, rl), generator=g, dtype=torch.float64).to(dev)
    gr = torch.rand((S, C), generator=g, dtype=torch.float64).to(dev)
    fl = 2.0 * S * p * rl * rr
    for mm32 in (False, True):
        t = timed(lambda: ops.env_step_right(R, X, core, mm32=mm32))
        line = f"S={S} {rl}x{p}x{rr} {'f32' if mm32 else 'f64'}: env_right {t:7.1f} us {fl / t * 1e-6:6.1f} TF/s"
        if p >= 16:
            t2 = timed(lambda: ops.rhs_feature(L, X, R, gr, mm32=mm32))
            line += f" | rhs_feature {t2:7.1f} us {fl / t2 * 1e-6:6.1f} TF/s"
            if not mm32 and ops.rhs_supported(S, C, rl, p, rr):
                t3 = timed(lambda: ops.rhs(L, X, R, gr))
                line += f" | three-factor rhs {t3:7.1f} us"
        print(line, flush=True)
