"""Scratch: core update at bonds above the headline (48, 64): Gram against the matrix-free operator, residual of the solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd import ops
from tensornetworksfork_amd.tensor.engine import pair_classes
dev = torch.device("cuda:0")
S, P = 20000, 6
for r in [int(a) for a in sys.argv[1:]] or [48, 64]:
    g = torch.Generator(device=dev).manual_seed(r)
    L = torch.rand(S, 1, r, generator=g, dtype=torch.float64, device=dev) - 0.5
    x = torch.rand(S, generator=g, dtype=torch.float64, device=dev) * 2 - 1
    X = torch.stack([x ** d for d in range(P)], dim=-1).contiguous()
    R = torch.rand(S, r, generator=g, dtype=torch.float64, device=dev) - 0.5
    H = torch.full((S, 1, 1), 2.0, dtype=torch.float64, device=dev)
    D = r * P * r
    rep = pair_classes(X)
    t0 = time.perf_counter()
    ws = ops.gram(L, H, X, R, sym=True, pair_rep=rep)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    v = torch.rand(r, P, r, generator=g, dtype=torch.float64, device=dev) - 0.5
    mv = ops.matvec(L, X, R, H, v).reshape(-1) if ops.matvec_fused_supported(S, 1, r, P, r) or True else None
    b = ops.rhs(L, X, R, torch.rand(S, 1, generator=g, dtype=torch.float64, device=dev) - 0.5)
    w = torch.rand(r, P, r, generator=g, dtype=torch.float64, device=dev) - 0.5
    scale = ops.gram_scale(ws)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    xs, info = ops.chol_solve_gram(ws, b, w, scale, 0.5)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    # residual through the matrix-free operator: (A/s + ridge) x + (b/s + ridge w) = 0
    Ax = ops.matvec(L, X, R, H, xs.reshape(r, P, r)).reshape(-1)
    res = Ax / scale + 0.5 * xs + b.reshape(-1) / scale + 0.5 * w.reshape(-1)
    rhsn = (b.reshape(-1) / scale + 0.5 * w.reshape(-1)).norm()
    print(f"r={r} D={D}: gram {1e3*(t1-t0):.1f} ms, solve {1e3*(t3-t2):.1f} ms, info {int(info.item())}, residual {(res.norm()/rhsn).item():.2e}", flush=True)
    if D <= 16000:
        A = ops.gram_expand(ws)
        print("   A v vs matvec", ((A @ v.reshape(-1) - mv).norm() / mv.norm()).item())
        del A
