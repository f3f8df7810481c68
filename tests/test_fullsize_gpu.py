"""The headline core update at full size (54 000 samples, bond 32, 6 polynomial features, D = 6144),
checked through properties that do not need a CPU answer of that size: the compressed Gram against
the matrix-free operator (two independent kernel paths), its trace against a closed form, linearity
in the samples (what sharding relies on), and the residual of the Cholesky solve."""
import pytest
import torch

gpu = pytest.mark.gpu
S, RL, P, RR = 54000, 32, 6, 32


def _problem(dev):
    g = torch.Generator().manual_seed(11)
    L = (torch.rand(S, 1, RL, generator=g, dtype=torch.float64) - 0.5).to(dev)
    x = (torch.rand(S, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
    X = torch.stack([x ** k for k in range(P)], dim=-1).contiguous()      # polynomial basis: feature-pair classes
    R = (torch.rand(S, RR, generator=g, dtype=torch.float64) - 0.5).to(dev)
    H = (torch.rand(S, 1, 1, generator=g, dtype=torch.float64) + 0.5).to(dev)
    return L, X, R, H


@gpu
def test_headline_gram_properties():
    from tensornetworksfork_amd import ops
    from tensornetworksfork_amd.tensor.engine import pair_classes
    dev = torch.device("cuda:0")
    L, X, R, H = _problem(dev)
    rep = pair_classes(X)
    assert rep is not None and len(set(rep)) == 2 * P - 1
    D = RL * P * RR
    ws = ops.gram(L, H, X, R, sym=True, pair_rep=rep)
    A = ops.gram_expand(ws)
    assert ((A - A.T).abs().max() / A.abs().max()).item() == 0.0
    # classes change nothing beyond rounding
    A_all = ops.gram_expand(ops.gram(L, H, X, R, sym=True, pair_rep=None))
    assert ((A - A_all).norm() / A_all.norm()).item() < 1e-13
    # against the matrix-free operator
    v = torch.rand(RL, P, RR, dtype=torch.float64, device=dev) - 0.5
    mv = ops.matvec(L, X, R, H, v).reshape(-1)
    assert ((A @ v.reshape(-1) - mv).norm() / mv.norm()).item() < 1e-11
    # trace = sum_s h |L_s|^2 |X_s|^2 |R_s|^2
    tr = (H.reshape(-1) * (L * L).sum(dim=(1, 2)) * (X * X).sum(dim=1) * (R * R).sum(dim=1)).sum()
    assert abs(A.diagonal().sum().item() - tr.item()) < 1e-11 * abs(tr.item())
    # linear in the samples: two shards add up
    half = S // 2
    G1 = ops.gram(L[:half].contiguous(), H[:half].contiguous(), X[:half].contiguous(), R[:half].contiguous(), sym=True,
                  pair_rep=rep).compressed.clone()
    G2 = ops.gram(L[half:].contiguous(), H[half:].contiguous(), X[half:].contiguous(), R[half:].contiguous(), sym=True,
                  pair_rep=rep).compressed.clone()
    Gf = ops.gram(L, H, X, R, sym=True, pair_rep=rep).compressed
    assert ((G1 + G2 - Gf).norm() / Gf.norm()).item() < 1e-13
    # ridge Cholesky solve: residual of (A/s + 2 eps I) x = -(b/s + 2 eps w)
    ws = ops.gram(L, H, X, R, sym=True, pair_rep=rep)
    gvec = torch.rand(S, 1, dtype=torch.float64, device=dev) - 0.5
    b = ops.rhs(L, X, R, gvec)
    w = torch.rand(RL, P, RR, dtype=torch.float64, device=dev) - 0.5
    eps = 0.25
    scale = ops.gram_scale(ws)
    xsol, info = ops.chol_solve_gram(ws, b, w, scale, 2 * eps)
    assert int(info.item()) == 0
    M = A / scale + 2 * eps * torch.eye(D, dtype=torch.float64, device=dev)
    rhs_vec = -(b.reshape(-1) / scale + 2 * eps * w.reshape(-1))
    assert ((M @ xsol - rhs_vec).norm() / rhs_vec.norm()).item() < 1e-10
    ref = torch.linalg.solve(M, rhs_vec)
    assert ((xsol - ref).norm() / ref.norm()).item() < 1e-9
