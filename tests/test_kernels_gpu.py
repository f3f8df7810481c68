"""HIP kernels (through the C ABI) against the CPU oracle on seeded inputs."""
import pytest
import torch

from oracle import tt_oracle as O
from tests import _golden as G

gpu = pytest.mark.gpu
TOL = 1e-10  # fp64 kernels vs fp64 oracle, relative Frobenius error


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _rand(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1


SHAPES = [  # S, C, rl, p, rr
    (96, 1, 4, 3, 4), (130, 3, 6, 4, 6), (77, 2, 1, 5, 4), (64, 1, 4, 3, 1), (513, 1, 32, 6, 32),
    (200, 2, 7, 5, 9), (1, 1, 3, 2, 3), (300, 1, 24, 50, 24), (100, 9, 9, 17, 20), (257, 1, 48, 5, 70),
    # many ragged 16-row tiles per wave of the resident-core kernel, 1..4 column tiles
    (5000, 3, 8, 3, 40), (40999, 1, 8, 4, 64), (2001, 2, 16, 2, 17), (70001, 1, 32, 6, 32), (9000, 1, 40, 3, 8),
    # bonds of 16 / 32 with at most 8 features: operands held in registers, both directions, 1..4 column tiles
    (4099, 3, 16, 8, 48), (3001, 2, 32, 5, 16), (1500, 1, 64, 3, 32), (2500, 1, 16, 7, 32), (777, 1, 32, 8, 16),
    # one output, square bonds of 16 / 32, at most 8 features: the one-pass matrix-free product (ragged slabs,
    # a single partial chunk, one feature, the widest core)
    (3001, 1, 16, 5, 16), (1000, 1, 32, 8, 32), (9, 1, 16, 1, 16), (16411, 1, 32, 3, 32), (33, 1, 16, 8, 16),
    # one row tile of the right-hand side, six column tiles per wave, the last block ragged
    (1500, 9, 10, 50, 10), (700, 1, 16, 3, 37),
]


@gpu
@pytest.mark.parametrize("S,C,rl,p,rr", SHAPES)
def test_env_steps(S, C, rl, p, rr):
    from tensornetworksfork_amd import ops
    dev = _dev()
    E, X, core, R = _rand((S, C, rl), 1), _rand((S, p), 2), _rand((rl, p, rr), 3), _rand((S, rr), 4)
    out = ops.env_step_left(E.to(dev), X.to(dev), core.to(dev))
    assert G.rel_err(out, O.env_left_step(E, core, X)) < TOL
    out = ops.env_step_right(R.to(dev), X.to(dev), core.to(dev))
    assert G.rel_err(out, O.env_right_step(R, core, X)) < TOL
    Lk = _rand((S, C, rr), 5)
    y = ops.predict(Lk.to(dev), R.to(dev))
    assert G.rel_err(y, torch.einsum('scr,sr->sc', Lk, R)) < TOL
    # accumulating form: out += step
    base_l, base_r = _rand((S, C, rr), 6), _rand((S, rl), 7)
    out = ops.env_step_left(E.to(dev), X.to(dev), core.to(dev), out=base_l.to(dev))
    assert G.rel_err(out, base_l + O.env_left_step(E, core, X)) < TOL
    out = ops.env_step_right(R.to(dev), X.to(dev), core.to(dev), out=base_r.to(dev))
    assert G.rel_err(out, base_r + O.env_right_step(R, core, X)) < TOL


@gpu
@pytest.mark.parametrize("S,C,rl,p,rr", [(64, 1, 33, 400, 2), (300, 2, 64, 257, 1), (129, 1, 24, 257, 64), (65, 9, 24, 256, 31),
                                         (200, 3, 16, 193, 63), (64, 1, 1, 256, 1), (257, 1, 128, 256, 1), (70, 1, 96, 400, 7)])
def test_sites_with_hundreds_of_features(S, C, rl, p, rr):
    """More features than the features-first kernels keep in LDS (192), with bonds that make a sample row too wide
    for the streaming kernels' LDS image as well: the steps go feature block by feature block, the right-hand side
    features last; the local problem of such a site against the oracle on every route the sweep takes."""
    from tensornetworksfork_amd import ops
    from tensornetworksfork_amd.tensor import engine
    dev = _dev()
    L, X, core, R = _rand((S, C, rl), 1), _rand((S, p), 2), _rand((rl, p, rr), 3), _rand((S, rr), 4)
    g = _rand((S, C), 5)
    Hh = _rand((S, C, C), 6)
    H = (Hh @ Hh.transpose(1, 2) + 0.1 * torch.eye(C, dtype=torch.float64)).contiguous()
    Ld, Xd, cd, Rd = L.to(dev), X.to(dev), core.to(dev), R.to(dev)
    assert G.rel_err(ops.env_step_left(Ld, Xd, cd), O.env_left_step(L, core, X)) < TOL
    assert G.rel_err(ops.env_step_right(Rd, Xd, cd), O.env_right_step(R, core, X)) < TOL
    base_l, base_r = _rand((S, C, rr), 7), _rand((S, rl), 8)
    assert G.rel_err(ops.env_step_left(Ld, Xd, cd, out=base_l.to(dev)), base_l + O.env_left_step(L, core, X)) < TOL
    assert G.rel_err(ops.env_step_right(Rd, Xd, cd, out=base_r.to(dev)), base_r + O.env_right_step(R, core, X)) < TOL
    lp = engine.LocalProblem(S, C, (rl, p, rr), cd, lambda t: t, terms=[(Ld, Xd, Rd)])
    assert G.rel_err(lp.rhs(g.to(dev)), O.local_rhs(L, X, R, g)) < TOL
    u = torch.einsum('scr,sr->sc', O.env_left_step(L, core, X), R)
    assert G.rel_err(lp.jv(cd), u) < TOL
    assert G.rel_err(lp.matvec_fn(H.to(dev))(cd), O.local_matvec(L, X, R, H, core)) < TOL
    assert G.rel_err(ops.jvp(Ld, Xd, Rd, H.to(dev), cd), torch.einsum('scd,sd->sc', H, u)) < TOL


@gpu
@pytest.mark.parametrize("kind,C", [("square", 1), ("square", 3), ("mse", 1), ("mse", 4), ("xe", 1), ("xe", 2), ("xe", 9)])
def test_losses(kind, C):
    from tensornetworksfork_amd import ops
    dev = _dev()
    S = 333
    yp = _rand((S, C), 7) * 3
    if kind == "xe":
        lab = torch.randint(0, C + 1, (S,), generator=torch.Generator().manual_seed(8))
        y = torch.nn.functional.one_hot(lab, C + 1).to(torch.float64)
    else:
        y = _rand((S, C), 9)
    lo, g, H = ops.loss(kind, yp.to(dev), y.to(dev), w=1.3 if kind == "xe" else 1.0)
    kw = {"w": 1.3} if kind == "xe" else {}
    lo_r, g_r, H_r = O.LOSSES[kind](yp, y, **kw)
    assert G.rel_err(lo, lo_r) < 1e-12
    assert G.rel_err(g, g_r) < 1e-12
    assert G.rel_err(H, H_r) < 1e-12
    for batch in (50, 333, 1000):
        m = ops.batch_mean(lo, batch).item()
        ref = sum(lo_r[b0:b0 + batch].mean().item() for b0 in range(0, S, batch)) / ((S + batch - 1) // batch)
        assert abs(m - ref) <= 1e-12 * max(1.0, abs(ref))


DENSE_GRAM_FLOP_BUDGET = 1.2e12   # CPU GEMM work of the dense oracle Gram that a case may cost


@gpu
@pytest.mark.parametrize("C", [1, 4])
def test_xe_soft_labels(C):
    """Targets that are not one-hot (label smoothing): XEAutogradBregman's gradient is taken against
    the arg-max label, as differentiating cross_entropy(z, argmax y) gives (tensor/bregman.py:198-204,
    re-evaluated here with autograd on the CPU); KLDivBregman's against the probabilities themselves
    (bregman.py:140).  Loss and Hessian agree between the two."""
    from tensornetworksfork_amd.tensor import bregman
    dev = _dev()
    S, w = 257, 0.7
    x = _rand((S, C), 3) * 2
    y = torch.softmax(_rand((S, C + 1), 4) * 3, dim=-1)
    xr = x.clone().requires_grad_(True)
    z = torch.cat((w * xr, torch.zeros_like(xr[..., :1])), dim=-1)
    loss_r = torch.nn.functional.cross_entropy(z, y.argmax(dim=-1), reduction='none')
    g_r = torch.autograd.grad(loss_r.sum(), xr, create_graph=True)[0]
    H_r = torch.stack([torch.autograd.grad(g_r[:, i].sum(), xr, retain_graph=True)[0] for i in range(C)], dim=-2)
    lo, g, H = bregman.XEAutogradBregman(w=w).forward(x.to(dev), y.to(dev))
    assert G.rel_err(lo, loss_r.detach()) < 1e-12 and G.rel_err(g, g_r.detach()) < 1e-12 and G.rel_err(H, H_r) < 1e-12
    lo2, g2, H2 = bregman.KLDivBregman(w=w).forward(x.to(dev), y.to(dev))
    lo_o, g_o, H_o = O.loss_xe(x, y, w)
    assert G.rel_err(lo2, lo_o) < 1e-12 and G.rel_err(g2, g_o) < 1e-12 and G.rel_err(H2, H_o) < 1e-12
    lo_h, g_h, H_h = O.loss_xe(x, y, w, hard=True)
    assert G.rel_err(g, g_h) < 1e-12 and G.rel_err(g2, g_h) > 1e-3


@gpu
def test_cg_zero_right_hand_side():
    """b == 0: SciPy's cg returns zeros (with or without a start vector); no NaN from a 0/0 step."""
    from tensornetworksfork_amd.tensor.solvers import cg_solve
    dev = _dev()
    M = _rand((50, 50), 1)
    A = (M @ M.T + torch.eye(50, dtype=torch.float64)).to(dev)
    b = torch.zeros(50, dtype=torch.float64, device=dev)
    for x0 in (None, _rand((50,), 2).to(dev)):
        x, info = cg_solve(lambda v: A @ v, b, x0, 20, 1e-6)
        assert info == 0 and bool((x == 0).all())


@gpu
@pytest.mark.parametrize("S,C,rl,p,rr", SHAPES)
def test_gram_rhs_matvec(S, C, rl, p, rr):
    """Right-hand side and matrix-free product of every shape against the oracle.  The Gram matrix is
    compared entry by entry with the oracle's dense one where that costs the CPU less than
    DENSE_GRAM_FLOP_BUDGET (which includes the headline order D = 6144), and through its action on a
    vector against the oracle's matrix-free product where it does not; shapes whose packed sample row
    does not fit the Gram kernel (many features) have no Gram to test."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    D = rl * p * rr
    L, X, R = _rand((S, C, rl), 11), _rand((S, p), 12), _rand((S, rr), 13)
    g = _rand((S, C), 14)
    Hh = _rand((S, C, C), 15)
    Hsym = Hh @ Hh.transpose(1, 2) + 0.1 * torch.eye(C, dtype=torch.float64)
    eye = None
    for sym, H in ((True, Hsym), (False, Hh)):
        if C == 1 and not sym and D > 4000:
            continue   # a 1x1 Hessian is symmetric either way; the unsymmetric layout is covered by the small shapes
        Ld, Xd, Rd, Hd = L.to(dev), X.to(dev), R.to(dev), H.contiguous().to(dev)
        b_r = O.local_rhs(L, X, R, g)
        b = ops.rhs(Ld, Xd, Rd, g.to(dev))
        assert G.rel_err(b, b_r) < TOL
        v = _rand((rl, p, rr), 16)
        Av_r = O.local_matvec(L, X, R, H, v)
        Av = ops.matvec(Ld, Xd, Rd, Hd, v.to(dev))
        assert G.rel_err(Av, Av_r) < TOL, ("matvec", sym)
        if not ops.gram_supported(S, C, rl, p, rr, sym):
            assert p > 10
            continue
        ws = ops.gram(Ld, Hd, Xd, Rd, sym=sym)
        A = ops.gram_expand(ws)
        scale = ops.gram_scale(ws)
        if 2.0 * S * C * D * D <= DENSE_GRAM_FLOP_BUDGET:
            A_r = O.local_gram_rhs(L, X, R, g, H)[0].reshape(D, D)
            assert G.rel_err(A, A_r) < TOL, ("gram", sym)
            sc_r = A_r.diag().abs().mean()
            if eye is None:
                eye = torch.eye(D, dtype=torch.float64)
            A2 = ops.gram_expand(ws, scale=scale, ridge=0.25)
            assert G.rel_err(A2, A_r / sc_r + 0.25 * eye) < TOL
            del A_r, A2
        else:
            assert G.rel_err(A @ v.to(dev).reshape(-1), Av_r.reshape(-1)) < TOL, ("gram through its action", sym)
            sc_r = None
        if sc_r is not None:
            assert abs(scale.item() - sc_r.item()) <= 1e-12 * sc_r.item()
            w = _rand((rl, p, rr), 17)
            sr = ops.solve_rhs(b, w.to(dev), scale, 0.5)
            assert G.rel_err(sr, -(b_r / sc_r + 0.5 * w).reshape(-1)) < TOL
        del A, ws


@gpu
def test_krylov_vector_ops():
    from tensornetworksfork_amd import ops
    dev = _dev()
    x, y = _rand((6144,), 21), _rand((6144,), 22)
    xd, yd = x.to(dev), y.to(dev)
    d = ops.dot(xd, yd)
    assert abs(d.item() - (x * y).sum().item()) < 1e-10
    ops.axpy(d, xd, yd, sign=-1.0)
    assert G.rel_err(yd, y - (x * y).sum() * x) < 1e-12


@gpu
def test_gram_large_slabbed():
    """Multi-slab path (S large enough for 8 sample slabs) against linearity: the Gram of
    the whole set equals the sum of Grams of two halves, and a dense check on a sub-block."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    S, C, rl, p, rr = 20000, 1, 8, 6, 8
    L, X, R = _rand((S, C, rl), 31), _rand((S, p), 32), _rand((S, rr), 33)
    H = torch.full((S, 1, 1), 2.0, dtype=torch.float64)
    A_r, _ = O.local_gram_rhs(L, X, R, torch.zeros(S, 1, dtype=torch.float64), H)
    Ld, Xd, Rd, Hd = L.to(dev), X.to(dev), R.to(dev), H.to(dev)
    A = ops.gram_expand(ops.gram(Ld, Hd, Xd, Rd, sym=True))
    D = rl * p * rr
    assert G.rel_err(A, A_r.reshape(D, D)) < TOL
    h = S // 2
    A1 = ops.gram_expand(ops.gram(Ld[:h].contiguous(), Hd[:h].contiguous(), Xd[:h].contiguous(), Rd[:h].contiguous(),
                                  sym=True))
    A2 = ops.gram_expand(ops.gram(Ld[h:].contiguous(), Hd[h:].contiguous(), Xd[h:].contiguous(), Rd[h:].contiguous(),
                                  sym=True))
    assert G.rel_err(A1 + A2, A) < 1e-12


@gpu
def test_gram_called_off():
    """A Gram whose generation stands in the cancel word does no work (every workgroup leaves: the workspace keeps
    what the slab sums of its old contents give); the next one, a generation later, is complete again, bit for bit."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    S, C, rl, p, rr = 20000, 1, 8, 6, 8
    L, X, R = _rand((S, C, rl), 31).to(dev), _rand((S, p), 32).to(dev), _rand((S, rr), 33).to(dev)
    H = torch.full((S, 1, 1), 2.0, dtype=torch.float64, device=dev)
    ws = ops.gram(L, H, X, R, sym=True)
    want = ws.compressed.clone()
    assert want.abs().max().item() > 0
    ws.buf.zero_()
    ws.cancel_word.fill_(ws.generation % 0xFFFFFFFF + 2)      # the generation the next call will carry
    ws2 = ops.gram(L, H, X, R, sym=True, ws=ws)
    assert ws2 is ws and ws.compressed.abs().max().item() == 0.0
    ops.gram_cancel(ws)                                        # what a sweep does: same word, same generation
    assert int(ws.cancel_word.item()) == ws.generation % 0xFFFFFFFF + 1
    ops.gram(L, H, X, R, sym=True, ws=ws)
    assert torch.equal(ws.compressed, want)


@gpu
@pytest.mark.parametrize("S,C,rl,p,rr", [(96, 1, 4, 3, 4), (400, 1, 8, 3, 8), (700, 2, 7, 5, 9), (900, 1, 16, 6, 16),
                                         (64, 1, 4, 3, 1), (50, 1, 1, 5, 4)])
def test_cholesky_solve(S, C, rl, p, rr):
    """Blocked MFMA Cholesky (with the RHS as an extra block row) against a CPU LAPACK solve of
    the same scaled, ridge-shifted system; orders below, at and above multiples of the 64 block."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    L, X, R = _rand((S, C, rl), 41), _rand((S, p), 42), _rand((S, rr), 43)
    g = _rand((S, C), 44)
    Hh = _rand((S, C, C), 45)
    H = (Hh @ Hh.transpose(1, 2) + 0.1 * torch.eye(C, dtype=torch.float64)).contiguous()
    w = _rand((rl, p, rr), 46)
    A_r, b_r = O.local_gram_rhs(L, X, R, g, H)
    D = rl * p * rr
    Ld, Xd, Rd, Hd = L.to(dev), X.to(dev), R.to(dev), H.to(dev)
    ws = ops.gram(Ld, Hd, Xd, Rd, sym=True)
    b = ops.rhs(Ld, Xd, Rd, g.to(dev))
    scale = ops.gram_scale(ws)
    for ridge in (2.0, 0.02):
        x, info = ops.chol_solve_gram(ws, b, w.to(dev), scale, ridge)
        assert int(info.item()) == 0
        x_r = O.solve_system(A_r, b_r, w, 'ridge_cholesky', ridge / 2)
        assert G.rel_err(x, x_r.reshape(-1)) < 1e-8, (D, ridge)
    x, info = ops.chol_solve_gram(ws, b, w.to(dev), scale, -10.0)
    assert int(info.item()) == 1


@gpu
@pytest.mark.parametrize("S,C,rl,p,rr,mode", [(600, 1, 16, 6, 16, 1), (900, 1, 16, 6, 16, -1), (700, 2, 7, 5, 9, 1),
                                              (800, 1, 20, 3, 20, 1), (4000, 1, 32, 3, 32, -1)])
def test_cholesky_from_gram_equals_two_calls(S, C, rl, p, rr, mode):
    """ttk_chol_solve_gram_f64 (the single-launch factorisation takes its tiles straight from the compressed Gram)
    against ttk_chol_prepare_f64 + ttk_chol_solve_f64 (matrix expanded into the storage first): the same bits,
    for orders that are and are not multiples of the 64 block, one and several outputs, and on the multi-launch
    route (order below the threshold with the default mode), where the one call expands internally."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    L, X, R = _rand((S, C, rl), 51).to(dev), _rand((S, p), 52).to(dev), _rand((S, rr), 53).to(dev)
    Hh = _rand((S, C, C), 55)
    H = (Hh @ Hh.transpose(1, 2) + 0.1 * torch.eye(C, dtype=torch.float64)).contiguous().to(dev)
    w = _rand((rl, p, rr), 56).to(dev)
    ws = ops.gram(L, H, X, R, sym=True)
    b = ops.rhs(L, X, R, _rand((S, C), 54).to(dev))
    scale = ops.gram_scale(ws)
    prev = ops.chol_mode(mode)
    try:
        for ridge in (2.0, 0.02):
            x1, i1 = ops.chol_solve_gram(ws, b, w, scale, ridge)
            x2, i2 = ops.chol_solve_gram_two_calls(ws, b, w, scale, ridge)
            assert int(i1.item()) == 0 and int(i2.item()) == 0
            assert torch.equal(x1, x2), (rl * p * rr, ridge)
        x1, i1 = ops.chol_solve_gram(ws, b, None, None, 0.5)      # no scale, no ridge weights
        x2, i2 = ops.chol_solve_gram_two_calls(ws, b, None, None, 0.5)
        assert int(i1.item()) == int(i2.item()) == 0 and torch.equal(x1, x2)
    finally:
        ops.chol_mode(prev)


@gpu
@pytest.mark.parametrize("S,C,rl,p,rr", [(300, 1, 4, 16, 4), (1000, 3, 24, 50, 24), (777, 9, 48, 65, 48), (129, 2, 1, 17, 5),
                                         (500, 1, 7, 130, 9), (260, 2, 3, 256, 2), (64, 1, 5, 33, 1)])
def test_rhs_feature_matches_oracle(S, C, rl, p, rr):
    """The features-first right-hand side of sites with many features against the oracle's einsum, plain and
    accumulating; extents below / at / above the 16-wide tiles, ragged sample counts (last chunk, last slab)."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    L, X, R, g = _rand((S, C, rl), 81), _rand((S, p), 82), _rand((S, rr), 83), _rand((S, C), 84)
    ref = torch.einsum('sc,scl,sp,sr->lpr', g, L, X, R)
    b = ops.rhs_feature(L.to(dev), X.to(dev), R.to(dev), g.to(dev))
    assert G.rel_err(b, ref) < TOL
    b2 = ops.rhs_feature(L.to(dev), X.to(dev), R.to(dev), g.to(dev), out=b.clone())
    assert G.rel_err(b2, 2 * ref) < TOL
    assert torch.equal(ops.rhs_feature(L.to(dev), X.to(dev), R.to(dev), g.to(dev)), b)   # fixed summation order


def _small_ints(shape, seed, hi=4):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(-hi, hi + 1, shape, generator=g).to(torch.float64)


@gpu
@pytest.mark.parametrize("S,C,rl,p,rr", [(300, 1, 4, 16, 4), (1000, 3, 24, 50, 24), (777, 9, 48, 65, 48), (129, 2, 1, 17, 5),
                                         (500, 1, 7, 130, 9), (260, 2, 3, 256, 2), (64, 1, 5, 33, 1), (200, 2, 6, 40, 64),
                                         (333, 9, 100, 65, 100), (150, 2, 3, 20, 128), (90, 1, 2, 96, 70)])
def test_float32_matrix_arithmetic_of_many_feature_sites(S, C, rl, p, rr):
    """The float32 instantiations (v_mfma_f32_16x16x4_f32) of the row GEMM, the dense right-environment step and
    the features-first right-hand side: (1) on small integers every float32 product and sum is exact, so the
    result must EQUAL the oracle's einsum -- this pins the operand / result lane maps and every ragged edge;
    (2) on random data the error against the fp64 einsum stays at float32 rounding (1e-5 of the largest entry:
    sums of up to 65 x 64 products per entry, 1 000 samples per slab)."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    for exact in (True, False):
        mk = (lambda sh, sd: _small_ints(sh, sd)) if exact else (lambda sh, sd: _rand(sh, sd))
        L, X, R, g = mk((S, C, rl), 91), mk((S, p), 92), mk((S, rr), 93), mk((S, C), 94)
        core = mk((rl, p, rr), 95)
        Ld, Xd, Rd, gd, cd = (t.to(dev) for t in (L, X, R, g, core))

        def check(got, ref):
            if exact:
                assert torch.equal(got.cpu(), ref)
            else:
                assert (got.cpu() - ref).abs().max() <= 1e-5 * ref.abs().max()

        ref_b = torch.einsum('sc,scl,sp,sr->lpr', g, L, X, R)
        check(ops.rhs_feature(Ld, Xd, Rd, gd, mm32=True), ref_b)
        check(ops.rhs_feature(Ld, Xd, Rd, gd, out=ops.rhs_feature(Ld, Xd, Rd, gd, mm32=True), mm32=True), 2 * ref_b)
        ref_z = torch.einsum('sr,sp,lpr->sl', R, X, core)
        z = ops.env_step_right(Rd, Xd, cd, mm32=True)
        check(z, ref_z)
        check(ops.env_step_right(Rd, Xd, cd, out=z.clone(), mm32=True), 2 * ref_z)
        ref_t = torch.einsum('sk,lkr->slr', X, core).reshape(S, rl * rr)
        check(ops.gemm_rows(Xd, cd, rl * rr, bmode=1, rl=rl, rr=rr, mm32=True), ref_t)
        Bm = mk((p, 37), 96)
        check(ops.gemm_rows(Xd, Bm.to(dev), 37, bmode=0, mm32=True), X @ Bm)
        check(ops.gemm_rows(Xd, Bm.T.contiguous().to(dev), 37, bmode=2, mm32=True), X @ Bm)
        ref_l = torch.einsum('scl,sp,lpr->scr', L, X, core)
        check(ops.env_step_left_dense(Ld, Xd, cd, mm32=True), ref_l)


@gpu
@pytest.mark.parametrize("n", [40, 64, 130, 512, 1000, 2304])
def test_cholesky_single_launch_matches_multi_launch(n):
    """The two forms of the factorisation (one launch per panel / trailing update; one launch of tile tasks
    that wait on each other through progress words) against LAPACK and against each other: solution, pivot
    report for a matrix that is not positive definite, and the same bits on every run (a stale read between
    workgroups would show here first).  Orders below, at and above multiples of the 64 block."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(70 + n)
    M = torch.randn(n, n + 32, generator=g, dtype=torch.float64)
    A = (M @ M.T / (n + 32) + 0.05 * torch.eye(n, dtype=torch.float64))
    b = torch.randn(n, generator=g, dtype=torch.float64)
    scale = A.diagonal().abs().mean()
    ref = torch.linalg.solve(A / scale, -b / scale)
    Ad, bd = A.to(dev), b.to(dev)
    bad = A.clone()
    k = (2 * n) // 3
    bad[k, k] = -1.0     # leading minor of order k+1 is not positive definite
    badd = bad.to(dev)
    out = {}
    old = ops.chol_mode(-1)
    try:
        for mode in (0, 1):
            ops.chol_mode(mode)
            x, info = ops.chol_solve_dense(Ad, bd)
            assert int(info.item()) == 0
            assert G.rel_err(x, ref) < 1e-10, (n, mode)
            for _ in range(4):
                assert torch.equal(ops.chol_solve_dense(Ad, bd)[0], x), (n, mode)
            _, info_bad = ops.chol_solve_dense(badd, bd)
            out[mode] = (x, int(info_bad.item()))
    finally:
        ops.chol_mode(old)
    assert out[0][1] == out[1][1] == k + 1
    assert G.rel_err(out[1][0], out[0][0]) < 1e-11


@gpu
@pytest.mark.parametrize("n", [40, 200, 257, 1000, 2304, 6144])
def test_backward_substitution_one_launch_equals_one_launch_per_group(n):
    """Behind the single-launch factorisation the backward substitution runs as one launch of workgroups that wait
    for each other's block rows; against one launch per group of block rows: the same bits (same operations in the
    same order), for group counts that do and do not divide the block rows, repeatably."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(90 + n)
    M = torch.randn(n, n + 32, generator=g, dtype=torch.float64, device=dev)
    A = M @ M.T / (n + 32) + 0.05 * torch.eye(n, dtype=torch.float64, device=dev)
    b = torch.randn(n, generator=g, dtype=torch.float64, device=dev)
    old = ops.chol_mode(1)
    try:
        ops.chol_mode(200)
        x0, i0 = ops.chol_solve_dense(A, b)
        ops.chol_mode(201)
        x1, i1 = ops.chol_solve_dense(A, b)
        assert int(i0.item()) == int(i1.item()) == 0
        assert torch.equal(x0, x1), n
        for _ in range(3):
            assert torch.equal(ops.chol_solve_dense(A, b)[0], x1)
        scale = A.diagonal().abs().mean()
        assert (((A / scale) @ x1 + b / scale).norm() / (b / scale).norm()).item() < 1e-10
    finally:
        ops.chol_mode(201)
        ops.chol_mode(old)


@gpu
def test_cholesky_single_launch_gives_up_cleanly():
    """Every wait inside the single-launch factorisation is bounded: with the patience of a waiting lane cut to
    a handful of polls the launch aborts itself (info = -1) instead of hanging, and the next solve is fine."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    n = 2304
    g = torch.Generator().manual_seed(5)
    M = torch.randn(n, n + 32, generator=g, dtype=torch.float64)
    A = (M @ M.T / (n + 32) + 0.05 * torch.eye(n, dtype=torch.float64)).to(dev)
    b = torch.randn(n, generator=g, dtype=torch.float64).to(dev)
    old = ops.chol_mode(1)
    try:
        ops.chol_mode(1000 + 1)
        _, info = ops.chol_solve_dense(A, b)
        assert int(info.item()) == -1
        ops.chol_mode(1000)
        x, info = ops.chol_solve_dense(A, b)
        assert int(info.item()) == 0
        scale = A.diagonal().abs().mean()
        assert (((A / scale) @ x + b / scale).norm() / (b / scale).norm()).item() < 1e-10
    finally:
        ops.chol_mode(1000)
        ops.chol_mode(old)


@gpu
def test_sweep_falls_back_when_single_launch_gives_up():
    """A sweep whose single-launch factorisation aborts (here: patience cut to one poll; in the field: a compute-unit
    mask that leaves its waiting workgroups without partners) undoes the update, switches this process to one launch
    per panel and does the node again: same cores as a sweep that factorised that way from the start."""
    from tensornetworksfork_amd import ops
    from tensornetworksfork_amd.tensor.bregman import SquareBregFunction
    from tensornetworksfork_amd.tensor.layers import TensorTrainLayer
    dev = _dev()
    g = torch.Generator().manual_seed(12)
    x = [(torch.rand(700, 3, generator=g, dtype=torch.float64) * 2 - 1).to(dev) for _ in range(4)]
    y = torch.randn(700, 1, generator=g, dtype=torch.float64).to(dev)

    def sweep():
        layer = TensorTrainLayer(4, 32, 3, output_shape=1, constrict_bond=False, seed=2).to(dev)
        tn = layer.tensor_network
        assert tn.accumulating_swipe(x, y, SquareBregFunction(), batch_size=128, num_swipes=1, method="ridge_cholesky",
                                     eps=0.5, orthonormalize=True)
        return [n.tensor.clone() for n in tn.main_nodes]

    old = ops.chol_mode(0)
    try:
        ref = sweep()                      # one launch per panel throughout
        ops.chol_mode(1)
        ops.chol_mode(1000 + 1)            # the first middle site (order 3072) aborts its single launch
        got = sweep()
        assert ops.chol_mode(-2) == 0      # (a query) the process has switched
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
    finally:
        ops.chol_mode(1000)
        ops.chol_mode(old)


@gpu
@pytest.mark.parametrize("m,n", [(192, 32), (6, 6), (36, 32), (17, 32), (12, 4), (5, 1), (1, 3), (300, 48), (64, 64)])
def test_qr_matches_lapack(m, n):
    """One-workgroup Householder QR against torch.linalg.qr on the CPU (same sign convention)."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    A = _rand((m, n), 51)
    Q, R = ops.qr(A.to(dev))
    Qr, Rr = torch.linalg.qr(A, mode='reduced')
    assert tuple(Q.shape) == tuple(Qr.shape) and tuple(R.shape) == tuple(Rr.shape)
    assert G.rel_err(Q, Qr) < 1e-10
    assert G.rel_err(R, Rr) < 1e-10
    assert G.rel_err(Q.cpu() @ R.cpu(), A) < 1e-12


@gpu
@pytest.mark.parametrize("n,m", [(32, 192), (16, 96), (6, 24), (10, 500), (8, 5), (1, 7), (24, 300), (64, 64)])
def test_qr_right_matches_flipped_lapack(n, m):
    """The gauge move to the left in one launch (core read in place, factors written as they are used) against the
    reference's construction on the CPU: QR of the doubly flipped transposed matrix, factors flipped back."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    T = _rand((n, m), 52)
    assert ops.qr_right_supported(T.to(dev))
    Qt, Rm = ops.qr_right(T.to(dev))
    Qr, Rr = torch.linalg.qr(torch.flip(T.T, dims=[0, 1]), mode='reduced')
    Rw = torch.flip(Rr.T, dims=[0, 1])
    Qw = torch.flip(Qr, dims=[0, 1])
    assert tuple(Qt.shape) == tuple(Qw.T.shape) and tuple(Rm.shape) == tuple(Rw.shape)
    assert G.rel_err(Qt, Qw.T) < 1e-10
    assert G.rel_err(Rm, Rw) < 1e-10
    assert G.rel_err(Rm.cpu() @ Qt.cpu(), T) < 1e-12


@gpu
def test_gram_feature_pair_classes():
    """Polynomial features: the classed Gram (one slice per value of p+q) equals the full one."""
    from tensornetworksfork_amd import ops
    from tensornetworksfork_amd.tensor.engine import pair_classes
    dev = _dev()
    S, C, rl, p, rr = 3000, 2, 5, 5, 6
    x = torch.rand(S, dtype=torch.float64, device=dev) * 2 - 1
    X = torch.stack([x ** d for d in range(p)], dim=-1).contiguous()
    rep = pair_classes(X)
    assert rep is not None and len(set(rep)) == 2 * p - 1
    L, R = _rand((S, C, rl), 1).to(dev), _rand((S, rr), 2).to(dev)
    Hs = _rand((S, C, C), 3).to(dev)
    H = (Hs @ Hs.transpose(1, 2)).contiguous()
    full = ops.gram_expand(ops.gram(L, H, X, R, sym=True))
    classed = ops.gram_expand(ops.gram(L, H, X, R, sym=True, pair_rep=rep))
    assert ((classed - full).abs().max() / full.abs().max()).item() < 1e-13
    big = ops.gram_expand(ops.gram(L[:40].contiguous(), H[:40].contiguous(), X[:40].contiguous(), R[:40].contiguous(),
                                   sym=False, pair_rep=rep))       # one slab: in-place replication
    ref = ops.gram_expand(ops.gram(L[:40].contiguous(), H[:40].contiguous(), X[:40].contiguous(), R[:40].contiguous(),
                                   sym=False))
    assert ((big - ref).abs().max() / ref.abs().max()).item() < 1e-13
    with pytest.raises(Exception):
        ops.gram(L, H, X, R, sym=True, pair_rep=[1] + list(range(1, p * (p + 1) // 2)))   # rep must not point forward


@gpu
def test_two_stage_products_match_kernels():
    """Sites with many features contract the features first (library GEMM): same J v, J^T g and
    mat-vec as the Khatri-Rao kernels."""
    from tensornetworksfork_amd import ops
    from tensornetworksfork_amd.tensor import engine
    dev = _dev()
    S, C, rl, p, rr = 1500, 3, 5, 20, 6
    L, X, R = _rand((S, C, rl), 1).to(dev), _rand((S, p), 2).to(dev), _rand((S, rr), 3).to(dev)
    Hs = _rand((S, C, C), 4).to(dev)
    H = (Hs @ Hs.transpose(1, 2)).contiguous()
    v, g = _rand((rl, p, rr), 5).to(dev), _rand((S, C), 6).to(dev)
    lp = engine.LocalProblem(S, C, (rl, p, rr), v, lambda t: t, terms=[(L, X, R)])
    assert p >= engine.TWO_STAGE_MIN_P
    u_ref = ops.predict(ops.env_step_left(L, X, v), R)
    assert ((lp.jv(v) - u_ref).norm() / u_ref.norm()).item() < 1e-13
    b_ref = ops.rhs(L, X, R, g)
    assert ((lp.rhs(g) - b_ref).norm() / b_ref.norm()).item() < 1e-13
    mv_ref = ops.matvec(L, X, R, H, v)
    assert ((lp.matvec_fn(H)(v) - mv_ref).norm() / mv_ref.norm()).item() < 1e-13


@gpu
@pytest.mark.parametrize("D", [7, 6144, 100003])
def test_krylov_vector_updates(D):
    """Fused CG / Lanczos vector kernels against the same recurrences in torch on the CPU."""
    from tensornetworksfork_amd import ops
    from tests import _kernel_standins as K
    dev = _dev()
    x, r, p, q = (_rand((D,), k) for k in (1, 2, 3, 4))
    q = q + 3.0 * p                       # <p,q> well away from zero
    for atol in (1e-9, 1e9):              # running / already converged (then nothing may change)
        st = torch.tensor([float((r * r).sum()), atol, 0.0, 0.0], dtype=torch.float64)
        ref = [t.clone() for t in (x, r, p)]
        st_ref = st.clone()
        K.cg_update(ref[0], ref[1], ref[2], q, st_ref)
        got = [t.clone().to(dev) for t in (x, r, p)]
        st_dev = st.to(dev)
        ops.cg_update(got[0], got[1], got[2], q.to(dev), st_dev)
        for g_, r_ in zip(got, ref):
            assert G.rel_err(g_, r_) < 1e-12
        assert G.rel_err(st_dev[:1], st_ref[:1]) < 1e-12 and torch.equal(st_dev[1:].cpu(), st_ref[1:])
    for j, vp in ((1, None), (3, _rand((D,), 6))):
        w, vc = _rand((D,), 7), _rand((D,), 8)
        al, be = torch.zeros(5, dtype=torch.float64), torch.rand(6, dtype=torch.float64)
        w_ref, al_ref, be_ref = w.clone(), al.clone(), be.clone()
        vn_ref = K.lanczos_update(w_ref, vp, vc, torch.empty(D, dtype=torch.float64), al_ref, be_ref, j, 1e-30,
                                  torch.zeros(1, dtype=torch.float64))
        w_dev, al_dev, be_dev = w.to(dev), al.to(dev), be.to(dev)
        vn = ops.lanczos_update(w_dev, None if vp is None else vp.to(dev), vc.to(dev),
                                torch.empty(D, dtype=torch.float64, device=dev), al_dev, be_dev, j, 1e-30,
                                torch.zeros(1, dtype=torch.float64, device=dev))
        assert G.rel_err(vn, vn_ref) < 1e-12 and G.rel_err(w_dev, w_ref) < 1e-12
        assert G.rel_err(al_dev, al_ref) < 1e-12 and G.rel_err(be_dev, be_ref) < 1e-12


@gpu
@pytest.mark.parametrize("n,maxiter,rtol", [(60, 200, 1e-10), (300, 25, 1e-6), (2000, 40, 1e-3), (9000, 12, 1e-8)])
def test_krylov_solvers_match_scipy(n, maxiter, rtol):
    """Device-resident CG and MINRES (one fused vector kernel per iteration, scalars on the device)
    against scipy.sparse.linalg.cg / minres in float64 on the same operator: same iterates, same
    stopping iteration (hence the same x) and the same info, with and without a start vector.  1e-7:
    summation orders differ, and tens of iterations without re-orthogonalisation amplify that
    (3e-9 observed after 60 MINRES iterations)."""
    import numpy as np
    import scipy.sparse.linalg as spla
    from tensornetworksfork_amd.tensor.solvers import cg_solve, minres_solve
    dev = _dev()
    M = _rand((n, 24), 1)
    A = M @ M.T + torch.diag(torch.linspace(0.5, 3.0, n, dtype=torch.float64))
    b = _rand((n,), 2)
    Ad = A.to(dev)
    for x0 in (None, _rand((n,), 3)):
        for name, solve, ref in (("cg", cg_solve, spla.cg), ("minres", minres_solve, spla.minres)):
            x_r, info_r = ref(A.numpy(), b.numpy(), x0=None if x0 is None else x0.numpy(), maxiter=maxiter, rtol=rtol)
            x, info = solve(lambda v: Ad @ v, b.to(dev), None if x0 is None else x0.to(dev), maxiter, rtol)
            assert (info == 0) == (info_r == 0), (name, info, info_r)
            assert G.rel_err(x, torch.tensor(np.asarray(x_r))) < 1e-7, (name, n, x0 is None)


@gpu
@pytest.mark.parametrize("S,rl,p", [(54000, 32, 6), (2000, 16, 4), (40, 32, 8)])
def test_cg_fused_matches_unfused(S, rl, p):
    """The two-launch conjugate-gradient iteration (product with the direction update in its prologue,
    slab reduction with the x / r updates behind it) against the same recurrence run kernel by kernel
    and against SciPy on the explicit Gram matrix, cold and warm started, converging and not."""
    import numpy as np
    import scipy.sparse.linalg as spla
    from tensornetworksfork_amd import ops
    from tensornetworksfork_amd.tensor.solvers import cg_solve, cg_solve_fused
    dev = _dev()
    L, X, R = _rand((S, 1, rl), 1).to(dev), _rand((S, p), 2).to(dev), _rand((S, rl), 3).to(dev)
    H = (_rand((S, 1, 1), 4).abs() + 0.5).to(dev)
    b = _rand((rl, p, rl), 5).to(dev)
    assert ops.matvec_fused_supported(S, 1, rl, p, rl)
    fused = ops.CGFused(L, X, R, H)
    mv = lambda v: ops.rhs(L, X, R, ops.jvp(L, X, R, H, v.reshape(rl, p, rl))).reshape(v.shape)   # noqa: E731
    A = ops.gram_expand(ops.gram(L, H, X, R, sym=True)).cpu().numpy() if rl * p * rl <= 2048 else None
    for x0 in (None, _rand((rl, p, rl), 6).to(dev)):
        # (with fewer samples than unknowns the operator is singular and long runs blow up chaotically)
        for maxiter, rtol in ((7, 1e-30), (60, 1e-3)) if S >= rl * p * rl else ((5, 1e-30),):
            x_f, info_f = cg_solve_fused(fused, mv, b, x0, maxiter, rtol)
            x_u, info_u = cg_solve(mv, b, x0, maxiter, rtol)
            assert info_f == info_u
            assert G.rel_err(x_f, x_u) < 1e-8, (S, rl, p, maxiter)
            if A is not None:
                x_r, info_r = spla.cg(A, b.reshape(-1).cpu().numpy(), x0=None if x0 is None else x0.reshape(-1).cpu().numpy(),
                                      maxiter=maxiter, rtol=rtol)
                assert (info_f == 0) == (info_r == 0)
                assert G.rel_err(x_f.reshape(-1), torch.tensor(np.asarray(x_r))) < 1e-7


@gpu
@pytest.mark.parametrize("S,C,rl,p,rr", [(300, 9, 24, 50, 24), (1000, 3, 48, 65, 48), (77, 1, 5, 130, 7), (4099, 2, 9, 17, 64),
                                         (64, 4, 16, 16, 16), (130, 9, 1, 33, 20), (333, 9, 100, 65, 100), (150, 2, 3, 20, 128),
                                         (90, 1, 2, 96, 81), (70, 1, 2, 130, 100)])
def test_many_feature_kernels(S, C, rl, p, rr):
    """Sites with many features (conv columns): right / left environment steps with the features
    contracted first, the plain row GEMM in its three operand layouts, and the per-sample
    contraction with the left environment, against einsum on the CPU (ragged row tiles, K above
    one 64-chunk, bonds that are no multiple of the 16-wide MFMA tile)."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    E, X, core, R = _rand((S, C, rl), 1), _rand((S, p), 2), _rand((rl, p, rr), 3), _rand((S, rr), 4)
    Ed, Xd, cd, Rd = E.to(dev), X.to(dev), core.to(dev), R.to(dev)
    assert p >= ops.DENSE_MIN_P
    z = ops.env_step_right(Rd, Xd, cd)
    z_r = torch.einsum('sr,sp,lpr->sl', R, X, core)
    assert G.rel_err(z, z_r) < TOL
    base = _rand((S, rl), 5)
    assert G.rel_err(ops.env_step_right(Rd, Xd, cd, out=base.to(dev)), base + z_r) < TOL
    out_r = torch.einsum('scl,sp,lpr->scr', E, X, core)
    assert G.rel_err(ops.env_step_left_dense(Ed, Xd, cd), out_r) < TOL
    base = _rand((S, C, rr), 6)
    assert G.rel_err(ops.env_step_left_dense(Ed, Xd, cd, out=base.to(dev)), base + out_r) < TOL
    T_r = torch.einsum('sp,lpr->slr', X, core).reshape(S, rl * rr)
    assert G.rel_err(ops.gemm_rows(Xd, cd, rl * rr, bmode=1, rl=rl, rr=rr), T_r) < TOL
    Bm = _rand((p, 70), 7)
    assert G.rel_err(ops.gemm_rows(Xd, Bm.to(dev), 70), X @ Bm) < TOL
    assert G.rel_err(ops.gemm_rows(Xd, Bm.T.contiguous().to(dev), 70, bmode=2), X @ Bm) < TOL
    Tt = _rand((S, rl, 37), 8)
    assert G.rel_err(ops.lcontract(Ed, Tt.to(dev)), torch.einsum('scl,slj->scj', E, Tt)) < TOL


@gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("S,P,px,ca,cb,C", [(500, 50, 17, 1, 1, 9), (333, 65, 49, 1, 1, 9), (90, 10, 5, 2, 3, 4), (1, 3, 2, 1, 2, 1)])
def test_conv_feature_and_filter_rows(dtype, S, P, px, ca, cb, C):
    """The two kernels that read the raw image patches (float64, or float32 as the CIFAR script holds
    them): feature vector G = X . F and the filter node's Jacobian rows J = M . X."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    X = _rand((S, P, px), 1).to(dtype)
    Fc = _rand((ca, px, cb), 2)
    Gd = ops.conv_feature(X.to(dev), Fc.to(dev))
    assert G.rel_err(Gd, torch.einsum('spx,axb->absp', X.to(torch.float64), Fc)) < TOL
    Mv = _rand((S, C, P), 3)
    J = torch.full((S, C, ca, px, cb), 7.0, dtype=torch.float64, device=dev)
    ref = torch.full((S, C, ca, px, cb), 7.0, dtype=torch.float64)
    for a in range(ca):
        for b in range(cb):
            if (a + b) % 2 == 0 or ca * cb == 1:
                ops.patch_contract(Mv.to(dev), X.to(dev), J, a, b)
                ref[:, :, a, :, b] = torch.einsum('scp,spx->scx', Mv, X.to(torch.float64))
    assert G.rel_err(J, ref) < TOL


@gpu
@pytest.mark.parametrize("n,D1,D2", [(5000, 17, 17), (333, 130, 70), (64, 64, 64), (1, 3, 5), (2049, 200, 200)])
def test_gemm_tn_and_dense_cholesky(n, D1, D2):
    from tensornetworksfork_amd import ops
    dev = _dev()
    A, B = _rand((n, D1), 1), _rand((n, D2), 2)
    assert G.rel_err(ops.gemm_tn(A.to(dev), B.to(dev)), A.T @ B) < TOL
    base = _rand((D1, D2), 3)
    assert G.rel_err(ops.gemm_tn(A.to(dev), B.to(dev), out=base.to(dev)), base + A.T @ B) < TOL
    if D1 == D2:
        Gm = A.T @ A + 0.01 * torch.eye(D1, dtype=torch.float64)
        b, w = _rand((D1,), 4), _rand((D1,), 5)
        x, info = ops.chol_solve_dense(Gm.to(dev), b.to(dev), w.to(dev), 0.5)
        assert int(info.item()) == 0
        x_r = O.solve_system(Gm, b, w, 'ridge_cholesky', 0.25)
        assert G.rel_err(x, x_r) < 1e-9
        x, info = ops.chol_solve_dense(Gm.to(dev), b.to(dev), None, -5.0)
        assert int(info.item()) >= 1


@gpu
@pytest.mark.parametrize("S,C,D", [(1000, 9, 49), (54001, 9, 17), (777, 1, 5), (300, 3, 33), (129, 2, 128), (64, 10, 1),
                                   (500, 9, 272), (100, 2, 1000)])
def test_rows_dot(S, C, D):
    """J v for explicit rows (filter nodes of a conv tensor train) against a float64 CPU product."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    J, v = _rand((S, C, D), 71), _rand((D,), 72)
    u = ops.rows_dot(J.to(dev), v.to(dev))
    ref = torch.einsum('scd,d->sc', J, v)
    assert G.rel_err(u, ref) < 1e-13


@gpu
@pytest.mark.parametrize("D", [9000, 28800, 149760])
def test_lanczos_update_wide(D):
    """The Lanczos vector step spread over many workgroups (vectors beyond one workgroup's registers) against the
    recurrence written out on the CPU, first iteration (no v_prev) and a later one; repeatable bit for bit."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    w0, vp, vc = _rand((D,), 81), _rand((D,), 82), _rand((D,), 83)
    vc = vc / vc.norm()
    for j, prev in ((1, None), (3, vp)):
        betas = torch.zeros(8, dtype=torch.float64)
        betas[j - 1] = 0.37
        alphas = torch.zeros(8, dtype=torch.float64)
        outs = []
        for _ in range(2):
            w, al, be = w0.clone().to(dev), alphas.clone().to(dev), betas.clone().to(dev)
            stop = torch.zeros(1, dtype=torch.float64, device=dev)
            vn = ops.lanczos_update(w, None if prev is None else prev.to(dev), vc.to(dev),
                                    torch.empty(D, dtype=torch.float64, device=dev), al, be, j, 1e-30, stop)
            assert stop.item() == 0.0
            outs.append((w.cpu(), vn.cpu(), al.cpu(), be.cpu()))
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b)
        wr = w0.clone() if prev is None else w0 - 0.37 * prev
        a_ref = wr @ vc
        wr = wr - a_ref * vc
        b_ref = wr.norm()
        w, vn, al, be = outs[0]
        assert abs(al[j - 1] - a_ref) <= 1e-12 * max(1.0, abs(a_ref))
        assert abs(be[j] - b_ref) <= 1e-12 * b_ref
        assert G.rel_err(w, wr) < 1e-12 and G.rel_err(vn, wr / b_ref) < 1e-12


@gpu
@pytest.mark.parametrize("D", [17, 5000, 9000, 40000])
def test_lanczos_stop_and_finish(D):
    """The exit of the reference's Lanczos loop (beta < tol, tensor/network.py:813) taken inside the vector kernel:
    the iteration is recorded, later calls change nothing; and the end of the solver (tridiagonal solve by
    elimination with partial pivoting + x0 + V y over the rows of the basis) against torch.linalg.solve on the dense
    tridiagonal matrix and a matrix-vector product, with and without the exit, for tridiagonal matrices that force
    row interchanges (small diagonal entries) and for sizes on both sides of the in-LDS limit."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    w0, vc = _rand((D,), 85), _rand((D,), 86)
    vc = vc / vc.norm()
    al, be = torch.zeros(4, dtype=torch.float64, device=dev), torch.zeros(5, dtype=torch.float64, device=dev)
    stop = torch.zeros(1, dtype=torch.float64, device=dev)
    w = w0.clone().to(dev)
    vn = torch.zeros(D, dtype=torch.float64, device=dev)
    ops.lanczos_update(w, None, vc.to(dev), vn, al, be, 2, 1e30, stop)        # beta < tol: iteration 2 is the last
    assert stop.item() == 2.0
    al2, be2, w2 = al.clone(), be.clone(), w.clone()
    ops.lanczos_update(w, None, vc.to(dev), vn, al, be, 3, 1e30, stop)        # frozen
    assert stop.item() == 2.0 and torch.equal(al, al2) and torch.equal(be, be2) and torch.equal(w, w2)
    for m_cap, m_run, stop_at in ((1, 1, 0), (2, 2, 0), (7, 7, 0), (40, 33, 0), (40, 40, 12), (3000, 2500, 0),
                                  (4200, 4100, 0)):
        if m_cap * D > 3e7:
            continue
        g = torch.Generator().manual_seed(m_cap + D)
        alphas = torch.rand(m_cap, generator=g, dtype=torch.float64) * 2 - 1
        alphas[::3] *= 1e-3                                                    # small pivots: rows get interchanged
        betas = torch.rand(m_cap + 1, generator=g, dtype=torch.float64) + 0.1
        V = torch.rand(m_cap + 1, D, generator=g, dtype=torch.float64) - 0.5
        x0 = torch.rand(D, generator=g, dtype=torch.float64)
        m = stop_at if stop_at else m_run
        Tm = torch.diag(alphas[:m]) + torch.diag(betas[1:m], 1) + torch.diag(betas[1:m], -1)
        rhs = torch.zeros(m, dtype=torch.float64)
        rhs[0] = betas[0]
        yv = torch.linalg.solve(Tm, rhs)
        ref = x0 + V[:m].T @ yv
        V[m:] = float("nan")                                                   # rows beyond the basis are never read
        got = ops.lanczos_finish(V.to(dev), alphas.to(dev), betas.to(dev),
                                 torch.tensor([float(stop_at)], dtype=torch.float64, device=dev), m_run, x0.to(dev))
        cond = torch.linalg.cond(Tm).item()
        assert G.rel_err(got, ref) < 1e-13 * max(cond, 10.0), (m_cap, m_run, stop_at, cond)


@gpu
@pytest.mark.parametrize("S,C,D", [(1000, 9, 49), (20011, 9, 17), (777, 1, 5), (300, 3, 33), (129, 2, 128), (64, 10, 1),
                                   (513, 32, 128), (2000, 9, 272), (333, 8, 512), (257, 3, 129)])
def test_rows_matvec_and_lhl(S, C, D):
    """The one-pass operators on per-sample blocks of rows, against float64 CPU products: J^T H J v for a narrow
    node, and GL = L^T H (L z) per sample; non-symmetric Hessians pin the orientation of the Hessian product."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    J, v, z = _rand((S, C, D), 91), _rand((D,), 92), _rand((S, D), 93)
    H = _rand((S, C, C), 94)
    u = torch.einsum('scd,d->sc', J, v)
    hu = torch.einsum('scd,sc->sd', H, u)
    q_ref = torch.einsum('scd,sc->d', J, hu)
    q = ops.rows_matvec(J.to(dev), H.to(dev), v.to(dev))
    assert G.rel_err(q, q_ref) < 1e-12
    assert torch.equal(q, ops.rows_matvec(J.to(dev), H.to(dev), v.to(dev)))      # fixed summation order
    gw = _rand((S, C), 95)
    assert G.rel_err(ops.rows_rhs(J.to(dev), gw.to(dev)), torch.einsum('scd,sc->d', J, gw)) < 1e-12
    if D > 128:      # the per-sample form serves left environments, whose bond stays below that
        return
    u2 = torch.einsum('scl,sl->sc', J, z)
    hu2 = torch.einsum('scd,sc->sd', H, u2)
    gl_ref = torch.einsum('scl,sc->sl', J, hu2)
    assert G.rel_err(ops.lhl(J.to(dev), H.to(dev), z.to(dev)), gl_ref) < 1e-12


@gpu
@pytest.mark.parametrize("D", [9001, 149760])
def test_wide_krylov_updates_match_one_workgroup(D):
    """CG and MINRES vector updates of long vectors (many workgroups, partial sums in a workspace) against the
    one-workgroup kernels on the same data, three iterations in a row: same recurrences, sums in another order."""
    from tensornetworksfork_amd import _lib, ops
    dev = _dev()
    lib = _lib.lib()
    st_ = torch.cuda.current_stream().cuda_stream

    def cg_run(wide):
        x, r, p = (_rand((D,), k).to(dev) for k in (1, 2, 3))
        state = torch.tensor([float((r * r).sum()), 1e-30, 0.0, 0.0], dtype=torch.float64, device=dev)
        for it in range(3):
            q = (3.0 * p + 0.01 * torch.roll(p, 1)).contiguous()          # a fixed SPD-like operator
            if wide:
                ops.cg_update(x, r, p, q, state)
            else:
                assert lib.ttk_cg_update_f64(x.data_ptr(), r.data_ptr(), p.data_ptr(), q.data_ptr(), state.data_ptr(), D, st_) == 0
        return x, r, p, state

    for a, b in zip(cg_run(True), cg_run(False)):
        assert G.rel_err(a, b) < 1e-12

    def minres_run(wide):
        b = _rand((D,), 5).to(dev)
        beta1 = b.norm()
        y, r1, r2, v = b.clone(), b.clone(), b.clone(), (b / beta1).contiguous()
        w, w2, x = torch.zeros_like(b), torch.zeros_like(b), torch.zeros_like(b)
        st = torch.zeros(18, dtype=torch.float64, device=dev)
        st[3], st[6], st[7], st[11], st[12], st[14] = beta1, beta1, beta1, 1e308, -1.0, beta1
        st[15], st[16], st[17] = 1e-12, 50.0, 2.220446049250313e-16
        for it in range(3):
            y = (3.0 * v + 0.01 * torch.roll(v, 1) + 0.01 * torch.roll(v, -1)).contiguous()
            if wide:
                ops.minres_update(y, r1, r2, v, w, w2, x, st)
            else:
                assert lib.ttk_minres_update_f64(y.data_ptr(), r1.data_ptr(), r2.data_ptr(), v.data_ptr(), w.data_ptr(),
                                                 w2.data_ptr(), x.data_ptr(), st.data_ptr(), D, st_) == 0
        return x, v, w, w2, r1, r2, st

    outs_w, outs_1 = minres_run(True), minres_run(False)
    assert float(outs_1[-1][1].item()) == 3.0          # three iterations were counted
    for a, b in zip(outs_w, outs_1):
        assert G.rel_err(a, b) < 1e-11


@gpu
@pytest.mark.parametrize("shape,k,st,dt_in,dt_out", [((300, 1, 28, 28), 4, 4, torch.float32, torch.float32),
                                                     ((257, 3, 32, 32), 4, 4, torch.float32, torch.float32),
                                                     ((64, 1, 28, 28), 4, 4, torch.float64, torch.float64),
                                                     ((33, 2, 13, 17), 3, 2, torch.float32, torch.float64),
                                                     ((5, 1, 4, 4), 4, 4, torch.float64, torch.float64)])
def test_patchify_matches_unfold(shape, k, st, dt_in, dt_out):
    """The patch rows of the image scripts (image_convolution_lanczos_MNIST.py:29-32): F.unfold + transpose + a zero
    row + a zero column + corner one, bit for bit (a byte shuffle)."""
    import torch.nn.functional as Fn
    from tensornetworksfork_amd.tensor.features import patchify
    dev = _dev()
    img = torch.rand(*shape, generator=torch.Generator().manual_seed(3), dtype=dt_in)
    x = Fn.unfold(img, kernel_size=(k, k), stride=(st, st), padding=0).transpose(-2, -1)
    x = torch.cat((x, torch.zeros((x.shape[0], 1, x.shape[2]), dtype=dt_in)), dim=-2)
    x = torch.cat((x, torch.zeros((x.shape[0], x.shape[1], 1), dtype=dt_in)), dim=-1)
    x[..., -1, -1] = 1.0
    got = patchify(img.to(dev), k, st, dtype=dt_out)
    assert got.dtype == dt_out and tuple(got.shape) == tuple(x.shape)
    assert torch.equal(got.cpu(), x.to(dt_out))


@gpu
def test_feature_maps_on_device():
    """polynomial_basis / fbasis of a device input (one kernel for all sites) against the reference's formulas
    (models/tnml.py:11-23) evaluated by torch on the CPU, and the feature-pair classes the polynomial basis
    announces against the ones detected from the data."""
    import math
    from tensornetworksfork_amd.models.tnml import fbasis, polynomial_basis
    from tensornetworksfork_amd.tensor.engine import pair_classes
    dev = _dev()
    X = torch.rand(1237, 7, generator=torch.Generator().manual_seed(4), dtype=torch.float64) * 2 - 1
    for degree in (1, 3, 5, 7):
        got = polynomial_basis(X.to(dev), degree)
        ref = [torch.stack([X[:, i] ** d for d in range(degree + 1)], dim=-1) for i in range(X.shape[1])]
        assert len(got) == len(ref)
        for a, b in zip(got, ref):
            assert a.is_contiguous() and tuple(a.shape) == tuple(b.shape) and G.rel_err(a, b) < 1e-15
        hint = getattr(got[2], "ttk_pair_rep", None)
        assert (None if hint is None else hint[0]) == pair_classes(got[2])
        assert hint is None or hint[1] == got[2]._version
    got = fbasis(X.to(dev))
    ref = [torch.stack([torch.cos(0.5 * math.pi * X[:, i]), torch.sin(0.5 * math.pi * X[:, i])], dim=-1) for i in range(7)]
    for a, b in zip(got, ref):
        assert G.rel_err(a, b) < 1e-15


@gpu
@pytest.mark.parametrize("S,C,rl,p,rr,ca,cb", [(1000, 9, 10, 50, 10, 4, 4), (77, 1, 3, 17, 5, 2, 3), (64, 10, 16, 31, 16, 1, 4),
                                                (129, 2, 1, 7, 13, 4, 1), (5, 3, 9, 128, 4, 3, 2), (4100, 1, 10, 50, 1, 4, 2),
                                                (100, 2, 14, 20, 6, 4, 2), (333, 12, 7, 16, 9, 2, 4)])
def test_conv_bond_kernels_match_einsum(S, C, rl, p, rr, ca, cb):
    """The one-launch halves of the operator of a column between convolution bonds against the plain sum over the
    ca cb Khatri-Rao terms (float64 einsum on the device), plain and accumulating."""
    from tensornetworksfork_amd import ops
    gen = torch.Generator().manual_seed(S + 7 * p + ca)
    dev = _dev()
    mk = lambda *sh: torch.randn(*sh, dtype=torch.float64, generator=gen).to(dev)   # noqa: E731
    L, Gf, R = mk(ca, S, C, rl), mk(ca, cb, S, p), mk(cb, S, rr)
    v, g = mk(rl, p, rr), mk(S, C)
    assert ops.bond_supported(L, Gf, R)
    u_ref = torch.einsum('ascl,absp,lpr,bsr->sc', L, Gf, v, R)
    u = ops.bond_jv(L, Gf, R, v)
    assert G.rel_err(u, u_ref) < 1e-13
    u2 = ops.bond_jv(L, Gf, R, v, out=u_ref.clone())
    assert G.rel_err(u2, 2 * u_ref) < 1e-13
    b_ref = torch.einsum('sc,ascl,absp,bsr->lpr', g, L, Gf, R)
    b = ops.bond_rhs(L, Gf, R, g)
    assert G.rel_err(b, b_ref) < 1e-12
    b2 = ops.bond_rhs(L, Gf, R, g, out=b_ref.clone())
    assert G.rel_err(b2, 2 * b_ref) < 1e-12
    assert torch.equal(b, ops.bond_rhs(L, Gf, R, g))          # fixed order of the slab partials
    Hh = mk(S, C, C)
    H = (Hh + Hh.transpose(1, 2)).contiguous()
    q_ref = torch.einsum('sc,ascl,absp,bsr->lpr', torch.einsum('scd,sc->sd', H, u_ref), L, Gf, R)
    assert G.rel_err(ops.bond_matvec(L, Gf, R, H, v), q_ref) < 1e-12
    assert not ops.bond_supported(mk(2, 8, 1, 17), mk(2, 2, 8, 5), mk(2, 8, 3))       # left bond beyond one tile


@gpu
@pytest.mark.parametrize("S,C,rl,n,ca", [(1000, 10, 10, 68, 4), (77, 1, 3, 5, 2), (4099, 9, 16, 196, 4), (5, 3, 7, 256, 1),
                                          (300, 33, 12, 40, 3)])
def test_factored_filter_operator_matches_rows(S, C, rl, n, ca):
    """J^T H J v, J v and J^T g of a conv filter node between convolution bonds from the factors (L, Z) against the
    same products on the explicit rows J[s,c,(a,k)] = sum_l L_a[s,c,l] Z[s,l,k] (float64 einsum on the device); the
    rows written in place by the strided contraction are checked on the way."""
    from tensornetworksfork_amd import ops
    gen = torch.Generator().manual_seed(S + n + ca)
    dev = _dev()
    mk = lambda *sh: torch.randn(*sh, dtype=torch.float64, generator=gen).to(dev)   # noqa: E731
    L, Z, v, g = mk(ca, S, C, rl), mk(S, rl, n), mk(ca * n), mk(S, C)
    Hh = mk(S, C, C)
    H = (Hh + Hh.transpose(1, 2)).contiguous()
    assert ops.filter_supported(L, Z)
    J = torch.einsum('ascl,slk->scak', L, Z).reshape(S, C, ca * n)
    Jk = torch.empty((S, C, ca, n), dtype=torch.float64, device=dev)
    for a in range(ca):
        ops.lcontract_into(L[a].contiguous(), Z, Jk, a)
    assert G.rel_err(Jk.reshape(S, C, -1), J) < 1e-13
    u_ref = torch.einsum('scd,d->sc', J, v)
    assert G.rel_err(ops.filter_jv(L, Z, v), u_ref) < 1e-12
    b_ref = torch.einsum('scd,sc->d', J, g)
    b = ops.filter_rhs(L, Z, g)
    assert G.rel_err(b, b_ref) < 1e-12
    q_ref = torch.einsum('scd,sc->d', J, torch.einsum('scd,sc->sd', H, u_ref))
    q = ops.filter_matvec(L, Z, H, v)
    assert G.rel_err(q, q_ref) < 1e-12
    assert torch.equal(q, ops.filter_matvec(L, Z, H, v))          # fixed order of the partial sums
    # factors that start 8 bytes off a 16-byte boundary (allowed by the C ABI) go by 4-byte DMA units: same numbers
    off = lambda t: torch.cat([t.new_zeros(1), t.reshape(-1)])[1:].reshape(t.shape)   # noqa: E731
    Lo, Zo, Ho = off(L), off(Z), off(H)
    assert Lo.data_ptr() % 16 == 8 and Zo.is_contiguous()
    assert torch.equal(ops.filter_matvec(Lo, Zo, Ho, v), q)
    assert torch.equal(ops.filter_rhs(Lo, Zo, g), b)
    assert not ops.filter_supported(mk(4, 8, 2, 17), mk(8, 17, 5))              # ca rl beyond one wave


@gpu
@pytest.mark.parametrize("S,C,P,px,nb,dt", [(300, 10, 50, 17, 4, torch.float64), (65, 3, 7, 5, 2, torch.float32),
                                            (4, 16, 65, 49, 3, torch.float32), (129, 1, 4, 4, 1, torch.float64)])
def test_patch_contract_all_channels_in_one_pass(S, C, P, px, nb, dt):
    """The patch contraction for the nb channels of a convolution bond in one pass over the image rows against one
    call per channel (same sums, same order: bit-equal) and against a float64 einsum."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    gen = torch.Generator().manual_seed(S + P)
    M = torch.randn(nb, S, C, P, dtype=torch.float64, generator=gen).to(dev)
    X = torch.randn(S, P, px, dtype=dt, generator=gen).to(dev)
    Z = ops.patch_contract_multi(M, X)
    assert Z is not None and tuple(Z.shape) == (S, C, px, nb)
    ref = torch.einsum('bscp,spx->scxb', M, X.to(torch.float64))
    assert G.rel_err(Z, ref) < 1e-13
    Z1 = torch.empty((S, C, 1, px, nb), dtype=torch.float64, device=dev)
    for b in range(nb):
        ops.patch_contract(M[b].contiguous(), X, Z1, 0, b)
    assert torch.equal(Z1.reshape(S, C, px, nb), Z)
    assert ops.patch_contract_multi(torch.zeros(4, 2, 16, 100, dtype=torch.float64, device=dev),
                                    torch.zeros(2, 100, 49, dtype=torch.float64, device=dev)) is None   # beyond the LDS


@gpu
@pytest.mark.parametrize("S,rl,p,max_iter", [(3000, 32, 6, 21), (700, 16, 4, 8), (260, 16, 8, 5)])
def test_lanczos_blocks_queued_by_one_call_equal_single_iterations(S, rl, p, max_iter):
    """The Lanczos solve with its iterations queued in blocks by one library call (ops.LanczosFused) against the same
    solve with one product call and one vector-kernel call per iteration: the same kernels in the same order, bit-equal."""
    from tensornetworksfork_amd import ops
    from tensornetworksfork_amd.tensor.solvers import lanczos_solve
    dev = _dev()
    L, X, R = _rand((S, 1, rl), 1).to(dev), _rand((S, p), 2).to(dev), _rand((S, rl), 3).to(dev)
    H = (_rand((S, 1, 1), 4).abs() + 0.5).to(dev)
    assert ops.matvec_fused_supported(S, 1, rl, p, rl)
    b, x0 = _rand((rl, p, rl), 5).to(dev), _rand((rl, p, rl), 6).to(dev)
    mv = lambda v: ops.matvec(L, X, R, H, v.reshape(rl, p, rl).contiguous())   # noqa: E731
    plain = lanczos_solve(mv, b, x0, max_iter, 1e-9)
    fused = lanczos_solve(mv, b, x0, max_iter, 1e-9, fused=ops.LanczosFused(L, X, R, H))
    assert torch.equal(plain, fused)
    # and it solves the Galerkin system: the residual is orthogonal to the start residual
    r0 = b - mv(x0)
    res = b - mv(fused)
    assert abs(float((res * r0).sum())) <= 1e-8 * float(r0.norm()) * float(res.norm() + r0.norm())


@gpu
@pytest.mark.parametrize("S,C,rl,n,ca,nb", [(500, 10, 10, 100, 4, 4), (77, 3, 5, 7, 2, 3), (33, 1, 16, 256, 1, 2)])
def test_left_step_sum_over_channels_in_one_launch(S, C, rl, n, ca, nb):
    """out[b] = sum_a L_a . T_ab in one launch against one accumulating contraction per term."""
    from tensornetworksfork_amd import ops
    dev = _dev()
    gen = torch.Generator().manual_seed(S + n)
    L = torch.randn(ca, S, C, rl, dtype=torch.float64, generator=gen).to(dev)
    T = torch.randn(ca, nb, S, rl, n, dtype=torch.float64, generator=gen).to(dev)
    out = ops.lcontract_sum(L, T)
    assert out is not None and tuple(out.shape) == (nb, S, C, n)
    ref = torch.einsum('ascl,absln->bscn', L, T)
    assert G.rel_err(out, ref) < 1e-13
    for b in range(nb):
        o = None
        for a in range(ca):
            o = ops.lcontract(L[a], T[a, b].contiguous()) if o is None else ops.lcontract(L[a], T[a, b].contiguous(), out=o)
        assert G.rel_err(out[b], o) < 1e-14
