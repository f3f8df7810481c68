"""CPU restatement of the reference's DMRG core-update path (TEST INFRASTRUCTURE).

This file is the oracle: a plain, fixed-layout, torch-CPU restatement of the
arithmetic the reference performs through generated einsum strings.  It is used
only as a checker (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
leg); the product path under tensornetworksfork_amd/ never imports it.

Parity pinning: the reference ships no tests or golden vectors for this path,
so the restatement is pinned against the reference itself, imported in the
build container by oracle/load_reference.py; the outputs are committed under
tests/golden/ by oracle/gen_golden.py and checked by tests/test_oracle_golden.py.

Index names: s sample, c/d output, l/m left bond, p/q physical, r/t right bond.
Canonical site layout used throughout: a core is (rl, p, rr).  The first core
of the reference, shaped (C, p, rr) (tensor/layers.py:77-94 after squeeze),
is the canonical core with rl == C and an identity left environment; the last
core (rl, p) is canonical with rr == 1 and an all-ones right environment.
"""
import math

import torch


# --------------------------------------------------------------------------
# model construction (tensor/layers.py:9-97, 194-221; tensor/node.py:7-11)
# --------------------------------------------------------------------------
def bond_schedule(N, r, f, constrict_bond=True):
    """(left, right) bond sizes per site; tensor/layers.py:17-30,58-75."""
    if N == 1:
        r = 1

    def build_left(b0, right=0):
        mx = min(r, b0 * f) if constrict_bond else r
        return (b0, right if right != 0 else mx)

    def build_right(b1, left=0):
        mx = min(r, b1 * f) if constrict_bond else r
        return (left if left != 0 else mx, b1)

    b0 = build_left(1)
    bn = build_right(1)
    left_stack, right_stack = [b0], [bn]
    middle = [b0, bn]
    for i in range(N - 2):
        left_r = left_stack[-1][1]
        right_r = right_stack[0][0]
        if i == N - 3:
            middle = [*left_stack, (left_r, right_r), *right_stack]
        if i % 2 == 0:
            left_stack.append(build_left(left_r))
        else:
            right_stack.insert(0, build_right(right_r))
    if N == 1:
        middle = [b0]
    return middle


def init_tt_cores(N, r, f, output_dim=1, constrict_bond=True, seed=None, dtype=None):
    """Random cores in the reference's draw order (tensor/layers.py:203-217).

    Each node is randn(shape)/||.|| drawn as (left, up, down, right) with
    up = output_dim on the first site and 1 elsewhere (layers.py:77-93), then
    squeezed: site 1 -> (C, p, rr); middle -> (rl, p, rr); last -> (rl, p).
    The N dummy input nodes randn(1, f) are drawn afterwards (layers.py:212)
    and discarded here; they only advance the generator.
    """
    if seed is not None:
        torch.manual_seed(seed)
    cores = []
    for i, (left, right) in enumerate(bond_schedule(N, r, f, constrict_bond)):
        up = output_dim if i == 0 else 1
        t = torch.randn((left, up, f, right), dtype=dtype)
        t = t / torch.norm(t)
        if i == 0:
            t = t[0] if N > 1 else t[0, :, :, 0]
        elif i == N - 1:
            t = t[:, 0, :, 0]
        else:
            t = t[:, 0]
        cores.append(t)
    for _ in range(N):
        d = torch.randn((1, f), dtype=dtype)
        d = d / torch.norm(d)
    return cores


def canon(cores):
    """Canonical (rl, p, rr) views of reference-shaped cores."""
    N = len(cores)
    out = []
    for k, t in enumerate(cores):
        if k == N - 1 and t.ndim == 2:
            t = t.unsqueeze(-1)
        out.append(t)
    return out


def site_inputs(x, N):
    """The reference feeds either one tensor to every site or a list with one
    tensor per site (tensor/network.py:329-345)."""
    if isinstance(x, (list, tuple)):
        return list(x)
    return [x] * N


# --------------------------------------------------------------------------
# feature maps (models/tnml.py:11-23)
# --------------------------------------------------------------------------
def polynomial_basis(X, degree=3):
    return [torch.stack([X[:, i] ** d for d in range(degree + 1)], dim=-1) for i in range(X.shape[-1])]


def fbasis(X):
    return [torch.stack([torch.cos((0.5 * math.pi) * X[:, i]), torch.sin((0.5 * math.pi) * X[:, i])], dim=-1)
            for i in range(X.shape[-1])]


# --------------------------------------------------------------------------
# environments and prediction (tensor/network.py:55-71, 115-137; node.py:28-45)
# --------------------------------------------------------------------------
def env_left_step(Lprev, core, X):
    """L'[s,c,r] = sum_{l,p} L[s,c,l] core[l,p,r] X[s,p]."""
    T = torch.einsum('scl,lpr->scpr', Lprev, core)
    return torch.einsum('scpr,sp->scr', T, X)


def env_right_step(Rnext, core, X):
    """R'[s,l] = sum_{p,r} core[l,p,r] X[s,p] R[s,r]."""
    T = torch.einsum('lpr,sr->slp', core, Rnext)
    return torch.einsum('slp,sp->sl', T, X)


def left_identity(S, C, dtype):
    return torch.eye(C, dtype=dtype).unsqueeze(0).expand(S, C, C)


def left_envs(cores, xs):
    """L[k] = environment entering site k from the left (k = 0..N-1); L[0] is
    the identity over the output index."""
    cc = canon(cores)
    S = xs[0].shape[0]
    envs = [left_identity(S, cc[0].shape[0], cc[0].dtype)]
    for k in range(len(cc) - 1):
        envs.append(env_left_step(envs[-1], cc[k], xs[k]))
    return envs


def right_envs(cores, xs):
    """R[k] = environment entering site k from the right; R[N-1] is all ones."""
    cc = canon(cores)
    N = len(cc)
    S = xs[0].shape[0]
    envs = [None] * N
    envs[N - 1] = torch.ones((S, 1), dtype=cc[0].dtype)
    for k in range(N - 1, 0, -1):
        envs[k - 1] = env_right_step(envs[k], cc[k], xs[k])
    return envs


def predict(cores, x):
    """y[s,c]; the reference contracts the first column with the right stack
    (tensor/network.py:121-131)."""
    cc = canon(cores)
    xs = site_inputs(x, len(cc))
    R = right_envs(cores, xs)
    S = xs[0].shape[0]
    L1 = env_left_step(left_identity(S, cc[0].shape[0], cc[0].dtype), cc[0], xs[0])
    return torch.einsum('scr,sr->sc', L1, R[0])


def forward_batch(cores, x, batch_size):
    """tensor/network.py:139-150."""
    xs = site_inputs(x, len(cores))
    S = xs[0].shape[0]
    if batch_size <= 0 or batch_size >= S:
        return predict(cores, xs)
    outs = []
    for b0 in range(0, S, batch_size):
        outs.append(predict(cores, [t[b0:b0 + batch_size] for t in xs]))
    return torch.cat(outs, dim=0)


# --------------------------------------------------------------------------
# losses (tensor/bregman.py)
# --------------------------------------------------------------------------
def loss_square(y_pred, y):
    """SquareBregFunction, tensor/bregman.py:9-23,34-52.  The Hessian is
    stored (S, C, 1); inside the Gram einsum that size-1 axis broadcasts over
    the second output index, i.e. an all-ones C x C coupling with weight 2
    (tensor/network.py:188,212).  Returned here already expanded to (S, C, C)."""
    x = y_pred.flatten(1) if y_pred.ndim > 1 else y_pred
    t = y.flatten(1) if y.ndim > 1 else y
    loss = (x ** 2).sum(-1) - (t ** 2).sum(-1) - ((2 * t) * (x - t)).sum(-1)
    g = 2 * x - 2 * t
    C = x.shape[-1]
    H = torch.full((x.shape[0], C, C), 2.0, dtype=x.dtype)
    return loss, g, H


def loss_mse(y_pred, y):
    """AutogradLoss(MSELoss(reduction='none')), tensor/bregman.py:266-292:
    element-wise loss (S, C), gradient 2(x-y), Hessian diag(2)."""
    loss = (y_pred - y) ** 2
    g = 2 * (y_pred - y)
    C = y_pred.shape[-1]
    H = (2.0 * torch.eye(C, dtype=y_pred.dtype)).unsqueeze(0).expand(y_pred.shape[0], C, C).contiguous()
    return loss, g, H


def loss_xe(y_pred, y, w=1.0, hard=False):
    """Softmax cross entropy in closed form: logits z=[w x, 0], pi = softmax(z),
    loss = -log pi[label], label = argmax y, H = w^2 (diag(pi) - pi pi^T)[:C,:C].
    KLDivBregman (tensor/bregman.py:121-146) has g = w (pi - y)[:C]; XEAutogradBregman
    (bregman.py:189-216) differentiates cross_entropy(z, label), i.e. g = w (pi - onehot(label))[:C]
    (hard=True).  For one-hot y over C+1 classes the two coincide."""
    z = torch.cat((w * y_pred, torch.zeros_like(y_pred[..., :1])), dim=-1)
    log_s = torch.log_softmax(z, dim=-1)
    label = y.argmax(dim=-1)
    loss = -log_s.gather(-1, label.unsqueeze(-1)).squeeze(-1)
    s = log_s.exp()
    target = torch.nn.functional.one_hot(label, y.shape[-1]).to(s.dtype) if hard else y
    g = w * (s - target)[..., :-1]
    H = (w ** 2) * (torch.diag_embed(s) - s.unsqueeze(-1) * s.unsqueeze(-2))[..., :-1, :-1]
    return loss, g, H


LOSSES = {'square': loss_square, 'mse': loss_mse, 'xe': loss_xe}


# --------------------------------------------------------------------------
# local normal equations (tensor/network.py:101-113, 174-217, 258-291)
# --------------------------------------------------------------------------
def jacobian(Lenv, X, Renv):
    """J[s,c,l,p,r] = L[s,c,l] X[s,p] R[s,r] (row-wise Khatri-Rao product)."""
    return torch.einsum('scl,sp,sr->sclpr', Lenv, X, Renv)


def _sample_chunks(S, C, D, budget_bytes=1 << 30):
    """Sample ranges whose explicit Jacobian rows (C*D doubles per sample) stay below the budget:
    sums over samples are taken chunk by chunk so that large sites fit in host memory."""
    step = max(1, int(budget_bytes // (8 * C * D)))
    return [(lo, min(S, lo + step)) for lo in range(0, S, step)]


def local_gram_rhs(Lenv, X, Renv, g, H):
    """A[lpr,mqt] = sum_{s,c,d} H[s,c,d] J[s,c,lpr] J[s,d,mqt];
    b[lpr] = sum_{s,c} g[s,c] J[s,c,lpr].  Evaluated as two GEMMs per sample chunk,
    not through the reference's (S, D, D) temporary."""
    S, C, rl = Lenv.shape
    p, rr = X.shape[1], Renv.shape[1]
    D = rl * p * rr
    A = torch.zeros(D, D, dtype=Lenv.dtype)
    b = torch.zeros(D, dtype=Lenv.dtype)
    for lo, hi in _sample_chunks(S, C, D):
        n = hi - lo
        J = jacobian(Lenv[lo:hi], X[lo:hi], Renv[lo:hi]).reshape(n, C, D)
        HJ = torch.einsum('scd,sdk->sck', H[lo:hi], J)
        A.addmm_(J.reshape(n * C, D).T, HJ.reshape(n * C, D))
        b.add_(J.reshape(n * C, D).T @ g[lo:hi].reshape(n * C))
    return A.reshape(rl, p, rr, rl, p, rr), b.reshape(rl, p, rr)


def local_rhs(Lenv, X, Renv, g):
    S, C, rl = Lenv.shape
    D = rl * X.shape[1] * Renv.shape[1]
    b = torch.zeros(D, dtype=Lenv.dtype)
    for lo, hi in _sample_chunks(S, C, D):
        J = jacobian(Lenv[lo:hi], X[lo:hi], Renv[lo:hi]).reshape((hi - lo) * C, D)
        b.add_(J.T @ g[lo:hi].reshape((hi - lo) * C))
    return b.reshape(rl, X.shape[1], Renv.shape[1])


def local_matvec(Lenv, X, Renv, H, v):
    """(A v)[lpr] = sum_{s,c} J[s,c,lpr] u[s,c],
    u[s,d] = sum_c H[s,c,d] sum_{lpr} J[s,c,lpr] v[lpr]  (network.py:788-789)."""
    S, C, rl = Lenv.shape
    D = rl * X.shape[1] * Renv.shape[1]
    out = torch.zeros(D, dtype=Lenv.dtype)
    for lo, hi in _sample_chunks(S, C, D):
        J = jacobian(Lenv[lo:hi], X[lo:hi], Renv[lo:hi]).reshape(hi - lo, C, D)
        Jv = J @ v.reshape(-1)
        u = torch.einsum('scd,sc->sd', H[lo:hi], Jv)
        out.add_(torch.einsum('sck,sc->k', J, u))
    return out.reshape(rl, X.shape[1], Renv.shape[1])


# --------------------------------------------------------------------------
# solve and update (tensor/network.py:293-327; tensor/node.py:178-203)
# --------------------------------------------------------------------------
def solve_system(A, b, w, method='exact', eps=0.0):
    """Scale by mean |diag|, then ridge 2*eps on the scaled system with the
    UNscaled current weights in the right-hand side, x = -A^-1 b."""
    if method.lower() == 'gradient':
        return (-b).reshape(b.shape)
    D = b.numel()
    A_f = A.reshape(D, D)
    b_f = b.reshape(D)
    scale = A_f.diag().abs().mean()
    if scale == 0:
        scale = 1
    A_f = A_f / scale
    b_f = b_f / scale
    m = method.lower()
    if m == 'exact':
        x = torch.linalg.solve(A_f, -b_f)
    elif m == 'ridge_exact':
        A_f = A_f + (2 * eps) * torch.eye(D, dtype=A_f.dtype)
        b_f = b_f + (2 * eps) * w.reshape(D)
        x = torch.linalg.solve(A_f, -b_f)
    elif m.startswith('ridge_cholesky'):
        A_f = A_f + (2 * eps) * torch.eye(D, dtype=A_f.dtype)
        b_f = b_f + (2 * eps) * w.reshape(D)
        Lc = torch.linalg.cholesky(A_f)
        x = torch.cholesky_solve(-b_f.unsqueeze(-1), Lc).squeeze(-1)
    elif m == 'cholesky':
        Lc = torch.linalg.cholesky(A_f)
        x = torch.cholesky_solve(-b_f.unsqueeze(-1), Lc).squeeze(-1)
    elif m == 'gradient':
        x = -b
    else:
        raise ValueError(f"Unknown method: {method}")
    return x.reshape(b.shape)


def update_core(w, step, lr=1.0, adaptive_step=False, max_norm=None):
    if adaptive_step:
        sn, pn = torch.norm(step), torch.norm(w)
        if sn > pn:
            step = step * (pn / sn)
    new = w + lr * step
    if max_norm is not None:
        cn = torch.norm(new)
        if cn > max_norm:
            new = new * (max_norm / cn)
    return new


# --------------------------------------------------------------------------
# QR gauge (tensor/network.py:625-707)
# --------------------------------------------------------------------------
def orthonormalize_left_site(cores, k):
    """QR of core k as ((rl p), rr); R is pushed into core k+1.  Cores are in
    reference shapes; modified in place in the list."""
    N = len(cores)
    if k >= N - 1:
        return
    t = cores[k]
    M = t.reshape(-1, t.shape[-1])
    Q, R = torch.linalg.qr(M, mode='reduced')
    cores[k] = Q.reshape(t.shape[:-1] + (Q.shape[-1],))
    cores[k + 1] = torch.einsum('ij,j...->i...', R, cores[k + 1])


def orthonormalize_right_site(cores, k):
    """RQ emulated by flips (network.py:683-689); R pushed into core k-1."""
    if k <= 0:
        return
    t = cores[k]
    nd = t.ndim
    perm = list(range(1, nd)) + [0]
    P = t.permute(perm)
    M = P.reshape(-1, P.shape[-1])
    Qr, Rr = torch.linalg.qr(torch.flip(M, dims=[0, 1]), mode='reduced')
    R = torch.flip(Rr.T, dims=[0, 1])
    Q = torch.flip(Qr, dims=[0, 1])
    Q = Q.reshape(P.shape[:-1] + (Q.shape[-1],))
    inv = sorted(range(nd), key=lambda i: perm[i])
    cores[k] = Q.permute(inv)
    cores[k - 1] = torch.einsum('ji,...j->...i', R, cores[k - 1])


# --------------------------------------------------------------------------
# the sweep driver (tensor/network.py:379-608), plain TT chain
# --------------------------------------------------------------------------
def site_problem(cores, xs, k, y, loss_fn, want_A=True, loss_kwargs=None):
    """Everything one mini-batch contributes for site k: (A, b, loss_mean, g, H).
    Environments are rebuilt from scratch, as the reference does for every
    mini-batch (network.py:456 -> 117 -> 342-344)."""
    cc = canon(cores)
    Ls = left_envs(cores, xs)
    Rs = right_envs(cores, xs)
    L1 = env_left_step(Ls[0], cc[0], xs[0])
    y_pred = torch.einsum('scr,sr->sc', L1, Rs[0])
    loss, g, H = LOSSES[loss_fn](y_pred, y, **(loss_kwargs or {}))
    if want_A:
        A, b = local_gram_rhs(Ls[k], xs[k], Rs[k], g, H)
    else:
        A, b = None, local_rhs(Ls[k], xs[k], Rs[k], g)
    return A, b, loss.mean(), g, H


def _ref_shape(cores, k, t):
    """Canonical (rl,p,rr) tensor -> the reference's shape for site k."""
    return t.reshape(cores[k].shape)


def accumulating_swipe(cores, x, y, loss_fn='square', batch_size=-1, num_swipes=1, lr=1.0,
                       method='exact', eps=1e-12, eps_decay=None, orthonormalize=False,
                       skip_second=False, direction='l2r', eps_per_node=False,
                       adaptive_step=False, max_norm=None, loss_kwargs=None, trace=None,
                       convergence_criterion=None):
    """Returns (ok, cores).  `trace`, if a list, receives (NS, site, loss, core
    snapshot list) after every update (the reference's loss_callback /
    block_callback points, network.py:493-505)."""
    cores = [c.clone() for c in cores]
    N = len(cores)
    xs = site_inputs(x, N)
    S = xs[0].shape[0]
    if batch_size <= 0:
        batch_size = S
    batches = (S + batch_size - 1) // batch_size
    last_l2r, last_r2l = None, None
    NS = 0

    def one_update(k, eps_, NS, leftward):
        A_out, b_out, total = None, None, 0.0
        for bi in range(batches):
            sl = slice(bi * batch_size, (bi + 1) * batch_size)
            xb = [t[sl] for t in xs]
            A, b, lm, _, _ = site_problem(cores, xb, k, y[sl], loss_fn, want_A=(method != 'gradient'),
                                          loss_kwargs=loss_kwargs)
            if b_out is None:
                A_out, b_out = A, b
            else:
                if A is not None:
                    A_out = A_out + A
                b_out = b_out + b
            if method == 'gradient' and not leftward:
                cores[k] = update_core(cores[k], _ref_shape(cores, k, b), lr, adaptive_step, max_norm)
            total += lm.item()
        _method = 'exact' if (eps_ == 0 and method == 'ridge_exact') else method
        w = canon(cores)[k]
        try:
            step = solve_system(A_out, b_out, w, _method, eps_)
        except torch.linalg.LinAlgError:
            return None
        if method != 'gradient' or leftward:
            cores[k] = update_core(cores[k], _ref_shape(cores, k, step), lr, adaptive_step, max_norm)
        if orthonormalize:
            (orthonormalize_right_site if leftward else orthonormalize_left_site)(cores, k)
        if trace is not None:
            trace.append((NS, k, total / batches, [c.clone() for c in cores]))
        return True

    for _ in range(num_swipes):
        eps_ = eps[NS] if isinstance(eps, list) else eps
        if eps_decay is not None:
            eps_ = eps_ * eps_decay ** NS
        order = list(range(N)) if direction == 'l2r' else list(reversed(range(N)))
        for i, k in enumerate(order):
            if eps_per_node:
                eps_ = eps[i if direction == 'l2r' else len(order) - 1 - i] if isinstance(eps, list) else eps
            last_l2r = k          # the loop variable is bound before the skip test (network.py:426-433)
            if last_r2l is not None and k == last_r2l:
                continue
            if one_update(k, eps_, NS, leftward=False) is None:
                return False, cores
            if convergence_criterion is not None and convergence_criterion(cores):
                return True, cores
        NS += 1
        if skip_second:
            continue
        eps_ = eps[NS] if isinstance(eps, list) else eps
        if eps_decay is not None:
            eps_ = eps_ * eps_decay ** NS
        order = list(range(N)) if direction == 'r2l' else list(reversed(range(N)))
        for i, k in enumerate(order):
            if eps_per_node:
                eps_ = eps[i if direction == 'r2l' else len(order) - 1 - i] if isinstance(eps, list) else eps
            last_r2l = k
            if last_l2r is not None and k == last_l2r:
                continue
            if one_update(k, eps_, NS, leftward=True) is None:
                return False, cores
            if convergence_criterion is not None and convergence_criterion(cores):
                return True, cores
        NS += 1
    return True, cores


# --------------------------------------------------------------------------
# matrix-free solvers (tensor/network.py:709-932)
# --------------------------------------------------------------------------
def lanczos_solver(matvec, b, x0, max_iter, tol):
    """Lanczos-Galerkin without re-orthogonalisation, network.py:796-824."""
    v = [torch.zeros_like(x0)]
    a = [0.0]
    bc = [0.0]
    r0 = b - matvec(x0)
    beta1 = torch.norm(r0)
    bc.append(beta1)
    v.append(r0 / beta1)
    j = 0
    for j in range(1, max_iter + 1):
        w = matvec(v[j]) - bc[j] * v[j - 1]
        a_j = (w * v[j]).sum()
        a.append(a_j)
        w = w - a_j * v[j]
        beta = torch.norm(w)
        bc.append(beta)
        v.append(w / beta)
        if beta < tol:
            break
    Vm = torch.stack(v[1:j + 1], dim=-1)
    Tm = torch.diag(torch.tensor(a[1:], dtype=x0.dtype))
    if len(a) > 2:
        off = torch.tensor(bc[2:j + 1], dtype=x0.dtype)
        Tm = Tm + torch.diag(off, diagonal=1) + torch.diag(off, diagonal=-1)
    rhs = torch.zeros(len(a) - 1, dtype=x0.dtype)
    rhs[0] = beta1
    yv = torch.linalg.solve(Tm, rhs)
    return x0 + Vm @ yv


def krylov_site_setup(cores, xs, k, y, loss_fn, batch_size, loss_kwargs=None):
    """RHS phase of lanczos_swipe / scipy_swipe (network.py:746-767, 870-892):
    returns (b_rhs in reference core shape, matvec(v), mean-of-batch-means loss)."""
    S = xs[0].shape[0]
    batches = (S + batch_size - 1) // batch_size
    cc = canon(cores)
    b_rhs = torch.zeros_like(cc[k])
    per_batch = []
    total = 0.0
    for bi in range(batches):
        sl = slice(bi * batch_size, (bi + 1) * batch_size)
        xb = [t[sl] for t in xs]
        _, b, lm, g, H = site_problem(cores, xb, k, y[sl], loss_fn, want_A=False, loss_kwargs=loss_kwargs)
        b_rhs = b_rhs + b
        total += lm.item()
        Ls = left_envs(cores, xb)
        Rs = right_envs(cores, xb)
        per_batch.append((Ls[k], xb[k], Rs[k], H))

    def matvec(v):
        out = 0
        vv = v.reshape(cc[k].shape)
        for (Le, Xe, Re, H) in per_batch:
            out = out + local_matvec(Le, Xe, Re, H, vv)
        return out.reshape(v.shape)

    return b_rhs.reshape(cores[k].shape), matvec, total / batches


def lanczos_swipe(cores, x, y, loss_fn='square', batch_size=1, num_swipes=1, lr=1.0, max_iter=50,
                  tol=1e-6, x0s=None, loss_kwargs=None, trace=None):
    """network.py:709-832.  x0s: list of start vectors, one per visited node in
    visiting order (the reference draws randn_like(b_rhs), network.py:793); when
    None they are drawn from the global torch generator exactly like that."""
    cores = [c.clone() for c in cores]
    N = len(cores)
    xs = site_inputs(x, N)
    if batch_size <= 0:
        batch_size = xs[0].shape[0]
    visit = 0
    for NS in range(num_swipes):
        order = range(N) if NS % 2 == 0 else reversed(range(N))
        for k in order:
            b_rhs, matvec, loss = krylov_site_setup(cores, xs, k, y, loss_fn, batch_size, loss_kwargs)
            x0 = torch.randn_like(b_rhs) if x0s is None else x0s[visit]
            visit += 1
            step = lanczos_solver(matvec, -b_rhs, x0, max_iter, tol)
            cores[k] = cores[k] + lr * step
            if trace is not None:
                trace.append((NS, k, loss, [c.clone() for c in cores]))
    return True, cores


def scipy_swipe(cores, x, y, solver, loss_fn='square', batch_size=1, num_swipes=1, lr=1.0,
                max_iter=50, tol=1e-6, loss_kwargs=None, trace=None):
    """network.py:834-932: the operator is handed to scipy.sparse.linalg.cg /
    minres; vectors cross the SciPy boundary as float32 (network.py:918, 921)
    and each node keeps its last solution as the next warm start (924-925)."""
    from scipy.sparse.linalg import LinearOperator
    cores = [c.clone() for c in cores]
    N = len(cores)
    xs = site_inputs(x, N)
    if batch_size <= 0:
        batch_size = xs[0].shape[0]
    sols = {}
    for NS in range(num_swipes):
        order = range(N) if NS % 2 == 0 else reversed(range(N))
        for k in order:
            b_rhs, matvec, loss = krylov_site_setup(cores, xs, k, y, loss_fn, batch_size, loss_kwargs)
            dt = b_rhs.dtype

            def mv(v_np):
                v = torch.tensor(v_np, dtype=dt).reshape(b_rhs.shape)
                return matvec(v).flatten().float().numpy()

            b_np = b_rhs.flatten().float().numpy()
            A_op = LinearOperator((b_np.shape[0], b_np.shape[0]), matvec=mv)
            x_sol, info = solver(A_op, -b_np, x0=sols.get(k), maxiter=max_iter, rtol=tol)
            sols[k] = x_sol
            step = torch.tensor(x_sol, dtype=dt).reshape(b_rhs.shape)
            cores[k] = cores[k] + lr * step
            if trace is not None:
                trace.append((NS, k, loss, [c.clone() for c in cores]))
    return True, cores
