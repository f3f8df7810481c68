"""TEST INFRASTRUCTURE ONLY: torch-CPU stand-ins for the HIP entry points behind
tensornetworksfork_amd.ops, installed by the `cpu_kernels` fixture (monkeypatch) so that the
host-side logic (topology planning, environment caching, sweep schedule, sharding) can be
exercised on a machine without a GPU.  Nothing in the package imports this module; without the
fixture every op raises TTKError on CPU tensors."""
import torch

F64 = torch.float64


def env_step_left(E, X, core, out=None):
    r = torch.einsum('scl,sp,lpr->scr', E, X, core)
    return r if out is None else out.add_(r)


def env_step_right(R, X, core, out=None, mm32=False):
    r = torch.einsum('sr,sp,lpr->sl', R, X, core)
    return r if out is None else out.add_(r)


def predict(L, R, out=None):
    r = torch.einsum('scr,sr->sc', L, R)
    return r if out is None else out.add_(r)


def loss(kind, y_pred, y, w=1.0):
    from oracle import tt_oracle as O
    lo, g, H = O.loss_xe(y_pred, y, w, hard=(kind == "xe_hard")) if kind.startswith("xe") else O.LOSSES[kind](y_pred, y)
    S, C = y_pred.shape
    return lo, g.reshape(S, C).contiguous(), H.reshape(S, C, C).contiguous()


def batch_mean(v, batch):
    S = v.shape[0]
    means = [v[i:i + batch].mean() for i in range(0, S, batch)]
    return torch.stack(means).mean()


def kr_rows(L, X, R, out=None):
    S, C, _ = L.shape
    r = torch.einsum('scl,sp,sr->sclpr', L, X, R).reshape(S, C, -1)
    return r if out is None else out.add_(r)


def hess_left(H, L):
    return torch.einsum('scd,sdl->scl', H, L)


class _WS:
    def __init__(self, A, dims):
        self.A = A
        self.dims = dims
        self.buf = A
        self.generation = 1

    @property
    def compressed(self):
        return self.A


def gram(L, H, X, R, sym=True, ws=None, pair_rep=None):
    S, C, rl = L.shape
    J = kr_rows(L, X, R)
    A = torch.einsum('scd,sce,sef->df', J, H, J)
    return _WS(A.contiguous(), (S, C, rl, X.shape[1], R.shape[1], int(bool(sym))))


def gram_supported(S, C, rl, p, rr, sym=True):
    return True


def gram_scale(ws):
    s = ws.A.diagonal().abs().mean()
    return torch.where(s == 0, torch.ones_like(s), s)


def gram_expand(ws, scale=None, ridge=0.0, out=None):
    A = ws.A if scale is None else ws.A / scale
    A = A + ridge * torch.eye(A.shape[0], dtype=F64)
    if out is not None:
        out.copy_(A)
        return out
    return A


def rhs(L, X, R, g, out=None):
    r = torch.einsum('sc,scl,sp,sr->lpr', g, L, X, R)
    return r if out is None else out.add_(r)


def rhs_feature(L, X, R, g, out=None, mm32=False, GL=None):
    if GL is not None:
        r = torch.einsum('sl,sp,sr->lpr', GL, X, R)
        return r if out is None else out.add_(r)
    return rhs(L, X, R, g, out=out)


def jvp(L, X, R, H, v):
    u = torch.einsum('scl,sp,sr,lpr->sc', L, X, R, v)
    return torch.einsum('scd,sd->sc', H, u)


def matvec(L, X, R, H, v):
    return rhs(L, X, R, jvp(L, X, R, H, v))


def solve_rhs(b, w=None, scale=None, ridge=0.0):
    r = b.reshape(-1) if scale is None else b.reshape(-1) / scale
    if w is not None:
        r = r + ridge * w.reshape(-1)
    return -r


def chol_solve_gram(ws, b, w=None, scale=None, ridge=0.0):
    A = gram_expand(ws, scale, ridge)
    Lc, info = torch.linalg.cholesky_ex(A)
    if int(info) != 0:
        return torch.zeros(A.shape[0], dtype=F64), info.to(torch.int32)
    x = torch.cholesky_solve(solve_rhs(b, w, scale, ridge).unsqueeze(-1), Lc).squeeze(-1)
    return x, info.to(torch.int32)


def qr(A):
    return torch.linalg.qr(A, mode='reduced')


def dot(x, y):
    return (x * y).sum()


def axpy(a, x, y, sign=1.0):
    return y.add_(sign * a * x)


def rhs_supported(S, C, rl, p, rr):
    return True


def conv_feature(X, Fc):
    return torch.einsum('spx,axb->absp', X.to(F64), Fc).contiguous()


def gemm_rows(A, B, N, bmode=0, rl=0, rr=0, out=None, mm32=False):
    if bmode == 0:
        T = A @ B
    elif bmode == 1:
        T = torch.einsum('sk,lkr->slr', A, B).reshape(A.shape[0], N)
    else:
        T = A @ B.T
    return T if out is None else out.add_(T)


def lcontract(L, T, out=None):
    r = torch.einsum('scl,slj->scj', L, T)
    return r if out is None else out.add_(r)


def lcontract_sum(L, T):
    return torch.einsum('ascl,absln->bscn', L, T).contiguous()


def lcontract_into(L, T, J, a):
    J[:, :, a, :] = torch.einsum('scl,sln->scn', L, T)
    return J


def env_step_left_dense(E, X, core, out=None, mm32=False):
    return env_step_left(E, X, core, out=out)


def patch_contract(M, X, J, a=0, b=0):
    J[:, :, a, :, b] = torch.einsum('scp,spx->scx', M, X.to(F64))
    return J


def patch_contract_multi(M, X):
    return torch.einsum('bscp,spx->scxb', M, X.to(F64)).contiguous()


def gemm_tn(A, B, out=None):
    r = A.T @ B
    return r if out is None else out.add_(r)


def chol_solve_dense(A, b, w=None, ridge=0.0):
    s = A.diagonal().abs().mean()
    s = torch.where(s == 0, torch.ones_like(s), s)
    M = A / s + ridge * torch.eye(A.shape[0], dtype=F64)
    Lc, info = torch.linalg.cholesky_ex(M)
    if int(info) != 0:
        return torch.zeros(A.shape[0], dtype=F64), info.to(torch.int32)
    x = torch.cholesky_solve(solve_rhs(b, w, s, ridge).unsqueeze(-1), Lc).squeeze(-1)
    return x, info.to(torch.int32)


def matvec_fused_supported(S, C, rl, p, rr):
    return False


def gl(g, L):
    return torch.einsum('sc,scl->sl', g, L)


def rows_dot(J, v):
    return torch.einsum('scd,d->sc', J, v)


def rows_matvec_supported(S, C, D):
    return D <= 512 and C <= 32 and C * D <= 4096


def rows_matvec(J, H, v):
    u = torch.einsum('scd,d->sc', J, v)
    hu = torch.einsum('scd,sc->sd', H, u)
    return torch.einsum('scd,sc->d', J, hu)


def rows_rhs(J, g):
    return torch.einsum('scd,sc->d', J, g)


def lhl(L, H, z):
    u = torch.einsum('scl,sl->sc', L, z)
    hu = torch.einsum('scd,sc->sd', H, u)
    return torch.einsum('scl,sc->sl', L, hu)


def filter_supported(L, Z):
    return L.is_contiguous() and Z.is_contiguous() and L.shape[0] * L.shape[3] <= 64 and Z.shape[2] <= 256


def filter_jv(L, Z, v):
    return torch.einsum('ascl,slk,ak->sc', L, Z, v.reshape(L.shape[0], -1))


def filter_rhs(L, Z, g):
    return torch.einsum('ascl,slk,sc->ak', L, Z, g).reshape(-1)


def filter_matvec(L, Z, H, v):
    u = filter_jv(L, Z, v)
    return filter_rhs(L, Z, torch.einsum('scd,sc->sd', H, u))


def bond_supported(L, G, R):
    return L.is_contiguous() and G.is_contiguous() and R.is_contiguous() and L.shape[3] <= 16 and R.shape[2] <= 16 \
        and L.shape[0] <= 4 and R.shape[0] <= 4 and G.shape[3] <= 128


def bond_jv(L, G, R, v, out=None):
    u = torch.einsum('ascl,absp,lpr,bsr->sc', L, G, v, R)
    return u if out is None else out.add_(u)


def bond_rhs(L, G, R, g, out=None):
    b = torch.einsum('sc,ascl,absp,bsr->lpr', g, L, G, R)
    return b if out is None else out.add_(b)


def bond_z_supported(G, R, core):
    return G.dim() == 4 and G.is_contiguous() and R.is_contiguous() and core.shape[0] <= 16 and core.shape[2] <= 16 \
        and G.shape[0] <= 4 and G.shape[1] <= 4 and G.shape[3] <= 128


def bond_z(G, R, core):
    return torch.einsum('lpr,absp,bsr->asl', core, G, R).contiguous()


def bond_matvec(L, G, R, H, v):
    u = bond_jv(L, G, R, v)
    return bond_rhs(L, G, R, torch.einsum('scd,sc->sd', H, u))


def cg_update(x, r, p, q, state):
    rho, atol = state[0].clone(), state[1]
    if bool(rho.sqrt() < atol):
        state[2] = 1.0
        return
    alpha = rho / (p * q).sum()
    x.add_(alpha * p)
    r.sub_(alpha * q)
    rho_new = (r * r).sum()
    p.copy_(r + (rho_new / rho) * p)
    state[0] = rho_new
    state[3] += 1.0


def minres_update(y, r1, r2, v, w, w2, x, st):
    """scipy.sparse.linalg.minres (SciPy 1.15 _isolve/minres.py), one pass of its loop body on
    the state vector layout of ttk_minres_update_f64."""
    import math
    if float(st[0]) != 0.0:
        return
    (_, itn, oldb, beta, dbar, epsln, phibar, rhs1, rhs2, tnorm2, gmax, gmin, cs, sn, beta1, rtol, maxiter,
     eps) = [float(t) for t in st]
    itn += 1
    if itn >= 2:
        y.sub_((beta / oldb) * r1)
    alfa = float((v * y).sum())
    y.sub_((alfa / beta) * r2)
    r1.copy_(r2)
    r2.copy_(y)
    oldb = beta
    beta = math.sqrt(float((y * y).sum()))
    tnorm2 += alfa ** 2 + oldb ** 2 + beta ** 2
    istop = 0
    if itn == 1 and beta / beta1 <= 10 * eps:
        istop = -1
    oldeps = epsln
    delta = cs * dbar + sn * alfa
    gbar = sn * dbar - cs * alfa
    epsln = sn * beta
    dbar = -cs * beta
    root = math.hypot(gbar, dbar)
    gamma = max(math.hypot(gbar, beta), eps)
    cs = gbar / gamma
    sn = beta / gamma
    phi = cs * phibar
    phibar = sn * phibar
    w1 = w2.clone()
    w2.copy_(w)
    w.copy_((v - oldeps * w1 - delta * w2) / gamma)
    x.add_(phi * w)
    v.copy_(r2 / beta)
    gmax, gmin = max(gmax, gamma), min(gmin, gamma)
    z = rhs1 / gamma
    rhs1 = rhs2 - delta * z
    rhs2 = -epsln * z
    Anorm = math.sqrt(tnorm2)
    ynorm = math.sqrt(float((x * x).sum()))
    epsx = Anorm * ynorm * eps
    test1 = float('inf') if ynorm == 0 or Anorm == 0 else phibar / (Anorm * ynorm)
    test2 = float('inf') if Anorm == 0 else root / Anorm
    if istop == 0:
        if 1 + test2 <= 1:
            istop = 2
        if 1 + test1 <= 1:
            istop = 1
        if itn >= maxiter:
            istop = 6
        if gmax / gmin >= 0.1 / eps:
            istop = 4
        if epsx >= beta1:
            istop = 3
        if test2 <= rtol:
            istop = 2
        if test1 <= rtol:
            istop = 1
    for k, val in enumerate((istop, itn, oldb, beta, dbar, epsln, phibar, rhs1, rhs2, tnorm2, gmax, gmin, cs, sn)):
        st[k] = val


def lanczos_update(w, v_prev, v_cur, v_next, alphas, betas, j, tol, stop):
    if stop.item() != 0.0:
        return v_next
    if v_prev is not None:
        w.sub_(betas[j - 1] * v_prev)
    a = (w * v_cur).sum()
    w.sub_(a * v_cur)
    bj = (w * w).sum().sqrt()
    alphas[j - 1] = a
    betas[j] = bj
    if bj.item() < tol:
        stop[0] = float(j)
    v_next.copy_(w / bj)
    return v_next


def lanczos_finish(V, alphas, betas, stop, m_run, x0):
    m = int(stop.item()) if stop.item() != 0.0 else m_run
    Tm = torch.diag(alphas[:m])
    if m > 1:
        Tm = Tm + torch.diag(betas[1:m], diagonal=1) + torch.diag(betas[1:m], diagonal=-1)
    rhs = torch.zeros(m, dtype=F64)
    rhs[0] = betas[0]
    return x0 + V[:m].T @ torch.linalg.solve(Tm, rhs)


NAMES = ("gram_supported env_step_left env_step_right predict loss batch_mean kr_rows hess_left gram gram_scale gram_expand rhs jvp "
         "conv_feature gemm_rows lcontract env_step_left_dense patch_contract gemm_tn chol_solve_dense matvec_fused_supported matvec solve_rhs chol_solve_gram qr dot axpy cg_update lanczos_update lanczos_finish minres_update gl rows_dot rows_matvec_supported rows_matvec rows_rhs lhl rhs_supported rhs_feature bond_supported bond_jv bond_rhs bond_matvec bond_z_supported bond_z lcontract_sum patch_contract_multi lcontract_into filter_supported filter_jv filter_rhs filter_matvec").split()


def install(monkeypatch):
    from tensornetworksfork_amd import ops
    g = globals()
    for n in NAMES:
        monkeypatch.setattr(ops, n, g[n])
