"""Dense and Krylov solvers of the local systems (reference: tensor/network.py:293-327, 796-824, 897-924)."""
import torch

from .. import ops


def _cholesky_solve(A_f, rhs):
    """Cholesky factorisation + two triangular solves; a non-positive pivot raises
    torch.linalg.LinAlgError like torch.linalg.cholesky in the reference."""
    Lc, info = torch.linalg.cholesky_ex(A_f)
    if int(info.item()) != 0:
        raise torch.linalg.LinAlgError(
            f"linalg.cholesky: The factorization could not be completed because the input is not positive-definite "
            f"(the leading minor of order {int(info.item())} is not positive-definite).")
    return torch.cholesky_solve(rhs.unsqueeze(-1), Lc).squeeze(-1)


def lanczos_solve(matvec, b, x0, max_iter, tol, check_every=8, lockstep=False, fused=None):
    """Lanczos-Galerkin without re-orthogonalisation (tensor/network.py:796-824).
    Each iteration is the operator product plus one fused vector kernel that writes the next row of the
    basis; alpha / beta stay on the device.  The reference's exit (beta < tol) is taken inside that
    kernel: it records the iteration and turns every later call into a no-op, which gives the same
    result as leaving the loop there.  The host only looks at the record to stop queueing work: every
    `check_every` iterations it starts a copy of it to pinned memory and reads the copy started one
    period earlier, which has long arrived - no wait.  `lockstep` (sample-sharded runs: every rank
    must queue the same number of products, each holds an all-reduce) reads the record itself at
    those points instead.  The tridiagonal solve and x0 + V y run on the device (ops.lanczos_finish).
    `fused` (ops.LanczosFused): the iterations between two looks at the record are queued by one library call."""
    shape = x0.shape
    D = x0.numel()
    dev = x0.device
    x0 = x0.reshape(-1).contiguous()
    r0 = b.reshape(-1) - matvec(x0.reshape(shape)).reshape(-1)
    st = torch.zeros(2 * max_iter + 2, dtype=x0.dtype, device=dev)
    alphas, betas, stop = st[:max_iter], st[max_iter:2 * max_iter + 1], st[2 * max_iter + 1:]
    beta1 = ops.dot(r0, r0).sqrt()
    betas[0] = beta1
    if max_iter < 1:
        return x0.reshape(shape)
    V = torch.empty((max_iter + 1, D), dtype=x0.dtype, device=dev)     # row j = v_{j+1}
    torch.div(r0, beta1, out=V[0])
    m = 0
    in_flight = None
    j = 1
    while j <= max_iter:
        if fused is not None and x0.dtype == torch.float64:
            count = min(check_every - (j - 1) % check_every, max_iter - j + 1)
            fused.block(V, alphas, betas, stop, j, count, tol)
            j += count - 1
        else:
            w = matvec(V[j - 1].reshape(shape)).reshape(-1)
            if not w.is_contiguous():
                w = w.contiguous()
            ops.lanczos_update(w, V[j - 2] if j > 1 else None, V[j - 1], V[j], alphas, betas, j, tol, stop)
        m = j
        if j % check_every == 0 and j < max_iter:
            if lockstep or not dev.type == 'cuda':
                if stop.item() != 0.0:
                    break
            else:
                if in_flight is not None:
                    host, ready = in_flight
                    ready.synchronize()
                    if host.item() != 0.0:
                        break
                host = torch.empty(1, dtype=x0.dtype, pin_memory=True)
                host.copy_(stop, non_blocking=True)
                ready = torch.cuda.Event()
                ready.record()
                in_flight = (host, ready)
        j += 1
    return ops.lanczos_finish(V, alphas, betas, stop, m, x0).reshape(shape)


def cg_solve(matvec, b, x0, maxiter, rtol, check_every=10):
    """Conjugate gradients with scipy.sparse.linalg.cg's recurrence and stopping
    rule (||r|| < max(0, rtol*||b||) tested before each iteration; SciPy 1.15
    _isolve/iterative.py), in float64 on the device.  The test runs inside the fused
    vector kernel, which freezes x once it holds; the host looks at the flag every
    `check_every` iterations only to stop queueing work."""
    state = torch.zeros(5, dtype=b.dtype, device=b.device)   # [rho, atol, stopped, updates, ||b||^2]
    if x0 is None:
        xk = torch.zeros_like(b)
        r = b.clone()
    else:
        xk = x0.clone()
        r = b - matvec(xk)
    p = r.clone()
    state[0] = ops.dot(r, r)
    state[4] = ops.dot(b, b)
    state[1] = rtol * state[4].sqrt()
    for it in range(maxiter):
        q = matvec(p)
        ops.cg_update(xk, r, p, q if q.is_contiguous() else q.contiguous(), state)
        if (it + 1) % check_every == 0 and it + 1 < maxiter and state[2].item() != 0.0:
            break
    st = state.tolist()     # the one host read of the solve
    if st[4] == 0.0:        # SciPy returns b itself, i.e. zeros, for a zero right-hand side (whatever x0)
        return torch.zeros_like(b), 0
    return xk, (0 if st[2] != 0.0 else maxiter)


def cg_solve_fused(fused, matvec, b, x0, maxiter, rtol, check_every=20):
    """cg_solve on an operator that ops.CGFused takes: every iteration is two launches (the product
    with the direction update in its prologue; the slab reduction with the x / r updates behind it),
    all scalars stay on the device.  On the GPU the iterations run from the shape's ops.CGPlan: blocks of
    `check_every` iterations replayed from a HIP graph captured once per shape."""
    shape = b.shape
    bf = b.reshape(-1)
    plan = None
    if bf.is_cuda and check_every == ops.CGPlan.BLOCK and maxiter > 2:
        plan = ops.CGPlan.get(*fused.keep)
        plan.load(*fused.keep)
        xk, r, pbuf, state = plan.x, plan.r, plan.pbuf, plan.state
        if x0 is None:
            xk.zero_()
            r.copy_(bf)
        else:
            xk.copy_(x0.reshape(-1))
            r.copy_(bf - matvec(xk.reshape(shape)).reshape(-1))
        pbuf.zero_()
        bb = ops.dot(bf, bf)
        state.copy_(torch.stack((torch.zeros_like(bb), ops.dot(r, r), rtol * bb.sqrt(), torch.zeros_like(bb),
                                 torch.zeros_like(bb))))
    else:
        if x0 is None:
            xk = torch.zeros_like(bf)
            r = bf.clone()
        else:
            xk = x0.reshape(-1).clone()
            r = bf - matvec(xk.reshape(shape)).reshape(-1)
        pbuf = torch.zeros(2 * bf.numel(), dtype=b.dtype, device=b.device)
        bb = ops.dot(bf, bf)
        state = torch.stack((torch.zeros_like(bb), ops.dot(r, r), rtol * bb.sqrt(), torch.zeros_like(bb), torch.zeros_like(bb)))
        state = state.contiguous()
    if plan is None:
        for it in range(maxiter):
            fused.iterate(xk, r, pbuf, state, it)
            if (it + 1) % check_every == 0 and it + 1 < maxiter and state[3].item() != 0.0:
                break
    else:
        # the host reads the stop flag every check_every iterations, as above; between the reads whole blocks
        it = min(2, maxiter)
        for k in range(it):
            plan.fused.iterate(xk, r, pbuf, state, k)
        while it < maxiter:
            nxt = min(maxiter, (it // check_every + 1) * check_every)   # up to the next multiple of check_every
            plan.run(it, nxt - it)
            it = nxt
            if it < maxiter and state[3].item() != 0.0:
                break
    st = state.tolist() + [bb.item()]
    if st[5] == 0.0:
        return torch.zeros_like(b), 0
    return xk.reshape(shape).clone() if plan is not None else xk.reshape(shape), (0 if st[3] != 0.0 else maxiter)


def minres_solve(matvec, b, x0, maxiter, rtol, check_every=10):
    """MINRES (Paige & Saunders) following scipy.sparse.linalg.minres (SciPy 1.15
    _isolve/minres.py) without shift or preconditioner, in float64 on the device.  Every scalar of
    the recurrence stays in a device state vector that one fused kernel per iteration updates next
    to the vectors; the stopping tests run inside it and freeze x once one holds, so the host reads
    the state once before the loop (SciPy's early exits on beta1) and then only every
    `check_every` iterations, to stop queueing work."""
    shape = b.shape
    b = b.reshape(-1)
    eps = float(torch.finfo(b.dtype).eps)
    xk = torch.zeros_like(b) if x0 is None else x0.reshape(-1).clone()
    r1 = b.clone() if x0 is None else b - matvec(xk.reshape(shape)).reshape(-1)
    head = torch.stack((ops.dot(r1, r1), ops.dot(b, b))).tolist()     # one host read
    beta1sq, bnorm2 = head
    if beta1sq == 0:
        return xk.reshape(shape), 0
    if bnorm2 == 0:
        return b.clone().reshape(shape), 0
    beta1 = beta1sq ** 0.5
    st = torch.zeros(18, dtype=b.dtype)
    # [istop, itn, oldb, beta, dbar, epsln, phibar, rhs1, rhs2, tnorm2, gmax, gmin, cs, sn, beta1, rtol, maxiter, eps]
    st[3], st[6], st[7], st[14] = beta1, beta1, beta1, beta1
    st[11] = float(torch.finfo(b.dtype).max)
    st[12], st[15], st[16], st[17] = -1.0, rtol, float(maxiter), eps
    st = st.to(b.device)
    r2 = r1.clone()
    v = r1 / beta1
    w = torch.zeros_like(b)
    w2 = torch.zeros_like(b)
    for it in range(maxiter):
        y = matvec(v.reshape(shape)).reshape(-1)
        ops.minres_update(y if y.is_contiguous() else y.contiguous(), r1, r2, v, w, w2, xk, st)
        if (it + 1) % check_every == 0 and it + 1 < maxiter and st[0].item() != 0.0:
            break
    istop = st[0].item()
    return xk.reshape(shape), (maxiter if istop == 6.0 else 0)


def _scipy_kind(solver):
    mod = getattr(solver, "__module__", "") or ""
    name = getattr(solver, "__name__", "")
    if mod.startswith("scipy.sparse.linalg") and name in ("cg", "minres"):
        return name
    return None


def _host_solver(solver, mv, b, x0, maxiter, rtol):
    """Unknown solver callables get a SciPy LinearOperator whose mat-vec runs on the
    device; vectors cross the host boundary as float32 like the reference
    (tensor/network.py:897,918,921-924)."""
    from scipy.sparse.linalg import LinearOperator
    shape, dt, dev = b.shape, b.dtype, b.device

    def host_mv(v):
        return mv(torch.tensor(v, dtype=dt, device=dev).reshape(shape).contiguous()).flatten().float().cpu().numpy()

    b_np = b.flatten().float().cpu().numpy()
    op = LinearOperator((b_np.shape[0], b_np.shape[0]), matvec=host_mv)
    x_sol, _ = solver(op, b_np, x0=None if x0 is None else x0.flatten().float().cpu().numpy(), maxiter=maxiter, rtol=rtol)
    return torch.tensor(x_sol, dtype=dt, device=dev).reshape(shape)
