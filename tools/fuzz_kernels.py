"""Random shapes through the library's operators, each result against torch.einsum in float64 on the device.
A development tool for the dispatch boundaries between kernels (tile counts, LDS limits, ragged ends):
    python tools/fuzz_kernels.py [cases] [seed] [log]
Every case is written to the log before it runs, so a fault names its shape; a failing shape belongs in
tests/test_kernels_gpu.py."""
import random
import sys
import torch
sys.path.insert(0, ".")
from tensornetworksfork_amd import ops, _lib

NCASES = int(sys.argv[1]) if len(sys.argv) > 1 else 200
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 0
LOG = open(sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/fuzz.log", "w")
dev = torch.device("cuda:0")
rnd = random.Random(SEED)
S_SET = [1, 3, 63, 64, 65, 257, 1000, 4097, 20011]
C_SET = [1, 1, 1, 2, 3, 9, 10]
B_SET = [1, 2, 3, 5, 7, 8, 15, 16, 17, 24, 31, 32, 33, 40, 48, 63, 64, 65, 70, 96, 100, 128]
P_SET = [1, 2, 3, 4, 6, 8, 9, 15, 16, 17, 33, 50, 65, 130, 192, 193, 256, 257, 400]
bad = []
unsupported = {}


def say(*a):
    print(*a, file=LOG, flush=True)


def rel(a, b):
    d = (a - b).norm().item()
    n = b.norm().item()
    return d / n if n > 0 else d


def check(name, shape, fn, ref, tol=1e-11):
    say("  ", name)
    try:
        got = fn()
    except (_lib.TTKError, ValueError) as e:
        unsupported.setdefault(name, []).append(shape)
        say("     unsupported:", e)
        return
    torch.cuda.synchronize()
    want = ref()
    e = rel(got.reshape(want.shape), want)
    if not (e < tol):
        bad.append((name, shape, e))
        say("     MISMATCH", e)
        print("MISMATCH", name, shape, e, flush=True)


for case in range(NCASES):
    S, C, rl, p, rr = rnd.choice(S_SET), rnd.choice(C_SET), rnd.choice(B_SET), rnd.choice(P_SET), rnd.choice(B_SET)
    D = rl * p * rr
    if D > 400000 or S * max(rl, rr) * p > 3e8:
        continue
    shape = (S, C, rl, p, rr)
    say("case", case, shape)
    g = torch.Generator(device=dev).manual_seed(case)
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64, device=dev) - 0.5   # noqa: E731
    L, X, R, core, v = r(S, C, rl), r(S, p), r(S, rr), r(rl, p, rr), r(rl, p, rr)
    gr = r(S, C)
    Hs = r(S, C, C)
    H = (Hs @ Hs.transpose(1, 2) + 0.1 * torch.eye(C, dtype=torch.float64, device=dev)).contiguous()
    z = r(S, rl)
    T = torch.einsum('sp,lpr->slr', X, core) if S * rl * rr < 5e7 else None
    if T is not None:
        check("env_step_left", shape, lambda: ops.env_step_left(L, X, core), lambda: torch.einsum('scl,slr->scr', L, T))
        check("env_step_right", shape, lambda: ops.env_step_right(R, X, core), lambda: torch.einsum('sr,slr->sl', R, T))
        o0 = r(S, rl)
        check("env_step_right+out", shape, lambda: ops.env_step_right(R, X, core, out=o0.clone()),
              lambda: o0 + torch.einsum('sr,slr->sl', R, T))
        if ops.DENSE_MIN_P <= p <= ops.DENSE_MAX_P:
            check("env_step_right mm32", shape, lambda: ops.env_step_right(R, X, core, mm32=True),
                  lambda: torch.einsum('sr,slr->sl', R, T), tol=2e-5)
    check("predict", shape, lambda: ops.predict(L, z), lambda: torch.einsum('scl,sl->sc', L, z))
    GLref = torch.einsum('sc,scl->sl', gr, L)
    check("gl", shape, lambda: ops.gl(gr, L), lambda: GLref)
    check("hess_left", shape, lambda: ops.hess_left(H, L), lambda: torch.einsum('scd,scl->sdl', H, L))
    if rl <= 128 and C <= 32 and C * rl <= 4096:
        check("lhl", shape, lambda: ops.lhl(L, H, z),
              lambda: torch.einsum('scl,scd,sd->sl', L, H, torch.einsum('sdl,sl->sd', L, z)))

    def bref():
        W = torch.einsum('sl,sr->slr', GLref, R).reshape(S, rl * rr)
        return (X.T @ W).reshape(p, rl, rr).permute(1, 0, 2)
    if S * rl * rr < 5e7:
        if ops.rhs_supported(S, C, rl, p, rr):
            check("rhs", shape, lambda: ops.rhs(L, X, R, gr), bref)
            b0 = r(rl, p, rr)
            check("rhs+out", shape, lambda: ops.rhs(L, X, R, gr, out=b0.clone()), lambda: b0 + bref())
        if ops.DENSE_MIN_P <= p <= ops.DENSE_MAX_P and rr <= 128 and p + rr <= 280:
            check("rhs_feature", shape, lambda: ops.rhs_feature(L, X, R, gr), bref)
            check("rhs_feature mm32", shape, lambda: ops.rhs_feature(L, X, R, gr, mm32=True), bref, tol=2e-5)
        if T is not None and ops.rhs_supported(S, C, rl, p, rr):
            def mvref():
                Tv = torch.einsum('sp,lpr->slr', X, v)
                u = torch.einsum('scl,slr,sr->sc', L, Tv, R)
                hu = torch.einsum('scd,sd->sc', H, u)
                W = torch.einsum('sc,scl,sr->slr', hu, L, R).reshape(S, rl * rr)
                return (X.T @ W).reshape(p, rl, rr).permute(1, 0, 2)
            check("matvec", shape, lambda: ops.matvec(L, X, R, H, v), mvref)
    if T is not None and S * rl * rr < 5e7:
        # the routes the sweep itself takes for this site (whatever kernels they pick)
        from tensornetworksfork_amd.tensor import engine
        lp = engine.LocalProblem(S, C, (rl, p, rr), v, lambda t: t, terms=[(L, X, R)])
        check("LocalProblem.rhs", shape, lambda: lp.rhs(gr), bref)
        check("LocalProblem.jv", shape, lambda: lp.jv(v),
              lambda: torch.einsum('scl,slr,sr->sc', L, torch.einsum('sp,lpr->slr', X, v), R))

        def mvref2():
            u = torch.einsum('scl,slr,sr->sc', L, torch.einsum('sp,lpr->slr', X, v), R)
            hu = torch.einsum('scd,sd->sc', H, u)
            W = torch.einsum('sc,scl,sr->slr', hu, L, R).reshape(S, rl * rr)
            return (X.T @ W).reshape(p, rl, rr).permute(1, 0, 2)
        check("LocalProblem.matvec", shape, lambda: lp.matvec_fn(H)(v), mvref2)
    if S * C * D <= 3e7:
        Jref = torch.einsum('scl,sp,sr->sclpr', L, X, R).reshape(S, C, D)
        check("kr_rows", shape, lambda: ops.kr_rows(L, X, R), lambda: Jref)
        if D <= 3000 and ops.gram_supported(S, C, rl, p, rr, True):
            Aref = torch.einsum('scx,scd,sdy->xy', Jref, H, Jref)
            check("gram", shape, lambda: ops.gram_expand(ops.gram(L, H, X, R, sym=True)), lambda: Aref)
            # unsymmetric per-sample matrix in place of the Hessian (the other compressed layout)
            check("gram unsym", shape, lambda: ops.gram_expand(ops.gram(L, Hs.contiguous(), X, R, sym=False)),
                  lambda: torch.einsum('scx,scd,sdy->xy', Jref, Hs, Jref))
            # powers of one variable as features: pairs (p,q) with equal p+q share one slice of the Gram
            from tensornetworksfork_amd.tensor.engine import pair_classes
            xv = r(S) * 2
            Xp = torch.stack([xv ** k for k in range(p)], dim=-1).contiguous()
            rep = pair_classes(Xp)
            Jp = torch.einsum('scl,sp,sr->sclpr', L, Xp, R).reshape(S, C, D)
            Ap = torch.einsum('scx,scd,sdy->xy', Jp, H, Jp)
            check("gram classes", shape, lambda: ops.gram_expand(ops.gram(L, H, Xp, R, sym=True, pair_rep=rep)), lambda: Ap, tol=1e-10)
            # ridge Cholesky solve straight from the compressed Gram
            wsg = ops.gram(L, H, X, R, sym=True)
            bb, ww = r(rl, p, rr), r(rl, p, rr)
            sc = ops.gram_scale(wsg)
            ridge = 0.3

            def solved():
                xsol, info = ops.chol_solve_gram(wsg, bb, ww, sc, ridge)
                assert int(info.item()) == 0
                return xsol
            check("chol_solve_gram", shape, solved,
                  lambda: torch.linalg.solve(Aref / sc + ridge * torch.eye(D, dtype=torch.float64, device=dev),
                                             -(bb.reshape(-1) / sc + ridge * ww.reshape(-1))), tol=1e-9)
        if D <= 512:
            vv = v.reshape(D)
            check("rows_dot", shape, lambda: ops.rows_dot(Jref, vv), lambda: torch.einsum('scx,x->sc', Jref, vv))
            if ops.rows_matvec_supported(S, C, D):
                check("rows_matvec", shape, lambda: ops.rows_matvec(Jref, H, vv),
                      lambda: torch.einsum('scx,scd,sd->x', Jref, H, torch.einsum('sdx,x->sd', Jref, vv)))
                check("rows_rhs", shape, lambda: ops.rows_rhs(Jref, gr), lambda: torch.einsum('scx,sc->x', Jref, gr))
    m, n = rl * p, rr
    if m * n <= 200000:
        Aq = r(m, n)
        say("   qr", (m, n))
        try:
            Q, Rq = ops.qr(Aq)
            Qr, Rr = torch.linalg.qr(Aq, mode='reduced')
            for name, got, want in (("qr Q", Q, Qr), ("qr R", Rq, Rr)):
                e = rel(got, want)
                if not e < 1e-10:
                    bad.append((name, (m, n), e))
                    print("MISMATCH", name, (m, n), e, flush=True)
        except (_lib.TTKError, ValueError) as e:
            unsupported.setdefault("qr", []).append((m, n))
            say("     unsupported:", e)
    if case % 20 == 0:
        print("case", case, shape, "mismatches so far", len(bad), flush=True)

print("done:", len(bad), "mismatches")
for b in bad:
    print("  ", b)
for k, v in unsupported.items():
    print("unsupported", k, len(v), v[:6])
