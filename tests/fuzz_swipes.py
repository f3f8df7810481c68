"""Random tensor-train models through accumulating_swipe on the GPU against the CPU oracle's sweep: losses after
every update and the final cores.  Ridge values are kept large enough for the local problems to be well
conditioned and bonds are constricted to what the features span (a bond wider than that leaves the QR gauge of a
rank-deficient core, and with it every later ridge step, to rounding: the reference and its own restatement then
differ as well), so that both sides are determined to rounding.
    python tests/fuzz_swipes.py [cases] [seed] [only this case]
Not collected by pytest (a development run, it uses the oracle as its checker); a failing configuration belongs in
tests/test_api_gpu.py."""
import random
import sys
import torch
sys.path.insert(0, ".")
from oracle import tt_oracle as O
from tensornetworksfork_amd.tensor.bregman import SquareBregFunction, XEAutogradBregman
from tensornetworksfork_amd.tensor.layers import TensorTrainLayer

NCASES = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ONLY = int(sys.argv[3]) if len(sys.argv) > 3 else None      # run this case alone (same random draws as in the full run)
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
if torch.cuda.is_available():
    DEV = torch.device("cuda:0")
else:       # no GPU: the host logic over the torch stand-ins for the kernels
    import pytest
    from tests import _kernel_standins
    DEV = torch.device("cpu")
    _kernel_standins.install(pytest.MonkeyPatch())
torch.set_default_dtype(torch.float64)
bad = 0
for case in range(NCASES):
    N, r, f = rnd.randint(2, 6), rnd.randint(1, 9), rnd.randint(1, 6)
    C = rnd.choice([1, 1, 2, 3])
    S = rnd.choice([200, 1025, 3000])
    loss = rnd.choice(["square", "square", "xe"])
    cfg = dict(batch_size=rnd.choice([-1, 64, 300, S]), num_swipes=rnd.randint(1, 2), lr=rnd.choice([1.0, 0.5]),
               method=rnd.choice(["ridge_exact", "ridge_cholesky", "ridge_cholesky", "gradient"]),
               eps=rnd.choice([1e-2, 0.1, 1.0]), orthonormalize=rnd.random() < 0.5, skip_second=rnd.random() < 0.3,
               direction=rnd.choice(["l2r", "r2l"]))
    if cfg["method"] == "gradient":
        cfg["lr"] = 0.05
        cfg["orthonormalize"] = False
    if rnd.random() < 0.2:
        cfg["eps"] = [cfg["eps"] * 0.5 ** k for k in range(2 * cfg["num_swipes"])]
    if rnd.random() < 0.2:
        cfg.update(adaptive_step=True, max_norm=0.7)
    # one tensor for every site makes the environments span symmetric tensors only: the ridge step then returns
    # rank-deficient cores, whose QR gauge is left to rounding (reference and oracle differ there as well)
    shared = rnd.random() < 0.3 and not cfg["orthonormalize"]
    g = torch.Generator().manual_seed(1000 + case)
    if shared:
        xs = [torch.rand(S, f, generator=g) * 2 - 1] * N
    else:
        xs = [torch.rand(S, f, generator=g) * 2 - 1 for _ in range(N)]
    if loss == "xe":
        lab = torch.randint(0, C + 1, (S,), generator=g)
        y = torch.nn.functional.one_hot(lab, C + 1).to(torch.float64)
        bf, kw = XEAutogradBregman(w=1.3), {"w": 1.3}
    else:
        y = torch.randn(S, C, generator=g)
        bf, kw = SquareBregFunction(), None
    layer = TensorTrainLayer(N, r, f, output_shape=C, constrict_bond=True, seed=case)
    cores0 = [n.tensor.clone() for n in layer.tensor_network.main_nodes]
    layer = layer.to(DEV)
    solver = rnd.choice(["normal equations"] * 3 + ["lanczos", "cg", "minres"])
    if C > 1 or S < 1000 or shared:
        # the matrix-free sweeps carry no ridge: where the local operator is singular (several outputs, few samples,
        # one tensor for all sites) a Krylov iteration past its rank returns rounding noise, in SciPy as here
        solver = "normal equations"
    kcfg = dict(batch_size=cfg["batch_size"] if cfg["batch_size"] > 0 else S, num_swipes=cfg["num_swipes"], lr=cfg["lr"],
                max_iter=rnd.randint(2, 6), tol=1e-6)
    if solver != "normal equations":
        cfg = kcfg
    if ONLY is not None and case != ONLY:
        continue
    print("case", case, dict(N=N, r=r, f=f, C=C, S=S, loss=loss, shared=shared, solver=solver), cfg, flush=True)
    losses = []
    tn = layer.tensor_network
    xd, yd = [t.to(DEV) for t in xs], y.to(DEV)
    tol_l, tol_p = 1e-7, 1e-6
    try:
        if solver == "normal equations":
            ok = tn.accumulating_swipe(xd, yd, bf, disable_tqdm=True, verbose=0,
                                       loss_callback=lambda NS, node, l: losses.append(l), **cfg)
        elif solver == "lanczos":
            torch.manual_seed(case)
            ok = tn.lanczos_swipe(xd, yd, bf, loss_callback=lambda l: losses.append(l), **cfg)
            tol_l, tol_p = 1e-5, 1e-5
        else:
            import scipy.sparse.linalg as spla
            ok = tn.scipy_swipe(xd, yd, bf, getattr(spla, solver), loss_callback=lambda l: losses.append(l), **cfg)
            tol_l, tol_p = 1e-4, 1e-4      # the reference's float32 boundary around SciPy
    except Exception as e:      # noqa: BLE001 - a development tool reports whatever the sweep raises
        raised = e
    else:
        raised = None
    trace = []
    try:
        if solver == "normal equations":
            ok_ref, cores_ref = O.accumulating_swipe(cores0, xs, y, loss_fn=loss, loss_kwargs=kw, trace=trace, **cfg)
        elif solver == "lanczos":
            torch.manual_seed(case)
            ok_ref, cores_ref = O.lanczos_swipe(cores0, xs, y, loss_fn=loss, loss_kwargs=kw, trace=trace, **cfg)
        else:
            ok_ref, cores_ref = O.scipy_swipe(cores0, xs, y, getattr(spla, solver), loss_fn=loss, loss_kwargs=kw, trace=trace, **cfg)
        raised_ref = None
    except Exception as e:      # noqa: BLE001
        raised_ref = e
    if raised is not None or raised_ref is not None:
        same = type(raised) is type(raised_ref) and len(losses) == len(trace) + (solver != "normal equations")
        print("   RAISED %s after %d updates / oracle %s after %d%s"
              % (type(raised).__name__, len(losses), type(raised_ref).__name__, len(trace), "" if same else "   <-- MISMATCH"))
        bad += not same
        continue
    ref_l = [t[2] for t in trace]
    upto = next((i for i, v in enumerate(ref_l) if not (abs(v) < 100 * abs(ref_l[0]))), None)
    if upto is not None:
        # a sweep that blows up amplifies rounding without bound (where a factorisation finally fails included);
        # the losses up to that point must still agree
        worst = max([abs(a - b) / abs(b) for a, b in zip(losses[:upto], ref_l[:upto])] + [0.0])
        okd = worst < 1e-6 and len(losses) >= upto
        print("   diverges (in the oracle too) at update %d; losses before it agree to %.2e%s"
              % (upto, worst, "" if okd else "   <-- MISMATCH"), flush=True)
        bad += not okd
        if not okd:
            print("      losses", ["%.4e" % v for v in losses])
            print("      oracle", ["%.4e" % v for v in ref_l])
        continue
    if ok != ok_ref or len(losses) != len(trace):
        print("   FLOW differs:", ok, ok_ref, len(losses), len(trace))
        print("      losses", ["%.4e" % v for v in losses])
        print("      oracle", ["%.4e" % v for v in ref_l])
        bad += 1
        continue
    worst = max([abs(a - b) / max(abs(b), 1e-300) for a, b in zip(losses, ref_l)] + [0.0])
    # predictions, not cores: the QR gauge (signs) is free
    with torch.inference_mode():
        pred = layer([t.to(DEV) for t in xs]).cpu()
    pref = O.predict(cores_ref, xs).reshape(pred.shape)
    perr = ((pred - pref).norm() / max(pref.norm().item(), 1e-300)).item()
    flag = "" if worst < tol_l and perr < tol_p else "   <-- MISMATCH"
    if flag:
        bad += 1
    print("   loss rel err %.2e, prediction rel err %.2e%s" % (worst, perr, flag), flush=True)
print("done:", bad, "bad of", NCASES)
