"""Random small conv tensor trains: the imported reference (CPU) against this package's conv engine over the torch
stand-ins for the kernels -- forward, A / b of every train node, one ridge_cholesky sweep and one short Lanczos sweep.
Runs only where the reference is present (the build container; no GPU needed):
    python tests/fuzz_conv_reference.py [cases] [seed] [only this case]
Not collected by pytest; a failing configuration belongs in oracle/gen_golden.py as a fixture."""
import random
import sys
import warnings
import pytest
import torch
sys.path.insert(0, ".")
from oracle.load_reference import load_reference
from tests import _kernel_standins

warnings.filterwarnings("ignore")
ref = load_reference()
_kernel_standins.install(pytest.MonkeyPatch())
from tensornetworksfork_amd.tensor import layers as PL, bregman as PB

NCASES = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ONLY = int(sys.argv[3]) if len(sys.argv) > 3 else None
torch.set_default_dtype(torch.float64)


def rel(a, b):
    return ((a - b).norm() / max(b.norm().item(), 1e-300)).item()


bad = 0
for case in range(NCASES):
    N, r = rnd.randint(2, 4), rnd.randint(1, 5)
    patches, pixels = rnd.choice([3, 6, 10, 17, 20]), rnd.randint(2, 6)
    C, CB = rnd.choice([1, 2, 3]), rnd.choice([-1, -1, 1, 2, 3])
    S = rnd.choice([40, 96])
    kind = rnd.choice(["xe", "square"])
    if ONLY is not None and case != ONLY:
        continue
    print("case", case, dict(N=N, r=r, patches=patches, pixels=pixels, C=C, CB=CB, S=S, loss=kind), flush=True)
    g = torch.Generator().manual_seed(500 + case)
    x = torch.rand(S, patches, pixels, generator=g) * 2 - 1
    if kind == "xe":
        y = torch.nn.functional.one_hot(torch.randint(0, C + 1, (S,), generator=g), C + 1).to(torch.float64)
    else:
        y = torch.randn(S, C, generator=g)
    res = []
    for L, B in ((ref["layers"], ref["bregman"]), (PL, PB)):
        def loss_fn():
            return B.XEAutogradBregman(w=1.0) if kind == "xe" else B.SquareBregFunction()
        out = {}
        try:
            torch.manual_seed(case)
            layer = L.TensorConvolutionTrainLayer(N, r, patches, pixels, output_shape=C, convolution_bond=CB)
            tn = layer.tensor_network
            yp = tn.forward(x, to_tensor=True)
            out["forward"] = yp.clone()
            _, gr, H = loss_fn().forward(yp, y)
            for k, node in enumerate(tn.train_nodes):
                A, b = tn.get_A_b(node, gr, H)
                out[f"A{k}"], out[f"b{k}"] = A.reshape(b.numel(), b.numel()).clone(), b.reshape(-1).clone()
            for which in ("chol", "lanczos"):
                torch.manual_seed(case)
                layer2 = L.TensorConvolutionTrainLayer(N, r, patches, pixels, output_shape=C, convolution_bond=CB)
                tn2 = layer2.tensor_network
                losses = []
                if which == "chol":
                    ok = tn2.accumulating_swipe(x, y, loss_fn(), batch_size=32, num_swipes=1, method="ridge_cholesky", eps=1.0,
                                                eps_decay=0.5, loss_callback=lambda NS, n, l: losses.append(l), disable_tqdm=True)
                else:
                    torch.manual_seed(case + 5)
                    ok = tn2.lanczos_swipe(x, y, loss_fn(), batch_size=32, num_swipes=1, max_iter=2, tol=1e-6,
                                           loss_callback=lambda l: losses.append(l))
                out[which + " ok"] = torch.tensor(float(bool(ok)))
                out[which + " losses"] = torch.tensor(losses)
                out[which + " forward"] = tn2.forward(x, to_tensor=True).clone()
            res.append((None, out))
        except Exception as e:      # noqa: BLE001 - a development tool reports whatever the model raises
            res.append((e, out))
    (er, a), (ep, b) = res
    if er is not None or ep is not None:
        same = type(er) is type(ep)
        print("   RAISED ref %s: %s | prd %s: %s%s" % (type(er).__name__, str(er)[:90], type(ep).__name__, str(ep)[:90],
                                                        "" if same else "   <-- MISMATCH"), flush=True)
        bad += not same
        continue
    for which in ("chol", "lanczos"):
        # a sweep that blows up amplifies rounding without bound: its losses up to that point are compared, not its end
        la = a[which + " losses"]
        upto = next((i for i, v in enumerate(la.tolist()) if not (abs(v) < 100 * abs(la[0].item()))), None)
        if upto is not None:
            print("   %s sweep diverges in the reference at update %d" % (which, upto))
            for d_ in (a, b):
                d_[which + " losses"] = d_[which + " losses"][:upto]
                del d_[which + " forward"]
    worst = max((rel(b[k].to(torch.float64), a[k].to(torch.float64)), k) for k in a if a[k].shape == b[k].shape)
    shapes = [k for k in a if a[k].shape != b[k].shape]
    okc = worst[0] < 1e-6 and not shapes
    print("   worst rel err %.2e at %s%s%s" % (worst[0], worst[1], " shapes differ: %s" % shapes if shapes else "", "" if okc else "   <-- MISMATCH"),
          flush=True)
    bad += not okc
print("done:", bad, "bad of", NCASES)
