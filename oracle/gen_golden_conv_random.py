"""Fixtures for random small conv tensor trains, generated from the imported reference (CPU):
    python oracle/gen_golden_conv_random.py
writes tests/golden/conv_random.npz -- per case the inputs and the layer keywords (JSON), and what the reference
produces: the forward pass, b and A v (v seeded) of every train node, the losses of one ridge_cholesky sweep and of
one two-iteration Lanczos sweep with the prediction after each.  Test infrastructure only; runs in the build
container, where /root/reference is present."""
import json
import os
import random
import sys
import warnings
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.load_reference import load_reference

warnings.filterwarnings("ignore")
NCASES, SEED = 16, 777


def main():
    ref = load_reference()
    L, B = ref["layers"], ref["bregman"]
    torch.set_default_dtype(torch.float64)
    rnd = random.Random(SEED)
    out, kept, case = {}, 0, 0
    while kept < NCASES:
        N, r = rnd.randint(2, 4), rnd.randint(1, 5)
        patches, pixels = rnd.choice([3, 6, 10, 17, 20]), rnd.randint(2, 6)
        C, CB = rnd.choice([1, 2, 3]), rnd.choice([-1, -1, 1, 2, 3])
        S, kind = rnd.choice([40, 96]), rnd.choice(["xe", "square"])
        case += 1
        g = torch.Generator().manual_seed(500 + case)
        x = torch.rand(S, patches, pixels, generator=g) * 2 - 1
        if kind == "xe":
            y = torch.nn.functional.one_hot(torch.randint(0, C + 1, (S,), generator=g), C + 1).to(torch.float64)
        else:
            y = torch.randn(S, C, generator=g)

        def loss_fn():
            return B.XEAutogradBregman(w=1.0) if kind == "xe" else B.SquareBregFunction()
        p = f"c{kept}_"
        rec = {}
        try:
            torch.manual_seed(case)
            tn = L.TensorConvolutionTrainLayer(N, r, patches, pixels, output_shape=C, convolution_bond=CB).tensor_network
            yp = tn.forward(x, to_tensor=True)
            rec["forward"] = yp.numpy().copy()
            _, gr, H = loss_fn().forward(yp, y)
            vg = torch.Generator().manual_seed(900 + case)
            for k, node in enumerate(tn.train_nodes):
                A, b = tn.get_A_b(node, gr, H)
                v = torch.randn(b.numel(), generator=vg)
                rec[f"b{k}"], rec[f"v{k}"] = b.reshape(-1).numpy().copy(), v.numpy()
                rec[f"Av{k}"] = (A.reshape(b.numel(), b.numel()) @ v).numpy()
            good = True
            for which in ("chol", "lanczos"):
                torch.manual_seed(case)
                tn2 = L.TensorConvolutionTrainLayer(N, r, patches, pixels, output_shape=C, convolution_bond=CB).tensor_network
                losses = []
                if which == "chol":
                    ok = tn2.accumulating_swipe(x, y, loss_fn(), batch_size=32, num_swipes=1, method="ridge_cholesky", eps=1.0,
                                                eps_decay=0.5, loss_callback=lambda NS, n, l: losses.append(l), disable_tqdm=True)
                else:
                    torch.manual_seed(case + 5)
                    ok = tn2.lanczos_swipe(x, y, loss_fn(), batch_size=32, num_swipes=1, max_iter=2, tol=1e-6,
                                           loss_callback=lambda l: losses.append(l))
                if not ok or any(not (abs(v_) < 100 * abs(losses[0])) for v_ in losses):
                    good = False        # a diverging sweep amplifies rounding without bound
                    break
                rec[which + "_losses"] = np.array(losses)
                rec[which + "_forward"] = tn2.forward(x, to_tensor=True).numpy().copy()
        except Exception:       # noqa: BLE001 - configurations the reference itself rejects are not fixtures
            continue
        if not good:
            continue
        for k_, v_ in rec.items():
            out[p + k_] = v_
        out[p + "x"], out[p + "y"] = x.numpy(), y.numpy()
        out[p + "meta"] = np.array(json.dumps(dict(N=N, r=r, patches=patches, pixels=pixels, C=C, CB=CB, loss=kind, torch_seed=case,
                                                   nodes=len(tn.train_nodes))))
        kept += 1
    out["ncases"] = np.array(kept)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "conv_random.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, kept, "cases from", case, "draws")


if __name__ == "__main__":
    main()
