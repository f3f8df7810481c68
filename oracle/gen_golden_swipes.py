"""Fixtures for random accumulating_swipe calls on tensor-train layers, generated from the imported reference (CPU):
    python oracle/gen_golden_swipes.py
writes tests/golden/swipes_random.npz -- per case the inputs, the layer and sweep keywords as a JSON string
(node_order as indices into train_nodes, a convergence_criterion that stops after `stop_after` calls), and what the
reference produces: return value, losses passed to loss_callback, (NS, node name) of every block_callback, the
prediction after the sweep.  Test infrastructure only; runs in the build container, where /root/reference is present."""
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
NCASES, SEED = 24, 4048


def draw(rnd):
    N, r, f = rnd.randint(2, 5), rnd.randint(1, 6), rnd.randint(2, 5)
    C = rnd.choice([1, 1, 2, 3])
    S = rnd.choice([200, 700])
    kind = rnd.choice(["square", "xe"])
    kw = dict(batch_size=rnd.choice([-1, 64, 300]), num_swipes=rnd.randint(1, 3), lr=rnd.choice([1.0, 0.5]),
              method=rnd.choice(["ridge_exact", "ridge_cholesky", "ridge_cholesky"]), eps=rnd.choice([0.5, 1.0]),
              orthonormalize=rnd.random() < 0.5, skip_second=rnd.random() < 0.3, direction=rnd.choice(["l2r", "r2l"]))
    if rnd.random() < 0.3:
        kw["eps_decay"] = 0.5
    if rnd.random() < 0.25:
        kw.update(eps_per_node=True, eps=[rnd.choice([0.5, 1.0, 2.0]) for _ in range(N)])
    if rnd.random() < 0.25:
        kw.update(adaptive_step=True, max_norm=0.7, min_norm=rnd.choice([None, 0.05]))
    order = None
    if rnd.random() < 0.3:
        order = rnd.sample(range(N), rnd.randint(1, N))
        if rnd.random() < 0.5:
            order = [order, list(reversed(order))]           # a pair: forward and backward order
            kw["_order_is_pair"] = True
    stop_after = rnd.choice([None, None, rnd.randint(1, 4)])
    return N, r, f, C, S, kind, kw, order, stop_after, rnd.random() < 0.5


def main():
    ref = load_reference()
    torch.set_default_dtype(torch.float64)
    rnd = random.Random(SEED)
    out, kept, case = {}, 0, 0
    while kept < NCASES:
        N, r, f, C, S, kind, kw, order, stop_after, list_input = draw(rnd)
        case += 1
        g = torch.Generator().manual_seed(700 + case)
        xs = [torch.rand(S, f, generator=g) * 2 - 1 for _ in range(N)]
        x = xs if list_input else xs[0]
        if kind == "xe":
            y = torch.nn.functional.one_hot(torch.randint(0, C + 1, (S,), generator=g), C + 1).to(torch.float64)
        else:
            y = torch.randn(S, C, generator=g)
        layer = ref["layers"].TensorTrainLayer(N, r, f, output_shape=C, constrict_bond=True, seed=case)
        tn = layer.tensor_network
        kw2 = {k: v for k, v in kw.items() if not k.startswith("_")}
        if order is not None:
            pick = lambda idx: [tn.train_nodes[i] for i in idx]      # noqa: E731
            kw2["node_order"] = (pick(order[0]), pick(order[1])) if kw.get("_order_is_pair") else pick(order)
        losses, blocks, calls = [], [], [0]

        def crit():
            calls[0] += 1
            return stop_after is not None and calls[0] >= stop_after
        bf = ref["bregman"].XEAutogradBregman(w=1.0) if kind == "xe" else ref["bregman"].SquareBregFunction()
        try:
            ok = tn.accumulating_swipe(x, y, bf, disable_tqdm=True, verbose=0, loss_callback=lambda NS, n, l: losses.append(l),
                                       block_callback=lambda NS, n: blocks.append([NS, n.name]), convergence_criterion=crit, **kw2)
        except Exception:       # noqa: BLE001 - calls the reference itself rejects are not fixtures
            continue
        if not losses or any(not (abs(v) < 100 * abs(losses[0])) for v in losses):
            continue            # a diverging sweep amplifies rounding without bound
        pred = tn.forward(x, to_tensor=True)
        p = f"c{kept}_"
        for i, t in enumerate(xs):
            out[p + f"x{i}"] = t.numpy()
        out[p + "y"], out[p + "losses"], out[p + "pred"] = y.numpy(), np.array(losses), pred.numpy()
        out[p + "meta"] = np.array(json.dumps(dict(N=N, r=r, f=f, C=C, loss=kind, layer_seed=case, list_input=list_input, order=order,
                                                   order_is_pair=bool(kw.get("_order_is_pair")), stop_after=stop_after, ok=bool(ok),
                                                   blocks=blocks, kw=kw2 if order is None else {k: v for k, v in kw2.items() if k != "node_order"})))
        kept += 1
    out["ncases"] = np.array(kept)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "swipes_random.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, kept, "cases from", case, "draws")


if __name__ == "__main__":
    main()
