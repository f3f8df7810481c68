"""Fixtures for random estimator configurations, generated from the imported reference (CPU):
    python oracle/gen_golden_estimators.py
writes tests/golden/estimators_random.npz -- per case the inputs (X, y, Xt), the constructor keywords as a JSON
string, and what the reference returns after fit (predictions on Xt, val_history).  Test infrastructure only; runs
in the build container, where /root/reference is present."""
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
NCASES, SEED = 24, 2024


def draw(rnd):
    which = rnd.choice(["tt", "tt", "tnml"])
    S, F = rnd.choice([120, 300]), rnd.randint(2, 5)
    task = rnd.choice(["regression", "regression", "classification"]) if which == "tt" else "regression"
    kw = dict(r=rnd.randint(2, 6), seed=rnd.randint(0, 99), lr=1.0, eps_start=rnd.choice([1.0, 0.5]),
              eps_decay=rnd.choice([0.5, 0.8]), batch_size=rnd.choice([64, 512]), method=rnd.choice(["ridge_cholesky", "ridge_exact"]),
              num_swipes=rnd.randint(1, 3), early_stopping=rnd.choice([0, 0, 2]),
              model_type=rnd.choice(["tt", "tt", "tt_type1", "cpd", "cpd_type1"]))
    if which == "tt":
        kw.update(N=rnd.randint(2, 4), constrict_bond=True, perturb=False, cum_sum=rnd.random() < 0.15,
                  train_operator=rnd.random() < 0.2)
        if rnd.random() < 0.25:
            kw.update(linear_dim=rnd.randint(2, 4))
    else:
        kw.update(basis=rnd.choice(["sin-cos", "polynomial"]), degree=rnd.randint(1, 3), train_operator=rnd.random() < 0.2)
    return which, S, F, task, kw


def main():
    ref = load_reference()
    torch.set_default_dtype(torch.float64)
    rnd = random.Random(SEED)
    out = {}
    kept = 0
    case = 0
    while kept < NCASES:
        which, S, F, task, kw = draw(rnd)
        case += 1
        rng = np.random.RandomState(case)
        X = rng.uniform(-1, 1, (S, F))
        if task == "regression":
            y = np.sin(X).sum(1, keepdims=True) + 0.05 * rng.randn(S, 1)
            bf = None
        else:
            lab = (X[:, 0] + X[:, 1] > 0).astype(int) + (X[:, 0] - X[:, 1] > 0.5).astype(int)
            y = np.eye(3)[lab]
            bf = ref["bregman"].XEAutogradBregman(w=1.0)
            kw["output_dim"] = 2
        Xt = rng.uniform(-1, 1, (40, F))
        cls = ref["tensor_train"].TensorTrainRegressor if which == "tt" else ref["tnml"].TNMLRegressor
        extra = dict(task=task) if which == "tt" else {}
        torch.manual_seed(1000 + case)      # layers that take `seed` without using it draw from the global generator
        try:
            est = cls(device="cpu", bf=bf, **extra, **kw)
            est.fit(X.copy(), y.copy())
        except Exception:       # noqa: BLE001 - configurations the reference itself rejects are not fixtures
            continue
        vh = est._early_stopper.val_history
        vals = np.array([vh[k] for k in sorted(vh)])
        pred = np.asarray(est.predict(Xt))
        if not (np.all(np.isfinite(vals)) and np.all(np.isfinite(pred)) and (vals.size == 0 or vals.max() < 100 * max(vals[0], 1e-12))):
            continue                        # a diverging fit amplifies rounding without bound
        p = f"c{kept}_"
        out[p + "X"], out[p + "y"], out[p + "Xt"], out[p + "pred"], out[p + "val_history"] = X, y, Xt, pred, vals
        out[p + "meta"] = np.array(json.dumps(dict(which=which, task=task, torch_seed=1000 + case, kw=kw)))
        kept += 1
    out["ncases"] = np.array(kept)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "estimators_random.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, kept, "cases from", case, "draws")


if __name__ == "__main__":
    main()
