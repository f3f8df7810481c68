"""Random estimator configurations: the imported reference (CPU) against this package's estimators over the torch
stand-ins for the kernels, i.e. the host logic of fit / early stopping / model types / prediction.  Runs only where
the reference is present (the build container; no GPU needed):
    python tests/fuzz_estimators.py [cases] [seed] [only this case]
Not collected by pytest; a failing configuration belongs in oracle/gen_golden.py as a fixture."""
import random
import sys
import warnings
import numpy as np
import pytest
import torch
sys.path.insert(0, ".")
from oracle.load_reference import load_reference
from tests import _kernel_standins

warnings.filterwarnings("ignore")
ref = load_reference()
_kernel_standins.install(pytest.MonkeyPatch())
from tensornetworksfork_amd.models.tensor_train import TensorTrainRegressor
from tensornetworksfork_amd.models.tnml import TNMLRegressor
from tensornetworksfork_amd.tensor.bregman import XEAutogradBregman

NCASES = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ONLY = int(sys.argv[3]) if len(sys.argv) > 3 else None
torch.set_default_dtype(torch.float64)
bad = 0
for case in range(NCASES):
    which = rnd.choice(["tt", "tt", "tnml"])
    S, F = rnd.choice([120, 300, 700]), rnd.randint(2, 5)
    task = rnd.choice(["regression", "regression", "classification"]) if which == "tt" else "regression"
    kw = dict(r=rnd.randint(2, 6), seed=rnd.randint(0, 99), lr=rnd.choice([1.0, 0.5]), eps_start=rnd.choice([1.0, 0.1]),
              eps_decay=rnd.choice([0.5, 0.8]), batch_size=rnd.choice([64, 512]), method=rnd.choice(["ridge_cholesky", "ridge_exact"]),
              num_swipes=rnd.randint(1, 4), early_stopping=rnd.choice([0, 0, 2]), model_type=rnd.choice(["tt", "tt", "tt_type1", "cpd", "cpd_type1"]))
    if which == "tt":
        kw.update(N=rnd.randint(2, 4), constrict_bond=rnd.random() < 0.5, perturb=rnd.random() < 0.3,
                  cum_sum=rnd.random() < 0.15, train_operator=rnd.random() < 0.2)
        if rnd.random() < 0.25:
            kw.update(linear_dim=rnd.randint(2, 4))
    else:
        kw.update(basis=rnd.choice(["sin-cos", "polynomial"]), degree=rnd.randint(1, 3), train_operator=rnd.random() < 0.2)
    if ONLY is not None and case != ONLY:
        continue
    rng = np.random.RandomState(case)
    X = rng.uniform(-1, 1, (S, F))
    if task == "regression":
        y = np.sin(X).sum(1, keepdims=True) + 0.05 * rng.randn(S, 1)
        bf_r = bf_p = None
    else:
        lab = (X[:, 0] + X[:, 1] > 0).astype(int) + (X[:, 0] - X[:, 1] > 0.5).astype(int)
        y = np.eye(3)[lab]
        bf_r, bf_p = ref["bregman"].XEAutogradBregman(w=1.0), XEAutogradBregman(w=1.0)
        kw["output_dim"] = 2
    Xt = rng.uniform(-1, 1, (40, F))
    print("case", case, which, task, dict(S=S, F=F), kw, flush=True)
    out = []
    for name, cls, bf in (("ref", ref["tensor_train"].TensorTrainRegressor if which == "tt" else ref["tnml"].TNMLRegressor, bf_r),
                          ("prd", TensorTrainRegressor if which == "tt" else TNMLRegressor, bf_p)):
        try:
            extra = dict(task=task) if which == "tt" else {}
            torch.manual_seed(1000 + case)      # layers that take `seed` without using it draw from the global generator
            est = cls(device="cpu", bf=bf, **extra, **kw)
            est.fit(X.copy(), y.copy())
            vh = est._early_stopper.val_history
            out.append((None, np.asarray(est.predict(Xt)), np.array([vh[k] for k in sorted(vh)])))
        except Exception as e:      # noqa: BLE001 - a development tool reports whatever fit raises
            out.append((e, None, None))
    (er, pr, vr), (ep, pp, vp) = out
    if er is not None or ep is not None:
        same = type(er) is type(ep)
        print("   RAISED ref %s: %s | prd %s: %s%s" % (type(er).__name__, str(er)[:80], type(ep).__name__, str(ep)[:80],
                                                        "" if same else "   <-- MISMATCH"), flush=True)
        bad += not same
        continue
    if vr.shape != vp.shape:
        print("   val_history lengths differ", vr.shape, vp.shape, "  <-- MISMATCH", flush=True)
        bad += 1
        continue
    ev = float(np.max(np.abs(vr - vp) / np.maximum(np.abs(vr), 1e-300))) if vr.size else 0.0
    epd = float(np.linalg.norm(pr - pp) / max(np.linalg.norm(pr), 1e-300))
    okc = ev < 1e-6 and epd < 1e-5
    print("   val_history rel err %.2e (%d entries), prediction rel err %.2e%s" % (ev, vr.size, epd, "" if okc else "   <-- MISMATCH"),
          flush=True)
    bad += not okc
print("done:", bad, "bad of", NCASES)
