"""Scratch (for rocprofv3 --kernel-trace --stats): sweeps whose kernel lists show which arithmetic runs where.
    python tools/prof_sweeps.py ridge_exact | exact | lanczos | mnist | mnist_cb4 | cifar
ridge_exact / exact / lanczos: tensor-train model at a reduced headline shape (8 sites, bond 16, degree 5, 20 000 samples)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from tensornetworksfork_amd.tensor.bregman import SquareBregFunction
from tensornetworksfork_amd.tensor.layers import TensorTrainLayer
from tensornetworksfork_amd.models.tnml import polynomial_basis
which = sys.argv[1]
torch.set_default_dtype(torch.float64)
dev = torch.device("cuda:0")
if which in ("ridge_exact", "exact", "lanczos"):
    rng = np.random.RandomState(0)
    X = torch.tensor(rng.uniform(-1, 1, (20000, 8)), device=dev)
    y = torch.sin(X).sum(1, keepdim=True)
    F = polynomial_basis(X, 5)
    layer = TensorTrainLayer(8, 16, 6, output_shape=1, constrict_bond=True, seed=42).to(dev)
    tn = layer.tensor_network
    for rep in range(2):
        if which == "lanczos":
            torch.manual_seed(0)
            tn.lanczos_swipe(F, y, SquareBregFunction(), batch_size=512, max_iter=30, tol=1e-9)
        else:
            tn.accumulating_swipe(F, y, SquareBregFunction(), batch_size=512, method=which, eps=1.0 if which == "ridge_exact" else 0.0,
                                  num_swipes=1)
    torch.cuda.synchronize()
    print(which, "done")
else:
    cfgs = {"mnist": (54000, 50, 17, 24, torch.float64, -1), "mnist_cb4": (54000, 50, 17, 10, torch.float64, 4),
            "cifar": (50000, 65, 49, 48, torch.float32, -1)}
    S, P, PX, r, dt, cb = cfgs[which]
    layer, x, y, bf = bench._conv_problem(dev, S, P, PX, r, dt, cb)
    tn = layer.tensor_network
    for rep in range(2):
        torch.manual_seed(1)
        tn.lanczos_swipe(x, y, bf, batch_size=512, num_swipes=1, max_iter=12)
    torch.cuda.synchronize()
    print(which, "done")
