"""Generate tests/golden/*.npz by running the REFERENCE in the build container.

Run:  python oracle/gen_golden.py          (needs /root/reference; CPU only)

Every fixture stores the exact inputs next to the reference's outputs, so the
tests need neither the reference nor this script at run time.  Only data is
written: arrays of inputs, cores, Gram matrices, right-hand sides, steps,
losses.  No reference source text goes into the repository.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from load_reference import load_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_default_dtype(torch.float64)
ref = load_reference()
layers, bregman, network = ref["layers"], ref["bregman"], ref["network"]


def npz(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def make_loss(kind):
    if kind == "square":
        return bregman.SquareBregFunction()
    if kind == "mse":
        return bregman.AutogradLoss()
    if kind == "xe":
        return bregman.XEAutogradBregman(w=1.0)
    raise ValueError(kind)


def make_data(S, N, p, C, kind, seed, list_input=True):
    g = torch.Generator().manual_seed(seed)
    if list_input:
        x = [torch.rand(S, p, generator=g) * 2 - 1 for _ in range(N)]
    else:
        x = torch.rand(S, p, generator=g) * 2 - 1
    if kind == "xe":
        lab = torch.randint(0, C + 1, (S,), generator=g)
        y = torch.nn.functional.one_hot(lab, C + 1).to(torch.float64)
    else:
        y = torch.randn(S, C, generator=g)
    return x, y


def cores_of(layer):
    return [n.tensor.detach().clone() for n in layer.tensor_network.main_nodes]


def save_x(d, x):
    if isinstance(x, list):
        for i, t in enumerate(x):
            d[f"x{i}"] = t
        d["x_is_list"] = 1
    else:
        d["x"] = x
        d["x_is_list"] = 0


# ---------------------------------------------------------------- 1. core update
def gen_core_update(tag, N, p, r, C, S, kind, constrict, list_input, seed):
    layer = layers.TensorTrainLayer(N, r, p, output_shape=C, constrict_bond=constrict, seed=seed)
    tn = layer.tensor_network
    x, y = make_data(S, N, p, C, kind, seed + 100, list_input)
    d = {"N": N, "p": p, "r": r, "C": C, "S": S, "seed": seed, "constrict": int(constrict), "y": y,
         "loss_kind": np.array(kind)}
    save_x(d, x)
    for i, c in enumerate(cores_of(layer)):
        d[f"core{i}"] = c
    loss_fn = make_loss(kind)
    y_pred = tn.forward(x, to_tensor=True)
    loss, g, H = loss_fn.forward(y_pred, y)
    d["y_pred"], d["loss"], d["g"], d["H"] = y_pred, loss, g, H
    v_gen = torch.Generator().manual_seed(seed + 7)
    for k, node in enumerate(tn.train_nodes):
        A, b = tn.get_A_b(node, g, H)
        d[f"A{k}"], d[f"b{k}"] = A, b
        d[f"b_only{k}"] = tn.get_b(node, g)
        # matrix-free product through get_J exactly as lanczos_swipe does (network.py:781-789)
        v = torch.randn(node.tensor.shape, generator=v_gen)
        pj = tn.get_J(node, g)
        coeff = torch.einsum(f"{pj['einsum']},{pj['node_ein']},{pj['dd_loss_ein']}->{pj['coeff_ein']}", pj['J'].tensor, v, H)
        Av = torch.einsum(f"{pj['einsum']},{pj['d_loss_ein']}->{pj['node_ein']}", pj['J'].tensor, coeff)
        d[f"v{k}"], d[f"Av{k}"] = v, Av
        for method in ["exact", "ridge_exact", "ridge_cholesky", "cholesky"]:
            for ei, eps in enumerate([0.0, 1e-3, 1.0]):
                if eps == 0.0 and method in ("exact", "cholesky", "ridge_cholesky"):
                    continue  # rank-deficient without ridge at these sizes
                if method in ("exact", "cholesky") and ei > 1:
                    continue
                try:
                    step = tn.solve_system(node, A, b, method=method, eps=eps)
                    d[f"step_{method}_{ei}_{k}"] = step
                except torch.linalg.LinAlgError:
                    d[f"fail_{method}_{ei}_{k}"] = 1
        d[f"step_gradient_{k}"] = tn.solve_system(node, A, b, method="gradient")
    d["eps_values"] = np.array([0.0, 1e-3, 1.0])
    npz(f"tt_core_update_{tag}", **d)


# ---------------------------------------------------------------- 2. swipe traces
def gen_swipe(tag, N, p, r, C, S, kind, bs, seed, **kw):
    layer = layers.TensorTrainLayer(N, r, p, output_shape=C, constrict_bond=True, seed=seed)
    tn = layer.tensor_network
    x, y = make_data(S, N, p, C, kind, seed + 100, True)
    d = {"N": N, "p": p, "r": r, "C": C, "S": S, "seed": seed, "bs": bs, "y": y, "loss_kind": np.array(kind)}
    save_x(d, x)
    for i, c in enumerate(cores_of(layer)):
        d[f"core{i}"] = c
    for k_, v_ in kw.items():
        d["kw_" + k_] = np.array(v_)
    losses, visits, snaps = [], [], []

    def loss_cb(NS, node, loss):
        losses.append(loss)
        visits.append((NS, tn.main_nodes.index(node)))

    def block_cb(NS, node):
        snaps.append([n.tensor.detach().clone() for n in tn.main_nodes])

    if kw.get("orthonormalize", False):
        tn.orthonormalize_left()
        for i, c in enumerate(cores_of(layer)):
            d[f"core_after_init_qr{i}"] = c
    ok = tn.accumulating_swipe(x, y, make_loss(kind), batch_size=bs, loss_callback=loss_cb,
                               block_callback=block_cb, **kw)
    d["ok"] = int(ok)
    d["losses"] = np.array(losses)
    d["visits"] = np.array(visits)
    for u, snap in enumerate(snaps):
        for i, c in enumerate(snap):
            d[f"u{u}_core{i}"] = c
    d["num_updates"] = len(snaps)
    npz(f"swipe_trace_{tag}", **d)


# ---------------------------------------------------------------- 3. estimators
def gen_tnml(tag, S, F, r, degree, basis, seed, **kw):
    rng = np.random.RandomState(seed)
    X = rng.uniform(-1, 1, (S, F))
    y = np.sin(X).sum(1, keepdims=True) + 0.01 * rng.randn(S, 1)
    est = ref["tnml"].TNMLRegressor(r=r, output_dim=1, seed=42, device="cpu", basis=basis, degree=degree, **kw)
    est.fit(X, y)
    Xt = rng.uniform(-1, 1, (64, F))
    d = {"X": X, "y": y, "Xt": Xt, "pred": est.predict(Xt), "r": r, "degree": degree, "basis": np.array(basis)}
    for k_, v_ in kw.items():
        d["kw_" + k_] = np.array(v_)
    vh = est._early_stopper.val_history
    d["val_history"] = np.array([vh[k] for k in sorted(vh)])
    for i, n in enumerate(est._model.tensor_network.main_nodes):
        d[f"core{i}"] = n.tensor
    npz(f"tnml_fit_{tag}", **d)


def gen_ttreg(tag, seed, task, **kw):
    rng = np.random.RandomState(seed)
    if task == "regression":
        S, F = 300, 4
        X = rng.uniform(-1, 1, (S, F))
        y = (X ** 2).sum(1, keepdims=True) + X[:, :1] * X[:, 1:2] + 0.01 * rng.randn(S, 1)
        bf = None
    else:
        from sklearn.datasets import load_iris
        data = load_iris()
        X = data.data
        X = (X - X.mean(0)) / X.std(0)
        yl = data.target
        y = np.eye(3)[yl]
        bf = bregman.XEAutogradBregman(w=1.0)
    est = ref["tensor_train"].TensorTrainRegressor(device="cpu", task=task, bf=bf, **kw)
    est.fit(X, y)
    d = {"X": X, "y": y, "pred": est.predict(X[:50]), "task": np.array(task)}
    for k_, v_ in kw.items():
        d["kw_" + k_] = np.array(v_)
    vh = est._early_stopper.val_history
    d["val_history"] = np.array([vh[k] for k in sorted(vh)])
    for i, n in enumerate(est._model.tensor_network.main_nodes):
        d[f"core{i}"] = n.tensor
    npz(f"ttreg_fit_{tag}", **d)


# ---------------------------------------------------------------- 4/5. Krylov swipes
def gen_lanczos(tag, N, p, r, C, S, kind, bs, seed, num_swipes, max_iter, tol):
    layer = layers.TensorTrainLayer(N, r, p, output_shape=C, constrict_bond=True, seed=seed)
    tn = layer.tensor_network
    x, y = make_data(S, N, p, C, kind, seed + 100, True)
    d = {"N": N, "p": p, "r": r, "C": C, "S": S, "seed": seed, "bs": bs, "y": y, "loss_kind": np.array(kind),
         "num_swipes": num_swipes, "max_iter": max_iter, "tol": tol, "torch_seed": seed + 5}
    save_x(d, x)
    for i, c in enumerate(cores_of(layer)):
        d[f"core{i}"] = c
    # x0 draws: one randn_like(b_rhs) per visited node, nothing else uses the generator
    torch.manual_seed(seed + 5)
    order = []
    for NS in range(num_swipes):
        order += list(range(N)) if NS % 2 == 0 else list(reversed(range(N)))
    for u, k in enumerate(order):
        d[f"x0_{u}"] = torch.randn_like(tn.main_nodes[k].tensor)
    snaps, losses = [], []
    torch.manual_seed(seed + 5)
    ok = tn.lanczos_swipe(x, y, make_loss(kind), batch_size=bs, num_swipes=num_swipes, max_iter=max_iter, tol=tol,
                          block_callback=lambda NS, node: snaps.append([n.tensor.detach().clone() for n in tn.main_nodes]),
                          loss_callback=lambda l: losses.append(l))
    d["ok"] = int(ok)
    d["losses"] = np.array(losses)
    d["order"] = np.array(order)
    for u, snap in enumerate(snaps):
        for i, c in enumerate(snap):
            d[f"u{u}_core{i}"] = c
    d["num_updates"] = len(snaps)
    npz(f"lanczos_{tag}", **d)


def gen_scipy(tag, solver_name, N, p, r, C, S, kind, bs, seed, num_swipes, max_iter, tol):
    import scipy.sparse.linalg as spla
    solver = getattr(spla, solver_name)
    layer = layers.TensorTrainLayer(N, r, p, output_shape=C, constrict_bond=True, seed=seed)
    tn = layer.tensor_network
    x, y = make_data(S, N, p, C, kind, seed + 100, True)
    d = {"N": N, "p": p, "r": r, "C": C, "S": S, "seed": seed, "bs": bs, "y": y, "loss_kind": np.array(kind),
         "num_swipes": num_swipes, "max_iter": max_iter, "tol": tol, "solver": np.array(solver_name)}
    save_x(d, x)
    for i, c in enumerate(cores_of(layer)):
        d[f"core{i}"] = c
    snaps, losses = [], []
    ok = tn.scipy_swipe(x, y, make_loss(kind), solver, batch_size=bs, num_swipes=num_swipes, max_iter=max_iter, tol=tol,
                        block_callback=lambda NS, node: snaps.append([n.tensor.detach().clone() for n in tn.main_nodes]),
                        loss_callback=lambda l: losses.append(l))
    d["ok"] = int(ok)
    d["losses"] = np.array(losses)
    for u, snap in enumerate(snaps):
        for i, c in enumerate(snap):
            d[f"u{u}_core{i}"] = c
    d["num_updates"] = len(snaps)
    npz(f"scipy_{solver_name}_{tag}", **d)


# ---------------------------------------------------------------- 8. QR gauge
def gen_qr(tag, N, p, r, C, seed):
    layer = layers.TensorTrainLayer(N, r, p, output_shape=C, constrict_bond=True, seed=seed)
    tn = layer.tensor_network
    d = {"N": N, "p": p, "r": r, "C": C}
    for i, c in enumerate(cores_of(layer)):
        d[f"core{i}"] = c
    tn.orthonormalize_left()
    for i, c in enumerate(cores_of(layer)):
        d[f"left_core{i}"] = c
    tn.orthonormalize_right()
    for i, c in enumerate(cores_of(layer)):
        d[f"leftright_core{i}"] = c
    npz(f"qr_gauge_{tag}", **d)


if __name__ == "__main__" and os.environ.get("TTK_GOLDEN_BASE", "1") == "1":
    gen_core_update("n4p3r4c1_sq", 4, 3, 4, 1, 96, "square", True, True, 1)
    gen_core_update("n5p4r6c3_sq", 5, 4, 6, 3, 96, "square", True, True, 2)
    gen_core_update("n4p3r4c3_mse", 4, 3, 4, 3, 96, "mse", True, True, 3)
    gen_core_update("n4p5r4c2_xe", 4, 5, 4, 2, 96, "xe", False, False, 4)
    gen_core_update("n3p5r4c2_xe_iris_shape", 3, 5, 4, 2, 120, "xe", False, False, 5)

    gen_swipe("chol_ortho", 5, 4, 6, 1, 200, "square", 64, 11, num_swipes=2, method="ridge_cholesky", eps=1.0,
              eps_decay=0.5, orthonormalize=True, direction="l2r")
    gen_swipe("exact_r2l", 4, 3, 4, 1, 150, "square", 50, 12, num_swipes=2, method="ridge_exact", eps=0.1,
              eps_decay=0.5, orthonormalize=False, direction="r2l")
    gen_swipe("skip_second_fullbatch", 4, 3, 4, 2, 128, "mse", -1, 13, num_swipes=3, method="ridge_cholesky", eps=0.5,
              eps_decay=0.7, skip_second=True)
    gen_swipe("eps_list_per_node", 4, 3, 4, 1, 128, "square", 48, 14, num_swipes=1, method="ridge_exact",
              eps=[1.0, 0.5, 0.25, 0.125], eps_per_node=True, lr=0.7)
    gen_swipe("xe_c2", 4, 5, 4, 2, 160, "xe", 64, 15, num_swipes=2, method="ridge_cholesky", eps=1.0, eps_decay=0.5)
    gen_swipe("gradient", 4, 3, 4, 1, 128, "square", 64, 16, num_swipes=1, method="gradient", lr=0.01)
    gen_swipe("adaptive_maxnorm", 4, 3, 4, 1, 128, "square", 64, 17, num_swipes=1, method="ridge_cholesky", eps=1e-3,
              adaptive_step=True, max_norm=0.8)

    gen_tnml("poly_r6", 400, 5, 6, 3, "polynomial", 0, early_stopping=4, num_swipes=2, batch_size=128)
    gen_tnml("sincos_r4", 300, 4, 4, 3, "sin-cos", 1, early_stopping=3, num_swipes=2, batch_size=100)
    gen_ttreg("reg_n3r4", 2, "regression", N=3, r=4, output_dim=1, early_stopping=5, num_swipes=2, batch_size=100)
    gen_ttreg("iris_n3r4", 3, "classification", N=3, r=4, output_dim=2, early_stopping=8, num_swipes=5,
              batch_size=512)

    gen_lanczos("n4p3r4c1", 4, 3, 4, 1, 128, "square", 48, 21, 2, 8, 1e-6)
    gen_lanczos("n4p4r4c2_xe", 4, 4, 4, 2, 128, "xe", 64, 22, 1, 12, 1e-6)
    gen_scipy("n4p3r4c1", "cg", 4, 3, 4, 1, 128, "square", 48, 23, 2, 5, 1e-6)
    gen_scipy("n4p3r4c1", "minres", 4, 3, 4, 1, 128, "square", 48, 24, 2, 5, 1e-6)

    gen_qr("n5p4r6c1", 5, 4, 6, 1, 31)
    gen_qr("n4p3r4c3", 4, 3, 4, 3, 32)


# ---------------------------------------------------------------- 6. conv-featurised TT
def _conv_layer(N, r, patches, pixels, C, CB, seed):
    torch.manual_seed(seed)
    return layers.TensorConvolutionTrainLayer(N, r, patches, pixels, output_shape=C, convolution_bond=CB)


def _snap_train(tn):
    return [n.tensor.detach().clone() for n in tn.train_nodes]


def gen_conv(tag, N, patches, pixels, r, CB, C, S, seed, kind="xe", dtype=torch.float64, lanczos_iters=5):
    """dtype=torch.float32 runs the reference the way image_convolution_lanczos_CIFAR10.py does: the
    script never sets a default dtype, so nodes, data and every intermediate are float32."""
    torch.set_default_dtype(dtype)
    try:
        _gen_conv(tag, N, patches, pixels, r, CB, C, S, seed, kind, dtype, lanczos_iters)
    finally:
        torch.set_default_dtype(torch.float64)


def _gen_conv(tag, N, patches, pixels, r, CB, C, S, seed, kind, dtype, lanczos_iters):
    layer = _conv_layer(N, r, patches, pixels, C, CB, seed)
    tn = layer.tensor_network
    g = torch.Generator().manual_seed(seed + 100)
    x = torch.rand(S, patches, pixels, generator=g) * 2 - 1
    if kind == "xe":
        lab = torch.randint(0, C + 1, (S,), generator=g)
        y = torch.nn.functional.one_hot(lab, C + 1).to(dtype)
    else:
        y = torch.randn(S, C, generator=g)
    d = {"dtype": np.array(str(dtype).replace("torch.", "")), "N": N, "patches": patches, "pixels": pixels, "r": r, "CB": CB, "C": C, "S": S, "seed": seed, "x": x, "y": y,
         "loss_kind": np.array(kind), "train_names": np.array([n.name for n in tn.train_nodes])}
    for i, t in enumerate(_snap_train(tn)):
        d[f"node{i}"] = t
    loss_fn = make_loss(kind)
    y_pred = tn.forward(x, to_tensor=True)
    loss, gr, H = loss_fn.forward(y_pred, y)
    d["y_pred"], d["loss"], d["g"], d["H"] = y_pred, loss, gr, H
    v_gen = torch.Generator().manual_seed(seed + 7)
    for k, node in enumerate(tn.train_nodes):
        A, b = tn.get_A_b(node, gr, H)
        d[f"A{k}"], d[f"b{k}"] = A, b
        v = torch.randn(node.tensor.shape, generator=v_gen)
        pj = tn.get_J(node, gr)
        coeff = torch.einsum(f"{pj['einsum']},{pj['node_ein']},{pj['dd_loss_ein']}->{pj['coeff_ein']}", pj['J'].tensor, v, H)
        Av = torch.einsum(f"{pj['einsum']},{pj['d_loss_ein']}->{pj['node_ein']}", pj['J'].tensor, coeff)
        d[f"v{k}"], d[f"Av{k}"] = v, Av
    # one ridge_cholesky sweep and one Lanczos sweep, each from the same seeded start
    for which in ("chol", "lanczos"):
        layer2 = _conv_layer(N, r, patches, pixels, C, CB, seed)
        tn2 = layer2.tensor_network
        snaps, losses = [], []
        if which == "chol":
            ok = tn2.accumulating_swipe(x, y, make_loss(kind), batch_size=32, num_swipes=1, method="ridge_cholesky", eps=1.0,
                                        eps_decay=0.5, loss_callback=lambda NS, n, l: losses.append(l),
                                        block_callback=lambda NS, n: snaps.append(_snap_train(tn2)))
        else:
            torch.manual_seed(seed + 5)
            for u, n in enumerate(tn2.train_nodes):
                d[f"lanczos_x0_{u}"] = torch.randn_like(n.tensor)
            torch.manual_seed(seed + 5)
            ok = tn2.lanczos_swipe(x, y, make_loss(kind), batch_size=32, num_swipes=1, max_iter=lanczos_iters, tol=1e-6,
                                   loss_callback=lambda l: losses.append(l),
                                   block_callback=lambda NS, n: snaps.append(_snap_train(tn2)))
            d["lanczos_max_iter"] = lanczos_iters
        d[f"{which}_ok"] = int(ok)
        d[f"{which}_losses"] = np.array(losses)
        d[f"{which}_num_updates"] = len(snaps)
        for u, snap in enumerate(snaps):
            for i, t in enumerate(snap):
                d[f"{which}_u{u}_node{i}"] = t
    d["lanczos_torch_seed"] = seed + 5
    npz(f"conv_tt_{tag}", **d)


# ---------------------------------------------------------------- 7. other topologies on the same path
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests._net_cases import NET_CASES, net_builders as _net_builders  # noqa: E402


# seeds follow this order; new cases go at the end so that the earlier fixtures regenerate unchanged
_NET_ORDER = ["cpd_c1", "cpd_c2", "cumsum_c1", "sum_cpd_type1_c2", "sum_tt_type1_c1", "sum_ttlinear_type1_c1", "ttlinear_c2",
              "graph_two_legs_c2", "graph_out_last_c3"]


def gen_net(case, S=72, features=6):
    kind, C = NET_CASES[case]
    seed = 200 + _NET_ORDER.index(case)
    build = _net_builders(layers, network)[case]
    torch.manual_seed(seed)          # CumSumLayer does not seed itself
    layer = build()
    tn = layer.tensor_network
    g = torch.Generator().manual_seed(seed + 100)
    x = torch.rand(S, features, generator=g) * 2 - 1
    x[:, -1] = 1.0                   # bias column, as the estimators append it
    if kind == "xe":
        y = torch.nn.functional.one_hot(torch.randint(0, C + 1, (S,), generator=g), C + 1).to(torch.float64)
    else:
        y = torch.randn(S, C, generator=g)
    d = {"case": np.array(case), "S": S, "features": features, "C": C, "seed": seed, "x": x, "y": y,
         "loss_kind": np.array(kind), "train_names": np.array([n.name for n in tn.train_nodes])}
    for i, t in enumerate(_snap_train(tn)):
        d[f"node{i}"] = t
    loss_fn = make_loss(kind)
    y_pred = tn.forward(x, to_tensor=True)
    loss, gr, H = loss_fn.forward(y_pred, y)
    d["y_pred"], d["loss"], d["g"], d["H"] = y_pred, loss, gr, H
    v_gen = torch.Generator().manual_seed(seed + 7)
    has_Av = 1
    for k, node in enumerate(tn.train_nodes):
        A, b = tn.get_A_b(node, gr, H)
        d[f"A{k}"], d[f"b{k}"] = A, b
        v = torch.randn(node.tensor.shape, generator=v_gen)
        d[f"v{k}"] = v
        try:
            pj = tn.get_J(node, gr)
            coeff = torch.einsum(f"{pj['einsum']},{pj['node_ein']},{pj['dd_loss_ein']}->{pj['coeff_ein']}", pj['J'].tensor, v, H)
            d[f"Av{k}"] = torch.einsum(f"{pj['einsum']},{pj['d_loss_ein']}->{pj['node_ein']}", pj['J'].tensor, coeff)
        except Exception as e:       # the reference's get_J does not cover every network class
            has_Av = 0
            d["Av_error"] = np.array(f"{type(e).__name__}: {e}"[:200])
    d["has_Av"] = has_Av
    for which in ("chol", "lanczos"):
        torch.manual_seed(seed)
        tn2 = build().tensor_network
        snaps, losses = [], []
        try:
            if which == "chol":
                ok = tn2.accumulating_swipe(x, y, make_loss(kind), batch_size=32, num_swipes=2, method="ridge_cholesky",
                                            eps=1.0, eps_decay=0.5, loss_callback=lambda NS, n, l: losses.append(l),
                                            block_callback=lambda NS, n: snaps.append(_snap_train(tn2)))
            else:
                torch.manual_seed(seed + 5)
                ok = tn2.lanczos_swipe(x, y, make_loss(kind), batch_size=32, num_swipes=1, max_iter=5, tol=1e-6,
                                       loss_callback=lambda l: losses.append(l),
                                       block_callback=lambda NS, n: snaps.append(_snap_train(tn2)))
        except Exception as e:
            d[f"{which}_error"] = np.array(f"{type(e).__name__}: {e}"[:200])
            d[f"has_{which}"] = 0
            continue
        d[f"has_{which}"] = 1
        d[f"{which}_ok"] = int(ok)
        d[f"{which}_losses"] = np.array(losses)
        d[f"{which}_num_updates"] = len(snaps)
        for u, snap in enumerate(snaps):
            for i, t in enumerate(snap):
                d[f"{which}_u{u}_node{i}"] = t
    d["lanczos_torch_seed"] = seed + 5
    npz(f"net_{case}", **d)
    print(case, "has_Av", has_Av, "chol", d.get("has_chol"), d.get("chol_error"), "lanczos", d.get("has_lanczos"),
          d.get("lanczos_error"), d.get("Av_error"))


def gen_extra():
    if os.environ.get("TTK_GOLDEN_CONV", "1") == "1":
        gen_conv("cbm1", 3, 10, 5, 4, -1, 3, 64, 41)
        gen_conv("cb2", 3, 10, 5, 4, 2, 3, 64, 42)
        gen_conv("cb2_sq_c1", 3, 6, 4, 3, 2, 1, 48, 43, kind="square")
    if os.environ.get("TTK_GOLDEN_NETS", "1") == "1":
        for case in NET_CASES:
            gen_net(case)
    # estimator level, every model_type family (models/tensor_train.py:133-208)
    common = dict(N=3, r=4, output_dim=1, early_stopping=4, num_swipes=2, batch_size=100)
    gen_ttreg("type1_n3r4", 6, "regression", model_type="tt_type1", **common)
    gen_ttreg("cpd_n3r4", 7, "regression", model_type="cpd", **common)
    gen_ttreg("cpd_type1_n3r4", 8, "regression", model_type="cpd_type1", **common)
    gen_ttreg("cumsum_n3r4", 9, "regression", model_type="tt", cum_sum=True, **common)
    gen_ttreg("linear_n3r4", 10, "regression", model_type="tt", linear_dim=3, **common)
    gen_ttreg("linear_type1_ops_n3r4", 11, "regression", model_type="tt_type1", linear_dim=3, train_operator=True, **common)
    gen_ttreg("iris_cpd_type1", 12, "classification", model_type="cpd_type1", N=3, r=4, output_dim=2, early_stopping=6,
              num_swipes=3, batch_size=512)


def gen_many_patches():
    """Conv columns with at least 16 patches (the sites of the MNIST / CIFAR configurations have 50 /
    65): the build contracts the features of such a site first, a different code route from the
    10-patch fixtures above.  One of them in float32, the arithmetic type of the CIFAR script."""
    # three Lanczos iterations: fewer than the 5 unknowns of a filter node, whose Krylov space five
    # iterations exhaust (the tridiagonal system then sits at the edge of breakdown and its solution is
    # rounding noise, in the reference as anywhere)
    gen_conv("p17_cbm1", 3, 17, 5, 4, -1, 3, 64, 51, lanczos_iters=3)
    gen_conv("p17_cb2", 3, 17, 5, 4, 2, 3, 64, 52, lanczos_iters=3)
    gen_conv("p17_cbm1_f32", 3, 17, 5, 4, -1, 3, 64, 53, dtype=torch.float32, lanczos_iters=3)
    gen_conv("p10_cb2_f32", 3, 10, 5, 4, 2, 3, 64, 54, dtype=torch.float32, lanczos_iters=3)


def gen_conv_full_shape(tag, N, patches, pixels, r, CB, C, S, seed, lanczos_iters=4, dtype=torch.float64):
    torch.set_default_dtype(dtype)
    try:
        _gen_conv_full_shape(tag, N, patches, pixels, r, CB, C, S, seed, lanczos_iters, dtype)
    finally:
        torch.set_default_dtype(torch.float64)


def _gen_conv_full_shape(tag, N, patches, pixels, r, CB, C, S, seed, lanczos_iters, dtype):
    """The image scripts' own column shapes (image_convolution_lanczos_MNIST.py:29-32,76: 50 patches of 17 pixels, bond
    10, convolution bond 4, 9 trained logits) on a few samples.  The Gram matrices of such nodes are 5000 x 5000 and stay
    out of the fixture: forward, loss, b and A v of every train node and the first two updates of a Lanczos sweep.
    dtype=torch.float32: the CIFAR script's arithmetic (CIFAR_lanczos_swipe.py:35-38,68-108 never sets a dtype)."""
    layer = _conv_layer(N, r, patches, pixels, C, CB, seed)
    tn = layer.tensor_network
    g = torch.Generator().manual_seed(seed + 100)
    # patch rows as the scripts build them: a zero last row, a zero last column, a one in the corner; labels from a
    # linear teacher; the loss weight the scripts' normalisation gives (1 / spread of the initial outputs)
    x = torch.randn(S, patches, pixels, generator=g)
    x[:, -1, :] = 0.0
    x[:, :, -1] = 0.0
    x[:, -1, -1] = 1.0
    teacher = torch.randn(patches * pixels, C + 1, generator=g)
    y = torch.nn.functional.one_hot((x.reshape(S, -1) @ teacher).argmax(dim=1), C + 1).to(dtype)
    w = 1.0 / tn.forward(x, to_tensor=True).std().item()
    d = {"dtype": np.array(str(dtype).replace("torch.", "")), "N": N, "patches": patches, "pixels": pixels, "r": r, "CB": CB, "C": C, "S": S, "seed": seed, "x": x, "y": y,
         "w": w, "train_names": np.array([n.name for n in tn.train_nodes])}
    for i, t in enumerate(_snap_train(tn)):
        d[f"node{i}"] = t
    make_loss = lambda kind: bregman.XEAutogradBregman(w=w)   # noqa: E731
    loss_fn = make_loss("xe")
    y_pred = tn.forward(x, to_tensor=True)
    loss, gr, H = loss_fn.forward(y_pred, y)
    d["y_pred"], d["loss"] = y_pred, loss
    v_gen = torch.Generator().manual_seed(seed + 7)
    for k, node in enumerate(tn.train_nodes):
        v = torch.randn(node.tensor.shape, generator=v_gen)
        pj = tn.get_J(node, gr)
        coeff = torch.einsum(f"{pj['einsum']},{pj['node_ein']},{pj['dd_loss_ein']}->{pj['coeff_ein']}", pj['J'].tensor, v, H)
        d[f"Av{k}"] = torch.einsum(f"{pj['einsum']},{pj['d_loss_ein']}->{pj['node_ein']}", pj['J'].tensor, coeff)
        d[f"b{k}"] = torch.einsum(f"{pj['einsum']},{pj['d_loss_ein']}->{pj['node_ein']}", pj['J'].tensor, gr)
        d[f"v{k}"] = v
    # A Lanczos pass over the first two train nodes (filter C1 and core A1).  Further on the reference's un-ridged
    # Gauss-Newton steps leave the basin on so few samples (loss 2.7, 2.3, 91, 2e3, 1e18, then a singular tridiagonal
    # system), and its ridge_cholesky sweep does not run at this shape (get_A_b asks for 518 GB).
    tn2 = _conv_layer(N, r, patches, pixels, C, CB, seed).tensor_network
    tn2.train_nodes = tn2.train_nodes[:2]
    snaps, losses = [], []
    torch.manual_seed(seed + 5)
    ok = tn2.lanczos_swipe(x, y, make_loss("xe"), batch_size=32, num_swipes=1, max_iter=lanczos_iters, tol=1e-6,
                           loss_callback=lambda l: losses.append(l),
                           block_callback=lambda NS, n: snaps.append(_snap_train(tn2)))
    d["lanczos_ok"] = int(ok)
    d["lanczos_losses"] = np.array(losses)
    d["lanczos_num_updates"] = len(snaps)
    d["lanczos_forward"] = tn2.forward(x, to_tensor=True)
    for u, snap in enumerate(snaps):
        for i, t in enumerate(snap):
            d[f"lanczos_u{u}_node{i}"] = t
    d["lanczos_max_iter"], d["lanczos_torch_seed"] = lanczos_iters, seed + 5
    npz(f"convfull_{tag}", **d)


if __name__ == "__main__" and os.environ.get("TTK_GOLDEN_FULL_SHAPE", "1") == "1":
    gen_conv_full_shape("mnist_r10_cb4", 3, 50, 17, 10, 4, 9, 96, 61, lanczos_iters=4)
    gen_conv_full_shape("mnist_r24", 3, 50, 17, 24, -1, 9, 96, 62, lanczos_iters=4)
    gen_conv_full_shape("cifar_r48_f32", 3, 65, 49, 48, -1, 9, 48, 63, lanczos_iters=4, dtype=torch.float32)
if __name__ == "__main__" and os.environ.get("TTK_GOLDEN_EXTRA", "1") == "1":
    gen_extra()
if __name__ == "__main__" and os.environ.get("TTK_GOLDEN_MANY_PATCHES", "1") == "1":
    gen_many_patches()
if __name__ == "__main__" and os.environ.get("TTK_GOLDEN_ONLY_NETS"):
    # e.g. TTK_GOLDEN_BASE=0 TTK_GOLDEN_EXTRA=0 TTK_GOLDEN_MANY_PATCHES=0 TTK_GOLDEN_ONLY_NETS=graph_two_legs_c2,graph_out_last_c3
    for case in os.environ["TTK_GOLDEN_ONLY_NETS"].split(","):
        gen_net(case)
