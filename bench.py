"""DMRG core-update benchmark: 16-site, bond-32, degree-5 tensor train on a 60k-sample synthetic set.

    python bench.py --gpus 1 --steps 12 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one core update of a middle site (rl = rr = 32, p = 6, D = 6144) over all 54 000
training samples through the public drop-in API (TensorNetwork.accumulating_swipe with
method='ridge_cholesky', orthonormalize=True, i.e. what TNMLRegressor.fit runs): environment
refresh, prediction, loss, Gram + RHS build, scaled ridge Cholesky solve, core update, QR
gauge move, and the per-update validation pass of the estimator's early-stopping hook.
With N > 1 the training samples are sharded over the ranks (strong scaling) and one
all-reduce of [compressed Gram | RHS] precedes the replicated solve.

Rank 0 prints ONE JSON line.  `roofline` describes the Gram kernel (HIP events recorded on its
launch stream inside the timed region); `cpu_baseline` is the CPU oracle (oracle/tt_oracle.py)
timed on the host cores on a bounded sample; `variants.cg` / `variants.lanczos` are the matrix-free
paths on the same model, `variants.sweep` a whole first swipe of fit() (also `sweep_updates_per_sec`),
`variants.mnist_lanczos*` / `variants.cifar_lanczos_f32` one lanczos_swipe pass over a conv tensor
train at the image scripts' sizes, each with the roofline of its dominant kernel and, as parity, the
size-independent identities its kernels must satisfy (N = 1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_SITES, BOND, DEGREE, S_ALL, N_FEAT = 16, 32, 5, 60000, 16
FP64_MFMA_PEAK_TFLOPS = 78.6    # MI355X dense fp64 matrix peak (v_mfma_f64_16x16x4: 64 cycles/SIMD at 2.4 GHz)
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X dense fp32 matrix peak (v_mfma_f32_16x16x4: 32 cycles/SIMD)
HBM_PEAK_GBS = 8000.0


def synthetic_problem(device):
    from tensornetworksfork_amd.models._common import split_train_val
    from tensornetworksfork_amd.models.tnml import polynomial_basis
    rng = np.random.RandomState(0)
    X = rng.uniform(-1, 1, (S_ALL, N_FEAT))
    y = np.sin(X).sum(1, keepdims=True) + 0.01 * rng.randn(S_ALL, 1)
    X = torch.tensor(X, dtype=torch.float64, device=device)
    y = torch.tensor(y, dtype=torch.float64, device=device)
    Xt, yt, Xv, yv = split_train_val(X, y, 42, 0.1)
    return polynomial_basis(Xt, DEGREE), yt.contiguous(), polynomial_basis(Xv, DEGREE), yv.contiguous()


def make_model(device):
    from tensornetworksfork_amd.tensor.layers import TensorTrainLayer
    layer = TensorTrainLayer(N_SITES, BOND, DEGREE + 1, output_shape=1, constrict_bond=True, seed=42).to(device)
    layer.tensor_network.orthonormalize_left()
    return layer


class StepCounter:
    """convergence_criterion hook: validation RMSE after every update (as EarlyStopping does),
    stop after `limit` updates."""

    def __init__(self, tn, F_val, y_val, limit):
        from tensornetworksfork_amd.models._common import root_mean_squared_error_torch
        self.tn, self.F_val, self.y_val, self.limit = tn, F_val, y_val, limit
        self.rmse = root_mean_squared_error_torch
        self.count = 0
        self.last = None

    def __call__(self):
        self.last = self.rmse(self.y_val, self.tn.forward_batch(self.F_val, 512))
        self.count += 1
        return self.count >= self.limit


def run_updates(tn, F, y, F_val, y_val, n_updates, eps0=1.0):
    from tensornetworksfork_amd.tensor.bregman import SquareBregFunction
    mid = tn.main_nodes[2:N_SITES - 2]
    hook = StepCounter(tn, F_val, y_val, n_updates)
    tn.accumulating_swipe(F, y, SquareBregFunction(), node_order=(mid, mid), batch_size=512, lr=1.0, eps=eps0,
                          eps_decay=0.5, convergence_criterion=hook, orthonormalize=True, method='ridge_cholesky',
                          num_swipes=10 ** 6, direction='l2r')
    assert hook.count == n_updates
    return hook.last


def cg_variant(device, F, y, steps):
    """Matrix-free CG (scipy.sparse.linalg.cg semantics, max_iter=100, rtol=1e-6) on middle sites: the
    solver BASELINE's configuration list names for this model.  One iteration = the fused product
    kernel (J^T H J p in one pass over the samples, direction update in its prologue) + the slab
    reduction with the x / r updates; its roofline entry is the product kernel's."""
    import scipy.sparse.linalg as spla
    from tensornetworksfork_amd import ops
    from tensornetworksfork_amd.tensor.bregman import SquareBregFunction
    layer = make_model(device)
    tn = layer.tensor_network
    all_nodes = tn.train_nodes
    try:
        tn.train_nodes = tn.main_nodes[2:3]
        tn.scipy_swipe(F, y, SquareBregFunction(), spla.cg, batch_size=512, max_iter=100, tol=1e-6)   # warm-up
        tn.train_nodes = tn.main_nodes[3:3 + steps]
        ops.gram_timing(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tn.scipy_swipe(F, y, SquareBregFunction(), spla.cg, batch_size=512, max_iter=100, tol=1e-6)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        mv_ms = ops.gram_timing_read()
        ops.gram_timing(0)
    finally:
        tn.train_nodes = all_nodes
    S, D = F[0].shape[0], BOND * (DEGREE + 1) * BOND
    per = dt / steps
    it_bytes = S * (BOND + DEGREE + 1 + BOND + 1) * 8.0
    it_flops = 4.0 * S * D
    avg_ms = float(np.mean(mv_ms)) if mv_ms else float("nan")
    tfl = it_flops / (avg_ms * 1e-3) * 1e-12
    return {"value": 1.0 / per, "unit": "core-updates/s", "ms_per_step": per * 1e3, "steps": steps, "max_iter": 100,
            "rtol": 1e-6, "samples_per_sec": S / per, "iterations_timed": len(mv_ms),
            "us_per_iteration": per * 1e6 / 100.0,
            "roofline": {"kernel": "matvec_fused_kernel<8,3,1>", "bound": "mfma", "achieved": tfl,
                         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / FP64_MFMA_PEAK_TFLOPS,
                         "avg_launch_us": avg_ms * 1e3, "flops_per_launch": it_flops,
                         "algorithmic_bytes_per_launch": it_bytes,
                         "hbm_gbs": it_bytes / (avg_ms * 1e-3) * 1e-9,
                         "hbm_frac": it_bytes / (avg_ms * 1e-3) * 1e-9 / HBM_PEAK_GBS,
                         "traffic": _recorded_traffic("cg_traffic.json")}}


def _recorded_traffic(name):
    """Fabric bytes per launch of a kernel from the committed counter passes (profiles/<name>; rocprofv3 --pmc
    FETCH_SIZE and WRITE_SIZE cannot be collected inside a timed run), or None."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name))).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def _timed_kernel(which, fn):
    """Run fn() with HIP events armed around every launch of one kernel (ttk_gram_timing: 2 the fused
    matrix-free product, 3 the features-first right-hand side, 4 the resident right-environment step, 5 / 6 the
    second / first half of the operator of a column between convolution bonds);
    returns the launch durations in ms."""
    from tensornetworksfork_amd import ops
    ops.gram_timing(which)
    torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
    ms = ops.gram_timing_read()
    ops.gram_timing(0)
    return ms


def _mfma_roofline(kernel, flops, ms_list, peak, bytes_=None):
    avg = float(np.mean(ms_list)) if ms_list else float("nan")
    tfl = flops / (avg * 1e-3) * 1e-12
    out = {"kernel": kernel, "bound": "mfma", "achieved": tfl, "peak": peak, "unit": "TFLOP/s", "frac": tfl / peak,
           "launches": len(ms_list), "avg_launch_us": avg * 1e3, "flops_per_launch": flops, "traffic": None}
    if bytes_ is not None:
        out["algorithmic_bytes_per_launch"] = bytes_
        out["hbm_gbs"] = bytes_ / (avg * 1e-3) * 1e-9
        out["hbm_frac"] = out["hbm_gbs"] / HBM_PEAK_GBS
    return out


def lanczos_variant(device, F, y, steps=8, max_iter=100):
    """lanczos_swipe (tensor/network.py:709-832) on middle sites of the timed model: one product
    v -> J^T H J v (the fused one-pass kernel) and one fused vector kernel per iteration, the exit test on
    the device, the tridiagonal solve and x0 + V y in two more launches."""
    from tensornetworksfork_amd.tensor.bregman import SquareBregFunction
    layer = make_model(device)
    tn = layer.tensor_network
    all_nodes = tn.train_nodes
    try:
        tn.train_nodes = tn.main_nodes[2:3]
        torch.manual_seed(0)
        tn.lanczos_swipe(F, y, SquareBregFunction(), batch_size=512, max_iter=max_iter, tol=1e-6)   # warm-up
        tn.train_nodes = tn.main_nodes[3:3 + steps]
        holder = {}

        def run():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tn.lanczos_swipe(F, y, SquareBregFunction(), batch_size=512, max_iter=max_iter, tol=1e-6)
            torch.cuda.synchronize()
            holder["dt"] = time.perf_counter() - t0
        mv_ms = _timed_kernel(2, run)
    finally:
        tn.train_nodes = all_nodes
    S, D = F[0].shape[0], BOND * (DEGREE + 1) * BOND
    per = holder["dt"] / steps
    roof = _mfma_roofline("matvec_fused_kernel<8,3,0>", 4.0 * S * D, mv_ms, FP64_MFMA_PEAK_TFLOPS,
                          S * (BOND + DEGREE + 1 + BOND + 1) * 8.0)
    roof["traffic"] = _recorded_traffic("cg_traffic.json")     # the same product kernel without the CG prologue
    return {"value": 1.0 / per, "unit": "core-updates/s", "ms_per_step": per * 1e3, "steps": steps, "max_iter": max_iter,
            "tol": 1e-6, "products_per_update": len(mv_ms) / steps, "us_per_iteration": per * 1e6 / (len(mv_ms) / steps),
            "roofline": roof}


def _conv_problem(device, S, patches, pixels, bond, dtype, conv_bond):
    """Synthetic data of the image scripts' shapes (image_convolution_lanczos_MNIST.py:29-32, CIFAR_lanczos_swipe.py:
    35-38: unfolded patches with a zero row / column and a corner 1), labels from a linear teacher, the scripts' model
    (3 carriages, 9 trained logits) and their loss scaling."""
    from tensornetworksfork_amd.tensor.bregman import XEAutogradBregman
    from tensornetworksfork_amd.tensor.layers import TensorConvolutionTrainLayer
    torch.set_default_dtype(dtype)
    try:
        torch.manual_seed(5)
        layer = TensorConvolutionTrainLayer(num_carriages=3, bond_dim=bond, num_patches=patches, patch_pixels=pixels,
                                            output_shape=9, convolution_bond=conv_bond).to(device)
    finally:
        torch.set_default_dtype(torch.float64)
    g = torch.Generator(device=device).manual_seed(6)
    x = torch.randn(S, patches, pixels, generator=g, dtype=dtype, device=device)
    x[:, -1, :] = 0.0
    x[:, :, -1] = 0.0
    x[:, -1, -1] = 1.0
    teacher = torch.randn(patches * pixels, 10, generator=g, dtype=dtype, device=device)
    y = torch.nn.functional.one_hot((x.reshape(S, -1) @ teacher).argmax(dim=1), 10).to(torch.float64)
    with torch.inference_mode():
        w = 1.0 / layer.tensor_network.forward(x[:64].contiguous(), to_tensor=True).std().item()
    return layer, x, y, XEAutogradBregman(w=w)


def _conv_identities(tn, x, y, bf):
    """Size-independent identities that tie the kernels on every train node's route to each other (no CPU answer
    exists at these sizes): linearity f(node + v) - f(node) = J v, adjointness <b, v> = <g, J v>, symmetry
    <u, A v> = <A u, v> and factorisation <v, A v> = sum_s (Jv)^T H (Jv) of A = J^T H J.  Worst relative residual
    of each over the nodes."""
    f64 = torch.float64
    f0 = tn.forward(x, to_tensor=True).to(f64)
    _, g, H = bf.forward(f0, y)
    g, H = g.to(f64), H.to(f64)
    gen = torch.Generator(device=x.device).manual_seed(7)
    worst = {"linearity": 0.0, "adjoint": 0.0, "symmetry": 0.0, "factorisation": 0.0}
    for node in tn.train_nodes:
        dt = node.tensor.dtype
        scale = node.tensor.abs().max().item()
        v = (torch.randn(node.tensor.shape, generator=gen, device=x.device, dtype=f64) * scale).to(dt)
        u = (torch.randn(node.tensor.shape, generator=gen, device=x.device, dtype=f64) * scale).to(dt)
        d, lp = tn._local(node)
        Jv = lp.jv(tn._engine.to_canon(node, v)).to(f64)
        old = node.tensor
        node.tensor = old + v
        f1 = tn.forward(x, to_tensor=True).to(f64)
        node.tensor = old
        tn.forward(x)
        b = tn.get_b(node, g).to(f64)
        Av, Au = tn.matvec(node, v, H).to(f64), tn.matvec(node, u, H).to(f64)
        quad = (Jv.unsqueeze(2) * H * Jv.unsqueeze(1)).sum()      # sum_s (Jv)_s^T H_s (Jv)_s
        res = {"linearity": ((f1 - f0) - Jv).norm() / Jv.norm(),
               "adjoint": abs((b * v.to(f64)).sum() - (g * Jv).sum()) / (g.norm() * Jv.norm()),
               "symmetry": abs((u.to(f64) * Av).sum() - (v.to(f64) * Au).sum()) / (Av.norm() * u.to(f64).norm()),
               "factorisation": abs((v.to(f64) * Av).sum() - quad) / abs(quad)}
        for k, r in res.items():
            worst[k] = max(worst[k], float(r.item()))
    return worst


def conv_lanczos_variant(device, S, patches, pixels, bond, dtype, conv_bond, tol, max_iter=12, repeats=3, what="",
                         shard=False):
    """One lanczos_swipe pass (tensor/network.py:709-832) over the six train nodes (filters C_i, carriages A_i) of a
    conv tensor train at an image script's size.  max_iter = 12: a filter node has `pixels` unknowns, beyond that
    many iterations its Krylov space is exhausted and the reference's own tridiagonal solve returns rounding noise."""
    layer, x, y, bf = _conv_problem(device, S, patches, pixels, bond, dtype, conv_bond)
    tn = layer.tensor_network
    world = 1
    if shard:   # every rank builds the same problem from the same seeds and works on its own block of samples
        import torch.distributed as dist
        world = dist.get_world_size()
        tn.distribute(shard if shard is not True else "auto")
        parity = None
    else:
        parity = _conv_identities(tn, x, y, bf)
        parity.update({"tolerance": tol, "ok": bool(max(parity.values()) <= tol),
                       "what": "worst relative residual over the six train nodes of: f(node+v)-f(node) = Jv, "
                               "<b,v> = <g,Jv>, <u,Av> = <Au,v>, <v,Av> = sum (Jv)^T H (Jv)"})
    nodes0 = [n.tensor.clone() for n in tn.train_nodes]
    stamps = []

    def block(NS, node):
        torch.cuda.synchronize()
        stamps.append((node.name, time.perf_counter()))

    sweeps, per_node = [], {}
    for rep in range(repeats + 1):
        for n, t in zip(tn.train_nodes, nodes0):
            n.tensor = t.clone()
        stamps.clear()
        torch.manual_seed(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ok = tn.lanczos_swipe(x, y, bf, batch_size=512, num_swipes=1, max_iter=max_iter, block_callback=block)
        assert ok
        if rep:   # the first pass warms caches and workspaces up
            sweeps.append(stamps[-1][1] - t0)
            prev = t0
            for name, t in stamps:
                per_node.setdefault(name, []).append((t - prev) * 1e3)
                prev = t
    sweep_ms = float(np.mean(sweeps)) * 1e3
    node_ms = {k: float(np.mean(v)) for k, v in per_node.items()}
    # the dominant kernels, on the middle carriage alone so that every timed launch has one shape
    mid = [n for n in tn.train_nodes if n in tn.main_nodes][1]
    rl, p, rr = bond, patches, bond
    all_nodes = tn.train_nodes
    roofs = []
    try:
        tn.train_nodes = [mid]
        Sl = (S + world - 1) // world
        if conv_bond > 0:
            # a column between two convolution bonds (ca = cb = conv_bond): the two halves of its operator, fp64; FLOPs
            # of the factorisations the kernels use: J^T: T_a = sum_b G_ab R_b, then sum_a T_a (x) GL_a; J: Y_b = v . R_b,
            # then sum_{a,b,p} G_ab Y_b
            ca = cb = conv_bond
            kernels = ((5, "bond_rhs_kernel", 2.0 * Sl * ca * p * rr * (cb + rl)),
                       (6, "bond_jv_kernel", 2.0 * Sl * cb * p * rl * (rr + ca)))
        else:
            kernels = ((3, "rhs_feature_kernel", 2.0 * Sl * rl * p * rr), (4, "env_right_resident_kernel", 2.0 * Sl * rl * p * rr))
        for which, kernel, flops in kernels:
            torch.manual_seed(1)
            ms = _timed_kernel(which, lambda: tn.lanczos_swipe(x, y, bf, batch_size=512, num_swipes=1, max_iter=max_iter))
            f32 = dtype == torch.float32 and conv_bond <= 0
            peak = FP32_MFMA_PEAK_TFLOPS if f32 else FP64_MFMA_PEAK_TFLOPS
            if ms:
                roofs.append(_mfma_roofline(kernel + ("<float>" if f32 else "<double>"), flops, ms, peak))
    finally:
        tn.train_nodes = all_nodes
    iters = max_iter + 1      # products per node: the residual of the start vector + one per iteration
    return {"workload": what, "samples": S, "ranks": world, "patches": patches, "pixels": pixels, "bond": bond, "conv_bond": conv_bond,
            "dtype": "f32" if dtype == torch.float32 else "f64", "max_iter": max_iter, "sweep_ms": sweep_ms,
            "node_updates_per_sec": len(node_ms) / (sweep_ms * 1e-3), "samples_per_sec_sweep": S / (sweep_ms * 1e-3),
            "node_ms": node_ms, "us_per_krylov_iteration": {k: v * 1e3 / iters for k, v in node_ms.items()},
            "roofline": roofs[0] if roofs else None, "roofline_env": roofs[1] if len(roofs) > 1 else None,
            "parity": parity, "repeats": repeats}


def parity_and_cpu_baseline(device, F, y, sites=(6, 7, 8), eps=1.0):
    """ONE problem, two paths.  For each of `sites` (middle sites, D = 6144) the core update of the
    timed workload - same features, targets, cores after orthonormalize_left, ridge - is computed by
    the CPU oracle (environments from scratch, Gram + RHS over all 54 000 samples, scaled
    ridge_cholesky solve, core update: tensor/network.py:174-217,293-327 as models/tnml.py:218-234
    drives them) and by the HIP path through the public API, both starting from the same cores.
    Returns (parity, cpu_baseline): relative Frobenius errors of loss, b, A, step and updated core
    (worst site), and the oracle's timing on the host cores."""
    from oracle import tt_oracle as O
    from tensornetworksfork_amd.tensor.bregman import SquareBregFunction
    layer = make_model(device)
    tn = layer.tensor_network
    nodes = tn.main_nodes
    cores0 = [n.tensor.detach().clone() for n in nodes]
    cores_cpu = [c.cpu() for c in cores0]
    xs_cpu = [f.cpu() for f in F]
    y_cpu = y.cpu()
    S_train = xs_cpu[0].shape[0]
    bf = SquareBregFunction()

    def rel(a, ref):
        ref = ref.to(a.device)
        return float(((a - ref).norm() / ref.norm()).item())

    worst = {"loss_rel_err": 0.0, "b_rel_err": 0.0, "A_rel_err": 0.0, "step_rel_err": 0.0, "core_rel_err": 0.0}
    t_gram = t_solve = 0.0
    step_norm = 0.0
    for k in sites:
        t0 = time.perf_counter()
        A, b, loss, g, H = O.site_problem(cores_cpu, xs_cpu, k, y_cpu, 'square')
        t1 = time.perf_counter()
        step = O.solve_system(A, b, O.canon(cores_cpu)[k], 'ridge_cholesky', eps)
        new_core = O.update_core(cores_cpu[k], step.reshape(cores_cpu[k].shape))
        t2 = time.perf_counter()
        t_gram += t1 - t0
        t_solve += t2 - t1
        step_norm = float(step.norm())
        for n, c in zip(nodes, cores0):
            n.tensor = c.clone()
        node = nodes[k]
        y_pred = tn.forward(F, to_tensor=True)
        _, gh, Hh = bf.forward(y_pred, y)
        A_h, b_h = tn.get_A_b(node, gh, Hh)
        errs = {"b_rel_err": rel(b_h.reshape(-1), b.reshape(-1)), "A_rel_err": rel(A_h.reshape(-1), A.reshape(-1))}
        del A_h, A
        losses = []
        ok = tn.accumulating_swipe(F, y, bf, node_order=([node], []), batch_size=-1, lr=1.0, eps=eps,
                                   method='ridge_cholesky', skip_second=True, num_swipes=1,
                                   loss_callback=lambda NS, n, l: losses.append(l))
        assert ok and len(losses) == 1
        errs["loss_rel_err"] = abs(losses[0] - float(loss)) / abs(float(loss))
        errs["step_rel_err"] = rel((node.tensor - cores0[k]).reshape(-1), step.reshape(-1))
        errs["core_rel_err"] = rel(node.tensor.reshape(-1), new_core.reshape(-1))
        for key, v in errs.items():
            worst[key] = max(worst[key], v)
    n = len(sites)
    per = (t_gram + t_solve) / n
    parity = dict(worst)
    parity.update({"sites": list(sites), "eps": eps, "tolerance": 1e-5,
                   "ok": bool(max(worst.values()) <= 1e-5),
                   "what": "worst relative Frobenius error over the sites; HIP path vs CPU oracle on identical inputs"})
    cpu = {"value": 1.0 / per, "unit": "core-updates/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"oracle environments + Gram + RHS on all {S_train} samples ({t_gram / n:.2f} s) + D=6144 "
                     f"ridge_cholesky solve + core update ({t_solve / n:.2f} s), mean of {n} middle sites of the timed "
                     f"problem ({t_gram + t_solve:.1f} s of CPU work); the same updates are the parity check",
           "samples_per_sec": S_train / per, "step_norm": step_norm}
    return parity, cpu


def sweep_variant(device, F, y, F_val, y_val, repeats=3):
    """Whole first swipes of TNMLRegressor.fit's call (31 core updates: every site left to right, then
    back, edge sites included, validation pass after every update), each from a fresh model."""
    from tensornetworksfork_amd.tensor.bregman import SquareBregFunction
    times, updates = [], 0
    for rep in range(repeats + 1):
        layer = make_model(device)
        tn = layer.tensor_network
        hook = StepCounter(tn, F_val, y_val, 10 ** 9)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ok = tn.accumulating_swipe(F, y, SquareBregFunction(), batch_size=512, lr=1.0, eps=1.0, eps_decay=0.5,
                                   convergence_criterion=hook, orthonormalize=True, method='ridge_cholesky',
                                   num_swipes=1, direction='l2r')
        torch.cuda.synchronize()
        if rep:      # the first one warms caches and workspaces up
            times.append(time.perf_counter() - t0)
        updates = hook.count
        assert ok
    t = float(np.mean(times))
    S = F[0].shape[0]
    return {"updates_per_sweep": updates, "sweep_ms": t * 1e3, "core_updates_per_sec": updates / t,
            "samples_per_sec_sweep": S / t, "repeats": repeats, "val_rmse": hook.last}


def dgemm_probe(device, n=8192, reps=3):
    """Library DGEMM rate on this box right now (torch.mm, fp64): the second denominator for the Gram kernel's
    rate next to the vendor peak; boxes of one pool differ by a few per cent under fp64 matrix load."""
    g = torch.Generator(device=device).manual_seed(0)
    a = torch.rand(n, n, generator=g, dtype=torch.float64, device=device)
    b = torch.rand(n, n, generator=g, dtype=torch.float64, device=device)
    torch.mm(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        torch.mm(a, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return {"tflops": 2.0 * n ** 3 / (ms * 1e-3) * 1e-12, "what": f"torch.mm fp64 {n}^3, mean of {reps}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.set_default_dtype(torch.float64)
    # rehearsals on a one-GPU box: TTK_BENCH_REHEARSE=1 puts every rank on cuda:0 and exchanges through gloo;
    # TTK_BENCH_REHEARSE=rccl (one rank) takes the multi-rank path through RCCL with every collective issued
    rehearse = os.environ.get("TTK_BENCH_REHEARSE") == "1"
    rccl_one = os.environ.get("TTK_BENCH_REHEARSE") == "rccl" and world == 1
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    elif rccl_one:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    sharded = world > 1 or rccl_one
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from tensornetworksfork_amd import ops
    F, y, F_val, y_val = synthetic_problem(device)
    layer = make_model(device)
    tn = layer.tensor_network
    shard = "auto"
    if rccl_one:
        from tensornetworksfork_amd.parallel import Shard
        shard = Shard(always_collective=True)
    if sharded:
        tn.distribute(shard)
    S_train = F[0].shape[0]

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if args.warmup > 0:
        run_updates(tn, F, y, F_val, y_val, args.warmup)
    ops.gram_timing(True)
    tn._marks, tn._mark_count, tn._mark_every = [], -1, 4   # device events at the phase boundaries of every 4th update
    fence()
    t0 = time.perf_counter()
    val_rmse = run_updates(tn, F, y, F_val, y_val, args.steps, eps0=0.5)
    fence()
    elapsed = time.perf_counter() - t0
    gram_ms = ops.gram_timing_read()
    ops.gram_timing(False)
    marks, tn._marks = tn._marks, None
    # The sweep queues the next node's Gram ahead; the one queued behind the last timed update is called off when the
    # step counter ends the sweep (most of its workgroups leave at once).  That launch belongs to no update: the
    # phase means and the kernel average below take the first `steps` updates, the wall clock covers everything.
    spans = {"gram_ms": [], "comm_ms": [], "solve_ms": []}
    for (n0, e0, i0), (n1, e1, i1) in zip(marks[:-1], marks[1:]):
        key = {("front", "gram"): "gram_ms", ("gram", "comm"): "comm_ms", ("pre", "solve"): "solve_ms"}.get((n0, n1))
        if key and i0 == i1 and i0 < args.steps:
            spans[key].append(e0.elapsed_time(e1))
    phase = {k: float(np.mean(v)) if v else 0.0 for k, v in spans.items()}
    called_off = len(gram_ms) - args.steps
    gram_ms = gram_ms[:args.steps]
    if dist is not None:
        tmax = torch.tensor([elapsed], device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()

    ups = args.steps / elapsed
    S_local = S_train // world + (1 if S_train % world else 0)
    MP = BOND * (BOND + 1) // 2
    # distinct entries of the Gram x samples: (l<=m) x classes of (p,q) x (r<=t); the polynomial basis
    # X_p = x^p has X_p X_q = x^(p+q), i.e. 2*degree+1 classes instead of (degree+1)(degree+2)/2 pairs
    from tensornetworksfork_amd.tensor.engine import pair_classes
    rep = pair_classes(F[N_SITES // 2][:4096].contiguous())
    NX = (DEGREE + 1) * (DEGREE + 2) // 2 if rep is None else len(set(rep))
    gram_flops = 2.0 * S_local * MP * NX * MP
    D = BOND * (DEGREE + 1) * BOND
    avg_ms = float(np.mean(gram_ms)) if gram_ms else float("nan")
    achieved = gram_flops / (avg_ms * 1e-3) * 1e-12
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "gram_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "dmrg_core_updates_per_sec", "value": ups, "unit": "core-updates/s",
        "samples_per_sec": ups * S_train, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "step_breakdown_ms": dict(phase, other_ms=elapsed / args.steps * 1e3 - sum(phase.values()),
                                  what="this rank, per core update, from device events around every 4th update: Gram (pack + kernel + slab "
                                       "reduction), exchange (one all-reduce of [compressed Gram | b]), scaled ridge "
                                       "Cholesky solve (replicated on every rank), everything else"),
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "cfg2 TNML tensor train: 16 sites, bond 32, polynomial degree 5 (p=6), 60000x16 synthetic "
                               "samples (54000 train / 6000 val); middle-site core updates (D=6144) via "
                               "accumulating_swipe(method='ridge_cholesky', orthonormalize=True) incl. per-update "
                               "validation pass; CG variant under variants.cg",
                   "sites": N_SITES, "bond": BOND, "degree": DEGREE, "train_samples": S_train, "D": D,
                   "batch_size": 512, "sharding": f"samples/{world}", "val_rmse": val_rmse},
        "roofline": {"kernel": "gram_kernel<true>", "bound": "mfma", "achieved": achieved,
                     "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                     "traffic": traffic, "launches": len(gram_ms), "launches_called_off": called_off, "avg_launch_ms": avg_ms,
                     "flops_per_launch": gram_flops, "feature_pair_classes": NX,
                     "dense_equivalent_tflops": 2.0 * S_local * D * D / (avg_ms * 1e-3) * 1e-12},
    }
    if rank == 0 and not sharded:
        probe = dgemm_probe(device)
        probe["gram_frac_of_dgemm"] = achieved / probe["tflops"]
        out["roofline"]["dgemm_probe"] = probe
    if rank == 0 and not sharded and not args.no_variants:
        sweep = sweep_variant(device, F, y, F_val, y_val)
        out["sweep_updates_per_sec"] = sweep["core_updates_per_sec"]     # whole first swipe of fit(): 31 updates, edge sites incl.
        out["variants"] = {"cg": cg_variant(device, F, y, steps=10), "sweep": sweep,
                           "lanczos": lanczos_variant(device, F, y)}
    if rank == 0 and not sharded and not args.no_cpu_baseline:
        out["parity"], out["cpu_baseline"] = parity_and_cpu_baseline(device, F, y)
    if not args.no_variants:
        # image-shaped conv tensor trains: on one GPU the three variants; with N > 1 every rank takes its sample
        # shard of the cfg-5 model (what CIFAR_lanczos_swipe.py would run on the node), rank 0 reports
        del F, y, F_val, y_val, layer, tn
        torch.cuda.empty_cache()
        f64, f32 = torch.float64, torch.float32
        v = out.setdefault("variants", {})
        if not sharded:
            v["mnist_lanczos"] = conv_lanczos_variant(
                device, 54000, 50, 17, 24, f64, -1, 1e-9,
                what="cfg4 shape: 3 carriages, bond 24, 50 patches x 17 pixels, 9 logits, 54000 samples, fp64 "
                     "(image_convolution_lanczos_MNIST.py:29-32,76-92), no convolution bond")
            v["mnist_lanczos_cb4"] = conv_lanczos_variant(
                device, 54000, 50, 17, 10, f64, 4, 1e-9,
                what="the MNIST script's own model: bond 10, convolution bond 4 (image_convolution_lanczos_MNIST.py:76)")
        v["cifar_lanczos_f32"] = conv_lanczos_variant(
            device, 50000, 65, 49, 48, f32, -1, 2e-5, shard=shard if rccl_one else sharded,
            what="cfg5 shape: 3 carriages, bond 48, 65 patches x 49 pixels, 9 logits, 50000 samples, float32 nodes and "
                 f"data (CIFAR_lanczos_swipe.py:35-38,68-108); samples sharded over {world} rank(s)")
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
