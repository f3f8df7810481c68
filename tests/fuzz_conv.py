"""Random conv tensor trains (carriages, bonds, patches, pixels, outputs, convolution bond, dtype) through the
identities that tie the kernels of every train node's route to each other (model linear in each node, b = J^T g
the adjoint of J, A = J^T H J symmetric and factoring) and one Lanczos step per node; see tests/test_fullsize_gpu.py.
    python tests/fuzz_conv.py [cases] [seed] [only this case]
Not collected by pytest (a development run); a failing configuration belongs in tests/test_conv_gpu.py."""
import random
import sys
import traceback
import torch
sys.path.insert(0, ".")
from tests.test_fullsize_gpu import _check_conv_identities



def _check_lanczos(layer, x, y, bf, max_iter):
    """The Galerkin property of the Lanczos step of every node (the quadratic model does not increase from the
    start vector) and one whole lanczos_swipe pass.  (The residual need not fall in so few iterations.)"""
    from tensornetworksfork_amd.tensor.solvers import lanczos_solve
    tn = layer.tensor_network
    f0 = tn.forward(x, to_tensor=True)
    _, g, H = bf.forward(f0, y)
    gen = torch.Generator(device=x.device).manual_seed(9)
    for node in tn.train_nodes:
        d, lp = tn._local(node)
        mv = lp.matvec_fn(tn._norm_h(H, d.S, lp.C))
        b = lp.rhs(g.to(torch.float64).contiguous())
        x0 = torch.randn(lp.shape, generator=gen, device=x.device, dtype=torch.float64)

        def q(z):
            return (0.5 * (z * mv(z)).sum() + (b * z).sum()).item()
        step = lanczos_solve(mv, -b, x0, max_iter, 1e-6)
        assert q(step) <= q(x0) + 1e-9 * abs(q(x0)), (node.name, "Galerkin", q(step), q(x0))
    losses = []
    assert tn.lanczos_swipe(x, y, bf, batch_size=512, num_swipes=1, lr=0.5, max_iter=max_iter,
                            loss_callback=lambda l: losses.append(l))
    assert len(losses) == len(tn.train_nodes) and all(l == l for l in losses), losses


NCASES = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ONLY = int(sys.argv[3]) if len(sys.argv) > 3 else None
dev = torch.device("cuda:0")
torch.set_default_dtype(torch.float64)
bad = 0
for case in range(NCASES):
    ncar = rnd.randint(2, 4)
    r = rnd.choice([1, 2, 3, 5, 8, 10, 16, 17, 24, 33, 40, 48, 64, 65, 100])
    patches = rnd.choice([3, 7, 15, 16, 17, 20, 33, 50, 65, 70])
    pixels = rnd.choice([2, 5, 9, 17, 26, 49])
    C = rnd.choice([1, 2, 9, 10])
    CB = rnd.choice([-1, -1, 1, 2, 4])
    dtype = rnd.choice([torch.float64, torch.float64, torch.float32])
    S = rnd.choice([100, 1000, 3001, 20011])
    if ONLY is not None and case != ONLY:
        continue
    print("case", case, dict(carriages=ncar, bond=r, patches=patches, pixels=pixels, C=C, CB=CB, dtype=str(dtype), S=S), flush=True)
    from tensornetworksfork_amd.tensor.layers import TensorConvolutionTrainLayer
    from tensornetworksfork_amd.tensor.bregman import XEAutogradBregman, SquareBregFunction
    try:
        torch.set_default_dtype(dtype)
        try:
            torch.manual_seed(case)
            layer = TensorConvolutionTrainLayer(num_carriages=ncar, bond_dim=r, num_patches=patches, patch_pixels=pixels,
                                                output_shape=C, convolution_bond=CB).to(dev)
        finally:
            torch.set_default_dtype(torch.float64)
        g = torch.Generator(device=dev).manual_seed(case)
        x = torch.randn(S, patches, pixels, generator=g, dtype=dtype, device=dev)
        x[:, -1, :] = 0.0
        x[:, :, -1] = 0.0
        x[:, -1, -1] = 1.0
        tn = layer.tensor_network
        with torch.inference_mode():
            w = 1.0 / max(tn.forward(x[:64].contiguous(), to_tensor=True).std().item(), 1e-30)
        if rnd.random() < 0.5:
            lab = torch.randint(0, C + 1, (S,), generator=g, device=dev)
            y = torch.nn.functional.one_hot(lab, C + 1).to(torch.float64)
            bf = XEAutogradBregman(w=w)
        else:
            y = torch.randn(S, C, generator=g, dtype=torch.float64, device=dev)
            bf = SquareBregFunction()
        _check_conv_identities(layer, x, y, bf, 1e-9 if dtype == torch.float64 else 5e-5)
        _check_lanczos(layer, x, y, bf, max_iter=4)
        print("   ok", flush=True)
    except AssertionError as e:
        bad += 1
        print("   ASSERT", e, flush=True)
    except Exception as e:      # noqa: BLE001 - a development tool reports whatever the model raises
        bad += 1
        print("   RAISED", type(e).__name__, e, flush=True)
        traceback.print_exc(limit=6)
print("done:", bad, "bad of", NCASES)
