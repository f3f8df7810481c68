"""Scratch: cfg-4-shaped conv tensor train (3 carriages, bond 24, 50 patches x 17 pixels, 9 outputs):
time the matrix-free operator of every train node (what one Lanczos / CG iteration costs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd.tensor.layers import TensorConvolutionTrainLayer
from tensornetworksfork_amd.tensor.bregman import XEAutogradBregman
torch.set_default_dtype(torch.float64)
dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
CB = int(sys.argv[2]) if len(sys.argv) > 2 else -1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
torch.manual_seed(0)
layer = TensorConvolutionTrainLayer(3, 24, 50, 17, output_shape=9, convolution_bond=CB).to(dev)
tn = layer.tensor_network
x = torch.rand(S, 50, 17, device=dev) * 0.5
y = torch.nn.functional.one_hot(torch.randint(0, 10, (S,), device=dev), 10).to(torch.float64)
out = tn.forward(x, to_tensor=True)
loss, g, H = XEAutogradBregman().forward(out, y)
d = tn._current
Hn = tn._norm_h(H, d.S, 9)
for node in tn.train_nodes:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lp = tn._engine.local(d, node)
    b = lp.rhs(g.contiguous())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    mv = lp.matvec_fn(Hn)
    v = torch.randn(lp.shape, device=dev)
    mv(v)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(reps):
        w = mv(v)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"{node.name}: D={lp.D} setup+rhs {(t1 - t0) * 1e3:.2f} ms, matvec {(t3 - t2) / reps * 1e3:.3f} ms", flush=True)
