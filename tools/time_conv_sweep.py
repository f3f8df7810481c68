"""Scratch: one lanczos_swipe over the cfg-4-shaped conv tensor train (max_iter=12: the 17-dimensional filter nodes exhaust their Krylov space beyond that), wall time per node update."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd.tensor.layers import TensorConvolutionTrainLayer
from tensornetworksfork_amd.tensor.bregman import XEAutogradBregman
torch.set_default_dtype(torch.float64)
dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 54000
torch.manual_seed(0)
layer = TensorConvolutionTrainLayer(3, 24, 50, 17, output_shape=9, convolution_bond=-1).to(dev)
tn = layer.tensor_network
x = torch.rand(S, 50, 17, device=dev) * 0.5
y = torch.nn.functional.one_hot(torch.randint(0, 10, (S,), device=dev), 10).to(torch.float64)
times = []
def cb(NS, node):
    torch.cuda.synchronize()
    times.append((node.name, time.perf_counter()))
torch.manual_seed(1)
warm = TensorConvolutionTrainLayer(3, 24, 50, 17, output_shape=9, convolution_bond=-1).to(dev).tensor_network
warm.lanczos_swipe(x, y, XEAutogradBregman(), batch_size=512, num_swipes=1, max_iter=12)   # warm-up on another model
times.clear()
torch.manual_seed(1)
torch.cuda.synchronize(); t0 = time.perf_counter()
tn.lanczos_swipe(x, y, XEAutogradBregman(), batch_size=512, num_swipes=1, max_iter=12, block_callback=cb)
prev = t0
for name, t in times:
    print(name, "ms", round((t - prev) * 1e3, 2)); prev = t
print("sweep ms", round((prev - t0) * 1e3, 1))
