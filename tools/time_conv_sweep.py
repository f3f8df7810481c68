"""Scratch: one lanczos_swipe over the cfg-4-shaped conv tensor train (max_iter=12: the 17-dimensional filter nodes exhaust their Krylov space beyond that), wall time per node update."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd.tensor.layers import TensorConvolutionTrainLayer
from tensornetworksfork_amd.tensor.bregman import XEAutogradBregman
cfg = sys.argv[1] if len(sys.argv) > 1 else "mnist"
# mnist: image_convolution_lanczos_MNIST.py shapes, fp64; cifar: CIFAR_lanczos_swipe.py:35-38 shapes, float32 default dtype
S, R_, P_, PX_, DT = (54000, 24, 50, 17, torch.float64) if cfg == "mnist" else (50000, 48, 65, 49, torch.float32)
if len(sys.argv) > 2:      # bond override (image_convolution_lanczos_CIFAR10.py itself uses 100)
    R_ = int(sys.argv[2])
CB_ = int(sys.argv[3]) if len(sys.argv) > 3 else -1   # convolution bond (image_convolution_lanczos_MNIST.py: r = 10, CB = 4)
torch.set_default_dtype(DT)
dev = torch.device("cuda:0")
torch.manual_seed(0)
layer = TensorConvolutionTrainLayer(3, R_, P_, PX_, output_shape=9, convolution_bond=CB_).to(dev)
tn = layer.tensor_network
x = torch.rand(S, P_, PX_, device=dev) * 0.5
y = torch.nn.functional.one_hot(torch.randint(0, 10, (S,), device=dev), 10).to(torch.float64)
times = []
def cb(NS, node):
    torch.cuda.synchronize()
    times.append((node.name, time.perf_counter()))
with torch.inference_mode():      # the scripts scale the loss by the spread of the first predictions
    wgt = 1.0 / layer(x[:64]).std().item()
torch.manual_seed(1)
warm = TensorConvolutionTrainLayer(3, R_, P_, PX_, output_shape=9, convolution_bond=CB_).to(dev).tensor_network
warm.lanczos_swipe(x, y, XEAutogradBregman(w=wgt), batch_size=512, num_swipes=1, max_iter=12)   # warm-up on another model
times.clear()
torch.manual_seed(1)
torch.cuda.synchronize(); t0 = time.perf_counter()
tn.lanczos_swipe(x, y, XEAutogradBregman(w=wgt), batch_size=512, num_swipes=1, max_iter=12, block_callback=cb)
prev = t0
for name, t in times:
    print(name, "ms", round((t - prev) * 1e3, 2)); prev = t
print(cfg, "sweep ms", round((prev - t0) * 1e3, 1), "(6 node updates, 12 Lanczos iterations each)")
