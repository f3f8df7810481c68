import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworksfork_amd.tensor.layers import TensorConvolutionTrainLayer
from tensornetworksfork_amd.tensor.bregman import XEAutogradBregman
torch.set_default_dtype(torch.float64)
dev = torch.device("cuda:0")
S = 300
torch.manual_seed(0)
layer = TensorConvolutionTrainLayer(3, 24, 50, 17, output_shape=9, convolution_bond=-1).to(dev)
tn = layer.tensor_network
x = torch.rand(S, 50, 17, device=dev) * 0.5
y = torch.nn.functional.one_hot(torch.randint(0, 10, (S,), device=dev), 10).to(torch.float64)
C1, A1, C2, A2, C3, A3 = [n.tensor for n in tn.train_nodes]
g1, g2, g3 = x @ C1, x @ C2, x @ C3
ref = torch.einsum('sp,cpr,sq,rqt,su,tu->sc', g1, A1, g2, A2, g3, A3)
out = tn.forward(x, to_tensor=True)
print("forward rel err", ((out - ref).norm() / ref.norm()).item(), ref.abs().max().item())
loss, g, H = XEAutogradBregman().forward(out, y)
print("loss", loss.mean().item(), "H finite", torch.isfinite(H).all().item(), H.abs().max().item())
for node in tn.train_nodes:
    A, b = tn.get_A_b(node, g, H)
    D = b.numel()
    v = torch.randn_like(b)
    Av = tn.matvec(node, v, H)
    Av2 = (A.reshape(D, D) @ v.reshape(D)).reshape(b.shape)
    print(node.name, tuple(b.shape), "matvec vs A rel err", ((Av - Av2).norm() / Av2.norm()).item(), "|A|", A.abs().max().item())
