"""Scratch: matrix-free CG on middle sites of the headline model, for rocprofv3."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scipy.sparse.linalg as spla
import bench
from tensornetworksfork_amd.tensor.bregman import SquareBregFunction
torch.set_default_dtype(torch.float64)
dev = torch.device("cuda:0")
F, y, Fv, yv = bench.synthetic_problem(dev)
layer = bench.make_model(dev)
tn = layer.tensor_network
allnodes = tn.train_nodes
tn.train_nodes = tn.main_nodes[2:3]
tn.scipy_swipe(F, y, SquareBregFunction(), spla.cg, batch_size=512, max_iter=100, tol=1e-6)
tn.train_nodes = tn.main_nodes[3:6]
torch.cuda.synchronize()
t0 = time.perf_counter()
tn.scipy_swipe(F, y, SquareBregFunction(), spla.cg, batch_size=512, max_iter=100, tol=1e-6)
torch.cuda.synchronize()
print("ms per update", (time.perf_counter() - t0) / 3 * 1e3)
