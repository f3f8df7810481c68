"""Scratch: one Lanczos update at the headline shape with and without the block-queued iterations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tensornetworksfork_amd.tensor import engine
from tensornetworksfork_amd.tensor.bregman import SquareBregFunction
dev = torch.device("cuda:0")
torch.set_default_dtype(torch.float64)
F, y, F_val, y_val = bench.synthetic_problem(dev)
orig = engine.LocalProblem.lanczos_fused
for label, fn in (("fused", orig), ("plain", lambda self, H: None), ("fused", orig)):
    engine.LocalProblem.lanczos_fused = fn
    tn = bench.make_model(dev).tensor_network
    nodes = tn.train_nodes
    tn.train_nodes = tn.main_nodes[2:3]
    torch.manual_seed(0)
    tn.lanczos_swipe(F, y, SquareBregFunction(), batch_size=512, max_iter=100, tol=1e-6)
    tn.train_nodes = tn.main_nodes[3:11]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tn.lanczos_swipe(F, y, SquareBregFunction(), batch_size=512, max_iter=100, tol=1e-6)
    torch.cuda.synchronize()
    print(label, "ms per update", (time.perf_counter() - t0) / 8 * 1e3, flush=True)
