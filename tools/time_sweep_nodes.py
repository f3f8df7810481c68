"""Scratch: wall time per core update along the first swipe of the headline model (device-synchronised after each)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tensornetworksfork_amd.tensor.bregman import SquareBregFunction
dev = torch.device("cuda:0")
torch.set_default_dtype(torch.float64)
F, y, F_val, y_val = bench.synthetic_problem(dev)
for rep in range(2):
    tn = bench.make_model(dev).tensor_network
    marks = []
    def cb(NS, node):
        torch.cuda.synchronize()
        marks.append((node.name, tuple(node.tensor.shape), time.perf_counter()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tn.accumulating_swipe(F, y, SquareBregFunction(), batch_size=512, lr=1.0, eps=1.0, eps_decay=0.5, orthonormalize=True,
                          method='ridge_cholesky', num_swipes=1, direction='l2r', block_callback=cb)
    torch.cuda.synchronize()
    if rep:
        prev = t0
        for name, shape, t in marks:
            print(f"{name:6s} {str(shape):22s} {1e3 * (t - prev):8.2f} ms")
            prev = t
        print("total", 1e3 * (time.perf_counter() - t0))
