"""Scratch: two ranks on one GPU through gloo, one sharded core update with diagnostics."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
from tensornetworksfork_amd.tensor.bregman import SquareBregFunction
torch.set_default_dtype(torch.float64)
dist.init_process_group("gloo")
rank = dist.get_rank()
dev = torch.device("cuda:0")
F, y, Fv, yv = bench.synthetic_problem(dev)
layer = bench.make_model(dev)
tn = layer.tensor_network
tn.distribute("auto")
_ar = tn._shard.allreduce_sum_
def ar(t):
    pre = float(t.double().abs().sum())
    r = _ar(t)
    torch.cuda.synchronize()
    print(rank, "allreduce", tuple(t.shape), "pre", pre, "post", float(t.abs().sum()), flush=True)
    return r
tn._shard.allreduce_sum_ = ar
from tensornetworksfork_amd import ops
_cs = ops.chol_solve_gram
def cs(ws, b, w=None, scale=None, ridge=0.0):
    x, info = _cs(ws, b, w, scale, ridge)
    torch.cuda.synchronize()
    print(rank, "solve: scale", float(scale), "ridge", ridge, "|b|", float(b.norm()), "|w|", float(w.norm()), "|x|", float(x.norm()), "info", int(info), flush=True)
    return x, info
ops.chol_solve_gram = cs
mid = tn.main_nodes[2:bench.N_SITES - 2]
cnt = [0]
def hook():
    cnt[0] += 1
    p = tn.forward_batch(Fv, 512)
    print(rank, "update", cnt[0], "val pred shape", tuple(p.shape), "rmse", float(((p - yv) ** 2).mean().sqrt()), flush=True)
    return cnt[0] >= 3
ok = tn.accumulating_swipe(F, y, SquareBregFunction(), node_order=(mid, mid), batch_size=512, lr=1.0, eps=1.0, eps_decay=0.5,
                           convergence_criterion=hook, orthonormalize=True, method='ridge_cholesky', num_swipes=10, direction='l2r',
                           verbose=1)
print(rank, "returned", ok, cnt[0], flush=True)
dist.barrier()
