"""Worker for the multi-rank tests (launched by tests/test_distributed.py, one process per rank).

mode "cpu": partition + collective logic on CPU tensors under gloo; the local Gram / RHS of
            each shard comes from the oracle, the sharded sum must equal the full-data result.
mode "gpu": the real engine, every rank on cuda:0 under gloo (device tensors staged through
            the host); a sharded sweep must reproduce the single-process sweep.
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    from tensornetworksfork_amd.parallel import Shard
    from oracle import tt_oracle as O
    torch.set_default_dtype(torch.float64)
    sh = Shard()
    ok = True
    if mode == "cpu":
        S, C, rl, p, rr, bs = 203, 2, 3, 4, 5, 32
        g = torch.Generator().manual_seed(5)
        L, X, R = torch.rand(S, C, rl, generator=g), torch.rand(S, p, generator=g), torch.rand(S, rr, generator=g)
        yp, y = torch.rand(S, C, generator=g), torch.rand(S, C, generator=g)
        loss, grad, H = O.loss_mse(yp, y)
        A_full, b_full = O.local_gram_rhs(L, X, R, grad, H)
        lo, hi = sh.bounds(S)
        assert sum(sh.bounds(S, r)[1] - sh.bounds(S, r)[0] for r in range(sh.world)) == S
        A, b = O.local_gram_rhs(sh.rows(L), sh.rows(X), sh.rows(R), sh.rows(grad), sh.rows(H))
        sh.allreduce_sum_(A)
        sh.allreduce_sum_(b)
        ok &= bool(torch.allclose(A, A_full, rtol=1e-12, atol=1e-14)) and bool(torch.allclose(b, b_full, rtol=1e-12, atol=1e-14))
        full_loss = sh.gather_rows(loss[lo:hi].contiguous(), S)
        ok &= bool(torch.equal(full_loss, loss))
        ref = sum(loss[b0:b0 + bs].mean().item() for b0 in range(0, S, bs)) / ((S + bs - 1) // bs)
        got = sum(full_loss[b0:b0 + bs].mean().item() for b0 in range(0, S, bs)) / ((S + bs - 1) // bs)
        ok &= abs(ref - got) < 1e-15
        v = torch.full((3,), float(sh.rank + 1))
        sh.broadcast_(v)
        ok &= bool(torch.equal(v, torch.ones(3)))
    else:
        from tensornetworksfork_amd.tensor.layers import TensorTrainLayer
        from tensornetworksfork_amd.tensor.bregman import SquareBregFunction
        dev = "cuda:0"
        g = torch.Generator().manual_seed(9)
        N, S = 5, 301
        x = [(torch.rand(S, 4, generator=g) * 2 - 1).to(dev) for _ in range(N)]
        y = torch.randn(S, 1, generator=g).to(dev)

        def run(shard, which):
            layer = TensorTrainLayer(N, 6, 4, output_shape=1, seed=3).to(dev)
            tn = layer.tensor_network
            if shard:
                tn.distribute(sh)
            losses = []
            if which == "gram":
                tn.accumulating_swipe(x, y, SquareBregFunction(), batch_size=64, num_swipes=1, method="ridge_cholesky",
                                      eps=0.5, orthonormalize=True, loss_callback=lambda NS, n, l: losses.append(l))
            elif which == "lanczos":
                torch.manual_seed(100 + (sh.rank if shard else 0))   # ranks disagree on purpose: x0 is broadcast
                tn.lanczos_swipe(x, y, SquareBregFunction(), batch_size=64, num_swipes=1, max_iter=6,
                                 loss_callback=lambda l: losses.append(l))
            else:
                import scipy.sparse.linalg as spla
                tn.scipy_swipe(x, y, SquareBregFunction(), spla.cg, batch_size=64, num_swipes=1, max_iter=5,
                               loss_callback=lambda l: losses.append(l))
            return [n.tensor.clone() for n in tn.main_nodes], losses

        for which in ("gram", "cg", "lanczos"):
            if which == "lanczos":
                torch.manual_seed(100)
            c1, l1 = run(False, which) if which != "lanczos" else (None, None)
            if which == "lanczos":
                torch.manual_seed(100)
                c1, l1 = run(False, which)
            c2, l2 = run(True, which)
            for a, b in zip(c1, c2):
                ok &= ((a - b).norm() / a.norm()).item() < 1e-9
            ok &= all(abs(a - b) <= 1e-10 * abs(a) for a, b in zip(l1, l2))
            # every rank must hold the same cores bit for bit
            for t in c2:
                ref = t.clone()
                sh.broadcast_(ref)
                ok &= bool(torch.equal(ref, t))
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if sh.rank == 0:
        with open(out_path, "w") as f:
            f.write("OK" if flag.item() == 1.0 else "FAIL")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
