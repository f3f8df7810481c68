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
        # mean of per-batch means from per-batch partial sums (loss is (S, C) here), several batch sizes
        # incl. ones that straddle the rank boundary, a single batch and a ragged last batch
        for bsz in (bs, 7, S, 1000, 101):
            ref = sum(loss[b0:b0 + bsz].mean().item() for b0 in range(0, S, bsz)) / ((S + bsz - 1) // bsz)
            got = sh.batch_mean(loss[lo:hi].contiguous(), S, bsz).item()
            ok &= abs(ref - got) < 1e-14
        ok &= sh.all_true(True, "cpu") and not sh.all_true(sh.rank != 0, "cpu")
        v = torch.full((3,), float(sh.rank + 1))
        sh.broadcast_(v)
        ok &= bool(torch.equal(v, torch.ones(3)))
        # feature-pair classes common to all ranks, and the packed all-reduce of a classed Gram
        ok &= sh.common_pair_classes([0, 0, 0, 3] if sh.rank == 0 else [0, 0, 2, 3], 4, "cpu") == [0, 0, 2, 3]
        ok &= sh.common_pair_classes([0, 1, 1, 3] if sh.rank == 0 else None, 4, "cpu") is None
        ok &= sh.common_pair_classes([0, 1, 1, 1], 4, "cpu") == [0, 1, 1, 1]

        class FakeWS:
            mppad, pair_rep = 6, [0, 1, 1, 3, 0]
        G = torch.arange(6 * 5 * 7, dtype=torch.float64).reshape(6, 5, 7) * (sh.rank + 1)
        G[:, 2], G[:, 4] = G[:, 1], G[:, 0]          # members equal their representative, as the kernel leaves them
        FakeWS.compressed = G.reshape(-1).clone()
        want = FakeWS.compressed.clone()
        sh.allreduce_sum_(want)
        bvec = torch.arange(5, dtype=torch.float64) * (sh.rank + 2)
        bwant = bvec.clone()
        sh.allreduce_sum_(bwant)
        sh.allreduce_gram_(FakeWS, bvec)                 # the right-hand side rides in the same message
        ok &= bool(torch.equal(FakeWS.compressed, want)) and bool(torch.equal(bvec, bwant))

        class PlainWS:                                   # no classes: the workspace itself is the message
            mppad, pair_rep = 6, None
        PlainWS.buf = torch.zeros(6 * 5 * 7 + 40, dtype=torch.float64)
        PlainWS.buf[:210] = torch.arange(210, dtype=torch.float64) * (sh.rank + 1)
        PlainWS.compressed = PlainWS.buf[:210]
        want = PlainWS.compressed.clone()
        sh.allreduce_sum_(want)
        bvec = torch.arange(5, dtype=torch.float64) * (sh.rank + 2)
        sh.allreduce_gram_(PlainWS, bvec)
        ok &= bool(torch.equal(PlainWS.compressed, want)) and bool(torch.equal(bvec, bwant))
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

        xraw = torch.rand(S, generator=g) * 2 - 1
        xpoly = [torch.stack([xraw ** k for k in range(4)], dim=-1).contiguous().to(dev) for _ in range(N)]
        for which in ("gram", "gram_classed", "cg", "lanczos"):
            if which == "gram_classed":       # powers of one variable: feature-pair classes, packed all-reduce
                x_plain, x = x, xpoly
                c1, l1 = run(False, "gram")
                c2, l2 = run(True, "gram")
                x = x_plain
                for a, b in zip(c1, c2):
                    ok &= ((a - b).norm() / a.norm()).item() < 1e-9
                for t in c2:
                    ref = t.clone()
                    sh.broadcast_(ref)
                    ok &= bool(torch.equal(ref, t))
                continue
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
