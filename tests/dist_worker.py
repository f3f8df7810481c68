"""Worker for the multi-rank tests (launched by tests/test_distributed.py, one process per rank).

mode "cpu": partition + collective logic on CPU tensors under gloo; the local Gram / RHS of
            each shard comes from the oracle, the sharded sum must equal the full-data result.
mode "cpu_engine": the package's host logic over the torch stand-ins of the kernels
            (tests/_kernel_standins.py) on CPU tensors: sharded conv / CPD / sum-of-networks sweeps.
mode "gpu": the real engine, every rank on cuda:0 under gloo (device tensors staged through
            the host); a sharded sweep must reproduce the single-process sweep.
mode "rccl": the cases of "gpu" in a group of ONE rank under the nccl backend (RCCL) with every
            collective issued all the same: device tensors go to RCCL on the sweep's own streams, as
            they do with one rank per GPU (two ranks cannot share a GPU under RCCL).
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _same_on_all_ranks(sh, tensors):
    ok = True
    for t in tensors:
        ref = t.clone()
        sh.broadcast_(ref)
        ok &= bool(torch.equal(ref, t))
    return ok


def _more_gpu_cases(sh, dev):
    """What a sample-sharded run of the image configurations and the estimators' other model types
    goes through: conv tensor trains (no convolution bond / bond 2, float64 / float32) under the
    matrix-free sweeps, a CPD layer and a type-I sum of networks under accumulating_swipe (dense
    Gram: [A | b] in one message), and shards so small that their workspaces differ in size.
    Each case: sharded == single process, and bit-identical nodes on every rank."""
    import scipy.sparse.linalg as spla
    from tensornetworksfork_amd.tensor import layers as Lm, network as Nm
    from tensornetworksfork_amd.tensor.bregman import SquareBregFunction, XEAutogradBregman
    from tests._net_cases import net_builders
    ok = True
    g = torch.Generator().manual_seed(21)

    def compare(make, run, tol, what):
        nonlocal ok
        outs = []
        for shard in (False, True):
            layer = make().to(dev)
            tn = layer.tensor_network
            if shard:
                tn.distribute(sh)
            torch.manual_seed(300)
            run(tn)
            outs.append([n.tensor.clone() for n in tn.train_nodes])
        for a, b in zip(*outs):
            err = ((a.double() - b.double()).norm() / a.double().norm()).item()
            if not err < tol:
                print(f"[rank {sh.rank}] {what}: sharded differs from single process by {err:.3e}", flush=True)
                ok = False
        if not _same_on_all_ranks(sh, outs[1]):
            print(f"[rank {sh.rank}] {what}: nodes differ between ranks", flush=True)
            ok = False

    # conv tensor trains, 17 patches (the features-first route of the image configurations)
    S, C = 203, 3
    for dt, cb, sweep, tol in ((torch.float64, -1, "lanczos", 1e-9), (torch.float64, 2, "minres", 1e-9),
                               (torch.float64, 2, "ridge", 1e-9), (torch.float32, -1, "lanczos", 2e-3),
                               (torch.float32, -1, "ridge", 1e-4)):
        x = (torch.rand(S, 17, 5, generator=g) * 0.5).to(dt).to(dev)
        y = torch.nn.functional.one_hot(torch.randint(0, C + 1, (S,), generator=g), C + 1).to(torch.float64).to(dev)

        def make(dt=dt, cb=cb):
            torch.set_default_dtype(dt)
            try:
                torch.manual_seed(11)
                return Lm.TensorConvolutionTrainLayer(3, 4, 17, 5, output_shape=C, convolution_bond=cb)
            finally:
                torch.set_default_dtype(torch.float64)

        def run(tn, sweep=sweep, x=x, y=y):
            bf = XEAutogradBregman(w=1.0)
            if sweep == "lanczos":
                tn.lanczos_swipe(x, y, bf, batch_size=64, num_swipes=1, max_iter=3, timeout=1e6)   # (a timeout is agreed on by the ranks)
            elif sweep == "minres":
                tn.scipy_swipe(x, y, bf, spla.minres, batch_size=64, num_swipes=1, max_iter=4)
            else:
                tn.accumulating_swipe(x, y, bf, batch_size=64, num_swipes=1, method="ridge_cholesky", eps=0.1, timeout=1e6)
        compare(make, run, tol, f"conv {dt} cb={cb} {sweep}")

    # CPD layer and type-I sums under the Gauss-Newton sweep
    builders = net_builders(Lm, Nm)
    for name, kind, Cn in (("cpd_c1", "square", 1), ("cpd_c2", "xe", 2), ("sum_tt_type1_c1", "square", 1),
                           ("sum_cpd_type1_c2", "xe", 2), ("ttlinear_c2", "xe", 2)):
        S = 157
        x = (torch.rand(S, 6, generator=g) * 2 - 1).to(dev)
        if kind == "xe":
            y = torch.nn.functional.one_hot(torch.randint(0, Cn + 1, (S,), generator=g), Cn + 1).to(torch.float64).to(dev)
        else:
            y = torch.randn(S, Cn, generator=g).to(dev)

        def run(tn, kind=kind, x=x, y=y):
            bf = XEAutogradBregman(w=1.0) if kind == "xe" else SquareBregFunction()
            tn.accumulating_swipe(x, y, bf, batch_size=50, num_swipes=1, method="ridge_cholesky", eps=0.1)
        compare(builders[name], run, 1e-9, name)

    # shards of 18 and 19 samples with generic features: the Gram workspaces of the two ranks differ in
    # size, the message they exchange must not
    S = 37
    xs = [(torch.rand(S, 6, generator=g) * 2 - 1).to(dev) for _ in range(4)]
    y = torch.randn(S, 1, generator=g).to(dev)

    def run(tn):
        tn.accumulating_swipe(xs, y, SquareBregFunction(), batch_size=16, num_swipes=1, method="ridge_cholesky", eps=0.5)
    compare(lambda: Lm.TensorTrainLayer(4, 16, 6, output_shape=1, constrict_bond=False, seed=5), run, 1e-9, "tiny uneven shards")
    return ok


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    if mode == "rccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    from tensornetworksfork_amd.parallel import Shard
    from oracle import tt_oracle as O
    torch.set_default_dtype(torch.float64)
    sh = Shard(always_collective=mode == "rccl")
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
        ok &= sh.common_pair_classes_many([[0, 0, 0, 3] if sh.rank == 0 else [0, 0, 2, 3], None, [0, 1, 1, 1]], 4,
                                          "cpu") == [[0, 0, 2, 3], None, [0, 1, 1, 1]]

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
        sh.allreduce_gram_(PlainWS, bvec)                # no `tail`: packed into a fresh message
        ok &= bool(torch.equal(PlainWS.compressed, want)) and bool(torch.equal(bvec, bwant))
        PlainWS.tail = 5                                 # room behind the Gram on every rank: sent in place
        PlainWS.buf[:210] = torch.arange(210, dtype=torch.float64) * (sh.rank + 1)
        bvec = torch.arange(5, dtype=torch.float64) * (sh.rank + 2)
        sh.allreduce_gram_(PlainWS, bvec)
        ok &= bool(torch.equal(PlainWS.compressed, want)) and bool(torch.equal(bvec, bwant))
        ok &= bool(torch.equal(PlainWS.buf[210:215], bwant))
    elif mode == "cpu_engine":
        # the real host logic (engines, sweeps, sharding) over the torch stand-ins of the kernels, on CPU tensors
        from tensornetworksfork_amd import ops
        from tests import _kernel_standins

        class _Patch:
            @staticmethod
            def setattr(obj, name, value):
                setattr(obj, name, value)
        _kernel_standins.install(_Patch)
        assert ops.gram is _kernel_standins.gram
        ok &= _more_gpu_cases(sh, "cpu")
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
        ok &= _more_gpu_cases(sh, dev)
    flag = torch.tensor([1.0 if ok else 0.0], device="cuda:0" if mode == "rccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if sh.rank == 0:
        with open(out_path, "w") as f:
            f.write("OK" if flag.item() == 1.0 else "FAIL")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
