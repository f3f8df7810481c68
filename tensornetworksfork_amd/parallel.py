"""Sample-sharded core updates: one process per GPU, torch.distributed (RCCL over xGMI).

The local normal equations are sums over independent samples
(reference: tensor/network.py:212,215,463-468), so each rank builds the Gram
matrix and right-hand side of its own contiguous block of samples and one
all-reduce(sum) per core update makes them global; every rank then solves the
same system and keeps bit-identical cores.  Only the symmetry-compressed Gram
(6.4x smaller than the dense matrix at bond 32) crosses the links.
"""
import torch
import torch.distributed as dist


class Shard:
    """Contiguous row block [lo, hi) of this rank out of S rows (sizes differ by at most one)."""

    def __init__(self, group=None, always_collective=False):
        """always_collective: issue every collective even in a group of one rank (where each is the
        identity), so that one GPU can walk the whole RCCL call path of a sharded sweep."""
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self._alone = self.world == 1 and not always_collective
        self._index = {}   # (device, indices) -> device index tensor: built once, no host-to-device copy per update

    def bounds(self, S, rank=None):
        r = self.rank if rank is None else rank
        base, extra = divmod(S, self.world)
        lo = r * base + min(r, extra)
        return lo, lo + base + (1 if r < extra else 0)

    def rows(self, t):
        lo, hi = self.bounds(t.shape[0])
        return t[lo:hi]

    def _index_tensor(self, values, device):
        key = (device, tuple(values))
        t = self._index.get(key)
        if t is None:
            if len(self._index) > 64:
                self._index.clear()
            t = self._index[key] = torch.tensor(list(values), dtype=torch.int64, device=device)
        return t

    def allreduce_sum_(self, t):
        """In-place sum over ranks.  RCCL handles device tensors; under gloo (CPU
        rehearsal of the multi-rank path) device tensors are staged through the host."""
        if self._alone:
            return t
        if t.is_cuda and self.backend == "gloo":
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        elif t.is_contiguous():
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        else:   # RCCL takes dense tensors only (a permuted core, a view of a workspace)
            c = t.contiguous()
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(c)
        return t

    def common_pair_classes(self, rep, n, device):
        """Feature-pair classes that hold on EVERY rank: two pairs stay in one class only if every
        rank's own samples put them in one (each rank detects classes on its shard).  rep: this
        rank's class representatives (None = no classes); returns the common ones in the same form."""
        return self.common_pair_classes_many([rep], n, device)[0]

    def common_pair_classes_many(self, reps, n, device):
        """common_pair_classes for several feature matrices with `n` pairs each in ONE all-gather (all sites of a
        data set when the first Gram is asked for: a sweep does not stop at every new site for the exchange)."""
        local = torch.tensor([list(range(n)) if rep is None else [int(v) for v in rep] for rep in reps], dtype=torch.int64)
        if not self._alone:
            t = local if self.backend == "gloo" else local.to(device)
            parts = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(parts, t, group=self.group)
            parts = [q.tolist() for q in parts]
        else:
            parts = [local.tolist()]
        out = []
        for m in range(len(reps)):
            first, common = {}, []
            for b, k in enumerate(zip(*[q[m] for q in parts])):
                common.append(first.setdefault(k, b))
            out.append(None if common == list(range(n)) else common)
        return out

    def allreduce_gram_(self, ws, b=None):
        """Sum of the compressed Gram (and, in the same message, of the right-hand side `b`) over
        ranks: ONE collective per core update.  With feature-pair classes only the representative
        slices G[:, rep, :] travel (packed, summed, copied back to every member of their class): at
        the headline shape 24.5 MB instead of 46.8 MB.  The classes must be the common ones
        (common_pair_classes), or ranks would disagree about the message size.  Without classes
        the message is the workspace itself: `b` rides in the `tail` doubles the workspace keeps
        right behind the Gram (a workspace without such a tail is copied into a fresh message)."""
        G = ws.compressed
        rep = getattr(ws, "pair_rep", None)
        if self._alone:
            return G
        nb = 0 if b is None else b.numel()
        if rep is None:
            # Whether `b` rides behind the Gram in the workspace must not depend on anything a rank measures on
            # its own shard (the scratch size does: shards differ by one sample), or ranks would issue different
            # collectives.  `tail` is set from the site's extents alone, which every rank shares.
            if not nb:
                self.allreduce_sum_(G)
                return G
            buf = getattr(ws, "buf", None)
            if buf is not None and getattr(ws, "tail", 0) >= nb and buf.data_ptr() == G.data_ptr():
                msg = buf[:G.numel() + nb]
                msg[G.numel():].copy_(b.reshape(-1))
                self.allreduce_sum_(msg)
            else:
                msg = torch.empty(G.numel() + nb, dtype=G.dtype, device=G.device)
                msg[:G.numel()].copy_(G.reshape(-1))
                msg[G.numel():].copy_(b.reshape(-1))
                self.allreduce_sum_(msg)
                G.reshape(-1).copy_(msg[:G.numel()])
            b.copy_(msg[G.numel():].reshape(b.shape))
            return G
        nx = len(rep)
        reps = sorted(set(int(v) for v in rep))
        pos = {r: i for i, r in enumerate(reps)}
        G3 = G.view(-1, nx, G.numel() // (ws.mppad * nx))
        npk = G3.shape[0] * len(reps) * G3.shape[2]
        msg = torch.empty(npk + nb, dtype=G.dtype, device=G.device)
        packed = msg[:npk].view(G3.shape[0], len(reps), G3.shape[2])
        torch.index_select(G3, 1, self._index_tensor(reps, G.device), out=packed)
        if nb:
            msg[npk:].copy_(b.reshape(-1))
        self.allreduce_sum_(msg)
        torch.index_select(packed, 1, self._index_tensor([pos[int(v)] for v in rep], G.device), out=G3)
        if nb:
            b.copy_(msg[npk:].reshape(b.shape))
        return G

    def broadcast_(self, t, src=0):
        """Rank `src`'s values on every rank (start vectors drawn from a random generator)."""
        if self._alone:
            return t
        if t.is_cuda and self.backend == "gloo":
            h = t.cpu()
            dist.broadcast(h, src=src, group=self.group)
            t.copy_(h)
        elif t.is_contiguous():
            dist.broadcast(t, src=src, group=self.group)
        else:
            c = t.contiguous()
            dist.broadcast(c, src=src, group=self.group)
            t.copy_(c)
        return t

    def all_true(self, flag, device):
        """Logical AND of a host flag over the ranks (one tiny all-reduce)."""
        if self._alone:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cpu" if self.backend == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(t.item())

    def batch_mean(self, local, S, batch):
        """Mean over the mini-batches (of `batch` consecutive global samples) of the per-batch mean
        of per-sample values, from this rank's rows `local` (n, k): every rank sums its part of each
        batch it touches, ONE all-reduce of the per-batch sums (number-of-batches doubles) makes
        them global.  0-d tensor, the same on every rank."""
        lo, hi = self.bounds(S)
        n, k = local.shape[0], local[0].numel() if local.shape[0] else 1
        nb = (S + batch - 1) // batch
        sums = torch.zeros(nb, dtype=local.dtype, device=local.device)
        if n:
            b0, off = lo // batch, lo % batch
            nbl = (off + n + batch - 1) // batch
            pad = torch.zeros((nbl * batch, k), dtype=local.dtype, device=local.device)
            pad[off:off + n] = local.reshape(n, k)
            sums[b0:b0 + nbl] = pad.reshape(nbl, batch * k).sum(dim=1)
        self.allreduce_sum_(sums)
        counts = torch.full((nb,), float(batch * k), dtype=local.dtype, device=local.device)
        counts[-1] = float((S - (nb - 1) * batch) * k)
        return (sums / counts).mean()


def default_shard():
    """Shard over the default process group, or None when not running distributed."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return Shard()
    return None
