"""Engines for the two non-chain models of the reference: the canonical polyadic network
(CPDNetwork, tensor/network.py:935-986) and the sum of networks (SumOfNetworks,
tensor/network.py:988-1060).  Both reduce every node's normal equations to the same
Khatri-Rao local problems the tensor-train engine produces, so the same kernels run them."""
import torch

from .. import ops
from .engine import EngineBase, LocalProblem, _changed

F64 = torch.float64


# =====================================================================================  CPD
class _Factor:
    __slots__ = ("A", "xin", "order", "out_labels", "rank_label", "sizes", "C", "R", "p")


class _CPDData:
    def __init__(self, raw, xs, N):
        self.raw = raw
        self.xs = xs
        self.S = xs[0].shape[0]
        self.S_total = self.S
        self.sharded = False
        self.T = [None] * N          # per-factor tables (S, C_i, R)
        self.seen = [None] * N

    def matches(self, raw):
        return len(raw) == len(self.raw) and all(a is b for a, b in zip(raw, self.raw))


class CPDEngine(EngineBase):
    """out[s,o] = sum_b prod_i T_i[s,b(,o)],  T_i[s,b] = sum_p X_i[s,p] A_i[b,p]
    (tensor/layers.py:1549-1625).  The Jacobian of factor k is the Khatri-Rao term
    W_k[s,c,b] (x) X_k[s,p] with W_k the product of the other factors' tables."""

    def __init__(self, net):
        super().__init__(net)
        self.cols = None
        try:
            self.cols = self._plan()
        except ValueError as e:
            self.error = str(e)

    def _plan(self):
        net = self.net
        s = net.sample_dim
        outs = [lab for lab in (net.output_labels or ()) if lab != s]
        if len(net.input_nodes) != len(net.main_nodes) or len(net.nodes) != 2 * len(net.main_nodes):
            raise ValueError("expected one input node per factor and no other nodes")
        facs, carrier = [], None
        for i, node in enumerate(net.main_nodes):
            f = _Factor()
            f.A = node
            if len(node.connections) != 1:
                raise ValueError(f"{node.name}: expected exactly one edge (to its input)")
            plab, xin = next(iter(node.connections.items()))
            if xin not in net.input_nodes or list(xin.dim_labels) != [s, plab]:
                raise ValueError(f"{node.name}: expected an input node labelled [sample, {plab}]")
            f.xin = net.input_nodes.index(xin)
            f.out_labels = [lab for lab in node.dim_labels if lab in outs]
            rest = [lab for lab in node.dim_labels if lab != plab and lab not in f.out_labels]
            if len(rest) > 1:
                raise ValueError(f"{node.name}: unexpected labels {node.dim_labels}")
            f.rank_label = rest[0] if rest else None
            f.order = f.out_labels + rest + [plab]
            if f.out_labels:
                if carrier is not None:
                    raise ValueError("output labels on more than one factor")
                carrier = i
            facs.append(f)
        ranks = {f.rank_label for f in facs if f.rank_label is not None}
        if len(ranks) > 1:
            raise ValueError("factors do not share one rank label")
        if any(t not in net.main_nodes for t in net.train_nodes):
            raise ValueError("train nodes outside the factors")
        self.carrier = carrier
        return facs

    def require(self):
        if self.cols is None:
            raise NotImplementedError(f"CPD topology not recognised: {self.error}")
        return self.cols

    # -- layout
    def out_labels(self):
        return [] if self.carrier is None else list(self.cols[self.carrier].out_labels)

    def out_sizes(self):
        if self.carrier is None:
            return []
        f = self.cols[self.carrier]
        return [f.A.dim_size(lab) for lab in f.out_labels]

    def canon(self, i):
        """(C_i, R, p) view of factor i."""
        f = self.cols[i]
        node = f.A
        perm = [node.dim_labels.index(lab) for lab in f.order]
        t = node.tensor.permute(perm) if perm != list(range(len(perm))) else node.tensor
        f.sizes = list(t.shape)
        no = len(f.out_labels)
        C = 1
        for n in f.sizes[:no]:
            C *= n
        f.C, f.p = C, f.sizes[-1]
        f.R = f.sizes[no] if f.rank_label is not None else 1
        return ops.prep(t.reshape(f.C, f.R, f.p))

    def _index(self, node):
        for i, f in enumerate(self.cols):
            if f.A is node:
                return i
        raise ValueError("node is not a factor of this network")

    def layout(self, node):
        i = self._index(node)
        self.canon(i)
        f = self.cols[i]
        return f.sizes, [node.dim_labels.index(lab) for lab in f.order]

    def to_canon(self, node, v):
        i = self._index(node)
        self.canon(i)
        f = self.cols[i]
        return ops.prep(v.permute([node.dim_labels.index(lab) for lab in f.order]).reshape(f.C * f.R, f.p, 1))

    def from_canon(self, node, t):
        f = self.cols[self._index(node)]
        perm = [node.dim_labels.index(lab) for lab in f.order]
        t = t.reshape(f.sizes)
        inv = [perm.index(a) for a in range(len(perm))]
        if inv != list(range(len(inv))):
            t = t.permute(inv)
        return t.to(node.tensor.dtype)

    # -- data
    def site_list(self, x):
        facs = self.require()
        per_input = list(x) if isinstance(x, (tuple, list)) else [x] * len(self.net.input_nodes)
        return [per_input[f.xin] for f in facs]

    def make_data(self, raw, shard=None):
        device = self.net.main_nodes[0].tensor.device
        done, xs = {}, []
        for t in raw:
            if id(t) not in done:
                tt = t if shard is None else shard.rows(t)
                tt = tt.to(device) if tt.device != device else tt
                if tt.ndim != 2:
                    raise ValueError(f"expected (samples, features) inputs, got shape {tuple(tt.shape)}")
                done[id(t)] = ops.prep(tt)
            xs.append(done[id(t)])
        d = _CPDData(raw, xs, len(raw))
        d.sharded = shard is not None
        d.S_total = raw[0].shape[0]
        return d

    def sync(self, d):
        for i, f in enumerate(self.cols):
            if _changed(d.seen[i], f.A):
                d.T[i] = None
                d.seen[i] = (f.A.tensor, f.A.tensor._version)

    def table(self, d, i):
        """T_i (S, C_i, R): an environment step from the identity over (c, b)."""
        if d.T[i] is None:
            core = self.canon(i)                                   # (C, R, p)
            f = self.cols[i]
            X = d.xs[i]
            d.T[i] = ops.gemm_rows(X, core.reshape(f.C * f.R, f.p).contiguous(), f.C * f.R, bmode=2).reshape(d.S, f.C, f.R)
        return d.T[i]

    def refresh(self, d):
        self.sync(d)
        for i in range(len(self.cols)):
            self.table(d, i)

    def _others(self, d, k, sum_outputs):
        """prod_{i != k} T_i as (S, C', R); the carrier's output axis is summed away when the
        reference does so (Jacobians of the factors that do not carry the output label)."""
        W = None
        for i in range(len(self.cols)):
            if i == k:
                continue
            T = self.table(d, i)
            if sum_outputs and T.shape[1] > 1:
                T = T.sum(dim=1, keepdim=True)
            W = T if W is None else W * T
        return W

    def predict(self, d, k=None):
        self.sync(d)
        P = None
        for i in range(len(self.cols)):
            T = self.table(d, i)
            P = T if P is None else P * T
        return P.sum(dim=-1)

    def local(self, d, node):
        self.sync(d)
        k = self._index(node)
        core = self.canon(k)
        f = self.cols[k]
        S, X = d.S, d.xs[k]
        C = 1
        for n in self.out_sizes():
            C *= n
        ones = torch.ones((S, 1), dtype=F64, device=X.device)
        back = lambda t, node=node: self.from_canon(node, t)   # noqa: E731
        if k == self.carrier or self.carrier is None:
            V = self._others(d, k, False)
            V = torch.ones((S, 1, f.R), dtype=F64, device=X.device) if V is None else V
            # J[s,c,(o,b,p)] = delta(c,o) V[s,b] X[s,p]
            eye = torch.eye(C, dtype=F64, device=X.device)
            L = (eye.reshape(1, C, C, 1) * V.reshape(S, 1, 1, f.R)).reshape(S, C, C * f.R).contiguous()
        else:
            # the reference contracts the carrier's output axis away and broadcasts the result
            # over the outputs (tensor/network.py:955-959 with 183): same row for every c
            V = self._others(d, k, True)
            L = V.expand(S, C, f.R).contiguous()
        return LocalProblem(S, C, (L.shape[2], f.p, 1), core.reshape(L.shape[2], f.p, 1), back, terms=[(L, X, ones)])

    def output_at(self, d, node, lp, core=None, rows=None):
        # the Jacobian of a non-carrier factor is not the derivative of the output when C > 1
        # (see local), so predictions come from the tables
        if core is not None:
            k = self._index(node)
            f = self.cols[k]
            lo, hi = rows if rows is not None else (0, d.S)
            T = ops.gemm_rows(d.xs[k][lo:hi].contiguous(), core.reshape(f.C * f.R, f.p).contiguous(), f.C * f.R,
                              bmode=2).reshape(hi - lo, f.C, f.R)
            W = self._others(d, k, False)
            P = T if W is None else T * W[lo:hi]
            return P.sum(dim=-1)
        y = self.predict(d)
        return y if rows is None else y[rows[0]:rows[1]]


# =====================================================================================  sum of networks
class _SumData:
    def __init__(self, x, parts, S_total, sharded):
        self.x = x
        self.parts = parts
        self.xs = parts[0].xs
        self.S = parts[0].S
        self.S_total = S_total
        self.sharded = sharded
        self.pred = [None] * len(parts)
        self.stamp = [None] * len(parts)


class SumEngine(EngineBase):
    """Output = sum of the member networks' outputs; each member reads the leading features of
    the shared input that its own input nodes were built for (tensor/network.py:1009-1021).
    A node's normal equations are those of its own network under the global gradient and
    Hessian (1023-1027)."""

    def __init__(self, net, members):
        super().__init__(net)
        self.members = members                       # TensorNetwork objects
        self.cols = None
        bad = [m for m in members if m._engine.cols is None]
        if bad:
            self.error = "; ".join(m._engine.error for m in bad)
        else:
            self.cols = [m._engine.cols for m in members]
        # feature count of every member input, fixed at construction (placeholder shapes)
        self.widths = [[tuple(b.tensor.shape[1:]) for b in m.input_nodes] for m in members]

    def require(self):
        if self.cols is None:
            raise NotImplementedError(f"member topology not recognised: {self.error}")
        return self.cols

    def _engine_of(self, node):
        for i, m in enumerate(self.members):
            if node in m.nodes:
                return i, m._engine
        raise ValueError("Node not found in any network")

    def out_labels(self):
        return self.members[0]._engine.out_labels()

    def out_sizes(self):
        return self.members[0]._engine.out_sizes()

    def layout(self, node):
        return self._engine_of(node)[1].layout(node)

    def to_canon(self, node, v):
        return self._engine_of(node)[1].to_canon(node, v)

    def from_canon(self, node, t):
        return self._engine_of(node)[1].from_canon(node, t)

    def data_for(self, x, shard=None):
        self.require()
        if not isinstance(x, torch.Tensor):
            raise TypeError("a sum of networks takes one input tensor shared by all members")
        for d in self.datasets:
            if d.x is x and d.sharded == (shard is not None):
                return d
        device = self.net.main_nodes[0].tensor.device
        xx = x if shard is None else shard.rows(x)
        xx = ops.prep(xx.to(device) if xx.device != device else xx)
        parts = []
        for m, widths in zip(self.members, self.widths):
            cut = {}
            per_input = []
            for w in widths:
                if w not in cut:
                    cut[w] = xx[(Ellipsis,) + tuple(slice(0, n) for n in w)].contiguous()
                per_input.append(cut[w])
            eng = m._engine
            raw = eng.site_list(per_input)
            parts.append(eng.make_data(raw, None))
        d = _SumData(x, parts, x.shape[0], shard is not None)
        for part in parts:
            part.sharded, part.S_total = d.sharded, d.S_total
        self.datasets.append(d)
        if len(self.datasets) > 3:
            self.datasets.pop(0)
        return d

    def sync(self, d):
        for i, m in enumerate(self.members):
            m._engine.sync(d.parts[i])

    def refresh(self, d):
        for i, m in enumerate(self.members):
            m._engine.refresh(d.parts[i])

    def _member_pred(self, d, i):
        m = self.members[i]
        stamp = tuple((id(n.tensor), n.tensor._version) for n in m.nodes if n not in m.input_nodes)
        if d.pred[i] is None or d.stamp[i] != stamp:
            d.pred[i] = m._engine.predict(d.parts[i])
            d.stamp[i] = stamp
        return d.pred[i]

    def predict(self, d, k=None):
        y = None
        for i in range(len(self.members)):
            p = self._member_pred(d, i)
            y = p.clone() if y is None else y.add_(p)
        return y

    def local(self, d, node):
        i, eng = self._engine_of(node)
        return eng.local(d.parts[i], node)

    def prefetch(self, d, node):
        i, eng = self._engine_of(node)
        eng.prefetch(d.parts[i], node)

    def output_at(self, d, node, lp, core=None, rows=None):
        i, eng = self._engine_of(node)
        y = eng.output_at(d.parts[i], node, lp, core, rows)
        for j in range(len(self.members)):
            if j != i:
                p = self._member_pred(d, j)
                y = y + (p if rows is None else p[rows[0]:rows[1]])
        return y
