"""Engine for node graphs that are none of the recognised chain shapes: the reference's own
algorithm (tensor/network.py:55-137) on pairwise contractions.

Left / right stacks are built column by column with TensorNode.contract_with (the row GEMM and
per-sample contraction kernels), the Jacobian of a node is the contraction of everything but
the node (compute_jacobian_stack, network.py:101-113), and the local problem is handed to the
sweep machinery as explicit rows (the dense Gram / right-hand side / mat-vec kernels).  Nothing
is cached across node updates beyond one pair of stack chains per parameter state: this engine
exists so that every graph the reference can contract runs, the recognised shapes have the fast
one (engine.ChainEngine).
"""
import torch

from .. import ops
from .engine import EngineBase, LocalProblem, _Data

F64 = torch.float64


class GraphEngine(EngineBase):
    def __init__(self, net, chain_error=None):
        super().__init__(net)
        self.cols = True
        self.chain_error = chain_error     # why the chain planner declined this graph

    def require(self):
        return True

    # ------------------------------------------------------------------ output layout
    def out_labels(self):
        s = self.net.sample_dim
        return [lab for lab in (self.net.output_labels or ()) if lab != s and self._carrier(lab) is not None]

    def _carrier(self, lab):
        for n in self.net.nodes:
            if lab in n.dim_labels and n not in self.net.input_nodes:
                return n
        return None

    def out_sizes(self):
        return [self._carrier(lab).dim_size(lab) for lab in self.out_labels()]

    def out_dim(self):
        C = 1
        for n in self.out_sizes():
            C *= n
        return C

    # ------------------------------------------------------------------ data sets
    def site_list(self, x):
        return list(x) if isinstance(x, (tuple, list)) else [x] * len(self.net.input_nodes)

    def make_data(self, raw, shard=None):
        device = self.net.main_nodes[0].tensor.device
        done, xs = {}, []
        for t in raw:
            if id(t) not in done:
                tt = t if shard is None else shard.rows(t)
                done[id(t)] = ops.prep(tt.to(device) if tt.device != device else tt)
            xs.append(done[id(t)])
        d = _Data(raw, xs, len(raw))
        d.sharded = shard is not None
        d.S_total = raw[0].shape[0]
        d.stacks = None
        return d

    def sync(self, d):
        pass

    def refresh(self, d):
        self._stacks(d)

    def prefetch(self, d, node):
        self._stacks(d)

    # ------------------------------------------------------------------ contractions (network.py:55-113)
    class _Bound:
        """Input nodes carry this data set's (device, float64, possibly sharded) tensors while a
        contraction runs; parameter nodes are read as float64."""

        def __init__(self, eng, d):
            self.eng, self.d = eng, d

        def __enter__(self):
            net = self.eng.net
            self.saved = [(n, n.tensor) for n in net.nodes]
            for n, t in zip(net.input_nodes, self.d.xs):
                n.tensor = t
            inputs = set(net.input_nodes)
            for n in net.nodes:
                if n not in inputs:
                    n.tensor = ops.prep(n.tensor)

        def __exit__(self, *exc):
            for n, t in self.saved:
                n.tensor = t

    def _params(self):
        inputs = set(self.net.input_nodes)
        return [n.tensor for n in self.net.nodes if n not in inputs]

    def _column(self, node):
        return [node] + self.net.get_column_nodes(node)

    def _chain(self, direction):
        stacks, prev = {}, None
        for node in (self.net.main_nodes if direction == "left" else reversed(self.net.main_nodes)):
            it = iter(self._column(node))
            contracted = next(it) if prev is None else prev
            for v in it:
                contracted = contracted.contract_with(v, v.get_connecting_labels(contracted))
            stacks[node] = contracted
            prev = contracted
        return stacks

    def _stacks(self, d):
        held = self._params()          # kept with the stacks: a live tensor's id cannot be handed out again
        sig = tuple((id(t), t._version) for t in held)
        if d.stacks is None or d.stacks[0] != sig:
            with self._Bound(self, d):
                d.stacks = (sig, self._chain("left"), self._chain("right"), held)
        return d.stacks[1], d.stacks[2]

    def _neighbours(self, d, node):
        left, right = self._stacks(d)
        mains = self.net.main_nodes
        i = self.net.node_indices[node]
        return (left[mains[i - 1]] if i > 0 else None), (right[mains[i + 1]] if i < len(mains) - 1 else None)

    def predict(self, d, k=None):
        net = self.net
        node = net.main_nodes[0]
        lstack, rstack = self._neighbours(d, node)
        with self._Bound(self, d):
            it = iter(self._column(node))
            contracted = next(it) if lstack is None else lstack
            for v in it:
                contracted = contracted.contract_with(v, v.get_connecting_labels(contracted))
            if rstack is not None:
                contracted = contracted.contract_with(rstack, rstack.get_connecting_labels(contracted))
        order = [net.sample_dim] + self.out_labels()
        if sorted(contracted.dim_labels) != sorted(order):
            raise NotImplementedError(f"network output carries labels {contracted.dim_labels}, expected {order}")
        y = contracted.tensor.permute([contracted.dim_labels.index(lab) for lab in order])
        return y.reshape(d.S, -1).contiguous()

    def local(self, d, node):
        """Jacobian rows of the output with respect to `node`: the contraction of its left stack,
        the other nodes of its column and its right stack.  An output label the node itself carries
        enters as an identity between the output axis and the node's axis."""
        net = self.net
        lstack, rstack = self._neighbours(d, node)
        with self._Bound(self, d):
            it = iter(net.get_column_nodes(node))
            contracted = next(it) if lstack is None else lstack
            for v in it:
                contracted = contracted.contract_with(v, v.get_connecting_labels(contracted))
            if rstack is not None:
                contracted = contracted.contract_with(rstack, rstack.get_connecting_labels(contracted))
            core = ops.prep(node.tensor)
        s = net.sample_dim
        outs, sizes = self.out_labels(), self.out_sizes()
        on_node = [lab for lab in outs if lab in node.dim_labels]
        on_j = [lab for lab in outs if lab not in node.dim_labels]
        rest = [lab for lab in node.dim_labels if lab not in on_node]
        want = [s] + on_j + rest
        if sorted(contracted.dim_labels) != sorted(want):
            raise NotImplementedError(f"Jacobian of {node.name} carries labels {contracted.dim_labels}, expected {want}")
        J = contracted.tensor.permute([contracted.dim_labels.index(lab) for lab in want])
        cur = list(want)
        for lab in on_node:       # d out[.., o, ..] / d node[.., o', ..] = delta(o, o') x the rest
            n = node.dim_size(lab)
            eye = torch.eye(n, dtype=F64, device=J.device).reshape([n, n] + [1] * J.ndim)
            J = eye * J.unsqueeze(0).unsqueeze(0)
            cur = [lab, '_' + lab] + cur
        final = [s] + outs + ['_' + lab if lab in on_node else lab for lab in node.dim_labels]
        J = J.permute([cur.index(lab) for lab in final])
        S, C = d.S, self.out_dim()
        rows = J.reshape(S, C, -1).contiguous()
        shape = tuple(node.tensor.shape)
        back = lambda t, node=node: t.reshape(shape).to(node.tensor.dtype)   # noqa: E731
        return LocalProblem(S, C, shape, core, back, J=rows)

    # ------------------------------------------------------------------ canonical views: the node's own layout
    def layout(self, node):
        return list(node.tensor.shape), list(range(node.tensor.ndim))

    def to_canon(self, node, v):
        return ops.prep(v)

    def from_canon(self, node, t):
        return t.reshape(tuple(node.tensor.shape)).to(node.tensor.dtype)
