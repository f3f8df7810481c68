"""Tensor-train network with the reference's method surface (reference: tensor/network.py).

The reference contracts a labelled graph with generated torch.einsum calls and
rebuilds every environment for every mini-batch.  Here the graph is inspected
once (engine.ChainEngine): a tensor-train chain with plain or conv-featurised
columns is mapped to canonical (rl, p, rr) cores, the per-sample left/right
environments of the whole data set stay resident in HBM and are refreshed only
where a node changed, and the arithmetic runs in the HIP kernels of libttk.so.
Sums over mini-batches in the reference are sums over samples, so one pass over
the full data gives the same Gram matrix, right-hand side and (batch-mean) loss.
"""
import os
import time
from collections import deque

import torch

from .. import ops
from .bregman import closed_form_symmetric_hessian
from .engine import ChainEngine
from .node import TensorNode
from .solvers import (_cholesky_solve, _host_solver, _scipy_kind, cg_solve, cg_solve_fused,  # noqa: F401
                      lanczos_solve, minres_solve)

_LEGACY_KWARGS = ("eps_r", "delta", "D_reg_fn")  # accepted and ignored: stale call sites pass them


class _FactorisationAborted(Exception):
    """The single-launch Cholesky gave up waiting (info = -1); the node update has been undone."""


class TensorNetwork:
    def __init__(self, input_nodes, main_nodes, train_nodes=None, output_labels=('s',), sample_dim='s'):
        self.input_nodes = input_nodes
        self.main_nodes = main_nodes
        self.train_nodes = main_nodes if train_nodes is None else train_nodes
        self.left_stacks = None
        self.right_stacks = None
        self.output_labels = output_labels
        self.sample_dim = sample_dim
        self.nodes, self.node_indices = self._discover_nodes()
        self._current = None
        self._shard = None
        self._pending = None
        self._spec = None
        self._streams = None        # (caller's stream, pipeline stream) while accumulating_swipe runs on two streams
        self._nodes_ready = None    # event on the pipeline stream: every node as of the last finished update
        self._in_user = False       # inside a user callback of that sweep
        self._sweep_d = None        # the data set that sweep works on
        self._engine = self._make_engine()

    def _make_engine(self):
        """The chain engine for the recognised shapes; any other graph the reference's stack algorithm
        can contract goes to the pairwise-contraction engine."""
        eng = ChainEngine(self)
        if eng.cols is None:
            from .graph_engine import GraphEngine
            return GraphEngine(self, chain_error=eng.error)
        return eng

    def distribute(self, shard="auto"):
        """Shard the training samples of the sweep methods over the ranks of a
        torch.distributed group (tensornetworksfork_amd.parallel.Shard); every rank
        passes the same full data and ends with identical cores.  "auto" picks the
        default process group when one is initialised."""
        from ..parallel import default_shard
        self._shard = default_shard() if shard == "auto" else shard
        self._engine.datasets = []
        return self

    # ------------------------------------------------------------------ graph
    def _discover_nodes(self):
        """Breadth-first over non-bond edges; every node gets the column index of
        the main node it hangs from (tensor/network.py:38-53)."""
        index = {n: i for i, n in enumerate(self.main_nodes)}
        queue = deque(self.main_nodes)
        while queue:
            n = queue.popleft()
            for label, other in n.connections.items():
                if other not in index and not n.is_horizontal_bond(label):
                    index[other] = index[n]
                    queue.append(other)
        return sorted(index, key=lambda n: n.name), index

    def _require_chain(self):
        return self._engine.require()

    def get_column_nodes(self, node):
        i = self.node_indices[node]
        return [n for n, j in self.node_indices.items() if j == i and n is not node]

    # ------------------------------------------------------------------ devices
    def cuda(self):
        for n in self.nodes:
            n.cuda()
        return self

    def to(self, device=None, dtype=None):
        for n in self.nodes:
            n.to(device=device, dtype=dtype)
        return self

    # ------------------------------------------------------------------ data sets
    def _data_for(self, x, shard=None):
        return self._engine.data_for(x, shard=shard)

    def _predict(self, d, k=None):
        self._user_guard(d)
        return self._engine.predict(d, k)

    # ------------------------------------------------------------------ two-stream sweep
    # accumulating_swipe queues its own work (local problems, solves, node updates, gauge moves) on a
    # pipeline stream and runs the user's callbacks on the caller's stream.  A callback that evaluates the
    # model on held-out data and reads a score back then waits for its own few kernels only, not for the
    # next node's Gram matrix that is already queued behind the update it looks at; the host is back in time
    # to queue the next solve while that Gram matrix is still being computed.
    _pipe_streams = {}

    def _two_streams(self, device):
        """(caller's stream, pipeline stream) for a sweep on `device`, or None where one stream is used:
        CPU tensors, a sweep started from inside a callback of another sweep, engines other than the chain
        engine (their callbacks may reach member networks this guard does not see)."""
        if device.type != 'cuda' or self._streams is not None or type(self._engine) is not ChainEngine \
                or os.environ.get("TTK_ONE_STREAM") == "1":
            return None
        user = torch.cuda.current_stream(device)
        pipe = TensorNetwork._pipe_streams.get(device)
        if pipe is None:
            # High priority: streams of one priority share a few hardware queues in creation order, and behind the
            # streams a collective library creates the pipeline stream can land on the caller's queue, which runs
            # the two one after the other (measured on the multi-rank path: 0.3 ms per core update).
            pipe = TensorNetwork._pipe_streams[device] = torch.cuda.Stream(device, priority=-1)
        return user, pipe

    def _user_guard(self, d=None, heavy=False):
        """Called by everything a callback may reach.  Work on other data (a validation set) runs beside
        the pipeline; anything that touches the sweep's own data set or a shared workspace (Gram,
        dense matrix, Cholesky storage) first lets the caller's stream wait for the pipeline."""
        if self._in_user and self._streams is not None and (heavy or d is None or d is self._sweep_d):
            user, pipe = self._streams
            user.wait_stream(pipe)

    def _user(self, fn, *args):
        """A user callback of the two-stream sweep: on the caller's stream, after the last node update;
        whatever it queued is waited for before the pipeline goes on."""
        if self._streams is None:
            return fn(*args)
        user, pipe = self._streams
        if self._nodes_ready is not None:
            user.wait_event(self._nodes_ready)
        else:
            user.wait_stream(pipe)
        self._in_user = True
        try:
            with torch.cuda.stream(user):
                return fn(*args)
        finally:
            self._in_user = False
            pipe.wait_stream(user)

    def _shape_output(self, y):
        """(S, C) -> TensorNode labelled like the reference's forward output."""
        eng = self._engine
        labels = eng.out_labels()
        node = TensorNode(y.reshape([y.shape[0]] + eng.out_sizes()), [self.sample_dim] + labels, name='O')
        if self.output_labels is not None:
            node.permute_first(*self.output_labels)
        return node

    # ------------------------------------------------------------------ reference stack API
    def reset_stacks(self, node=None):
        self.left_stacks = None
        self.right_stacks = None

    def recompute_all_stacks(self, exclude_nodes=set()):
        if self._current is not None:
            self._user_guard(self._current)
            self._engine.refresh(self._current)

    def left_update_stacks(self, node):
        """Environments are refreshed lazily from node identities; nothing to do eagerly."""

    def right_update_stacks(self, node):
        """See left_update_stacks."""

    def set_input(self, x):
        """Binds the inputs to the input nodes; True if any binding changed (tensor/network.py:329-345)."""
        changed = False
        tensors = list(x) if isinstance(x, (tuple, list)) else [x] * len(self.input_nodes)
        for node, t in zip(self.input_nodes, tensors):
            if node.tensor is not t:
                node.set_tensor(t)
                changed = True
        if changed:
            self.reset_stacks()
        self._current = self._data_for(x)
        return changed

    # ------------------------------------------------------------------ forward
    def forward(self, x, to_tensor=False):
        self._require_chain()
        self.set_input(x)
        out = self._shape_output(self._predict(self._current))
        return out.tensor if to_tensor else out

    def forward_batch(self, x, batch_size):
        """Per-sample results do not depend on the batching, so the whole set is
        evaluated in one pass (tensor/network.py:139-150)."""
        self._require_chain()
        d = self._data_for(x)
        return self._shape_output(self._predict(d)).tensor

    # ------------------------------------------------------------------ local problem
    def _norm_gh(self, grad, hessian, S, C):
        return ops.prep(grad.reshape(S, C)), self._norm_h(hessian, S, C)

    @staticmethod
    def _norm_h(hessian, S, C):
        """(S,C,C) Hessian; a trailing size-1 axis (the reference's Square loss) broadcasts."""
        H = hessian
        if H.ndim == 2:
            H = H.unsqueeze(-1)
        H = H.reshape(S, C, -1)
        if H.shape[-1] == 1 and C > 1:
            H = H.expand(S, C, C)
        return ops.prep(H)

    def _hessian_symmetric(self, H, sharded=False):
        """Whether every per-sample Hessian is symmetric (one host read when C > 1).  It selects the
        layout of the compressed Gram and with it the size of the all-reduce message, so sharded
        runs take the AND over the ranks."""
        if H.shape[1] == 1:
            return True
        sym = bool(torch.equal(H, H.transpose(1, 2)))
        if sharded and self._shard is not None:
            sym = self._shard.all_true(sym, H.device)
        return sym

    def _local(self, node):
        self._require_chain()
        if self._current is None:
            raise RuntimeError("no input bound: call forward(x) / set_input(x) first")
        d = self._current
        self._user_guard(d, heavy=True)
        return d, self._engine.local(d, node)

    def _gram_matrix(self, lp, H, scale=None, ridge=0.0, out=None):
        """Dense (D,D) Gram matrix of a local problem, optionally scaled and ridged."""
        sym = self._hessian_symmetric(H)
        if lp.compressed_gram(sym):
            ws = self._engine.gram_ws(lp, H, sym)
            return ops.gram_expand(ws, scale=scale, ridge=ridge, out=out)
        A = lp.dense_gram(H)
        if scale is not None:
            A.div_(scale)
        if ridge:
            A.diagonal().add_(ridge)
        return A

    def _node_axes(self, node, A):
        """Canonical (D,D) -> node.shape + node.shape in the node's own axis order."""
        sizes, perm = self._engine.layout(node)
        A = A.reshape(list(sizes) + list(sizes))
        inv = [perm.index(a) for a in range(len(perm))]
        if inv != list(range(len(inv))):
            n = len(inv)
            A = A.permute(inv + [n + a for a in inv])
        return A

    @torch.no_grad()
    def get_A_b(self, node, grad, hessian, method=None):
        """Gram matrix (node.shape + node.shape) and right-hand side (node.shape)
        of the local normal equations for the bound input (tensor/network.py:174-217)."""
        d, lp = self._local(node)
        g, H = self._norm_gh(grad, hessian, d.S, lp.C)
        b = lp.back(lp.rhs(g))
        if method is not None:
            return torch.randn((2, 2, 2, 2)), b   # the reference's placeholder for method='gradient'
        A = self._node_axes(node, self._gram_matrix(lp, H))
        return A.to(node.tensor.dtype), b

    @torch.no_grad()
    def get_b(self, node, grad):
        d, lp = self._local(node)
        return lp.back(lp.rhs(ops.prep(grad.reshape(d.S, lp.C))))

    @torch.no_grad()
    def matvec(self, node, v, hessian):
        """(J^T H J) v for the bound input, never forming the Gram matrix
        (the two einsums at tensor/network.py:788-789)."""
        d, lp = self._local(node)
        H = self._norm_h(hessian, d.S, lp.C)
        return lp.back(lp.matvec_fn(H)(self._engine.to_canon(node, v)))

    @torch.no_grad()
    def get_J(self, node, grad):
        """Per-sample Jacobian of the output with respect to `node` and the einsum strings that
        contract it (tensor/network.py:219-256): axes are the output labels the node does not
        carry (sample first), then the node's own labels; an output label that sits on the node
        itself appears once, as in the reference.  Materialises S*C*D numbers - the sweep methods
        of this build never call it; use matvec(node, v, hessian) for products."""
        import string
        d, lp = self._local(node)
        sdim = self.sample_dim
        sizes, perm = self._engine.layout(node)
        out0 = self._engine.out_labels()
        J = lp.rows(0, d.S).reshape([d.S] + self._engine.out_sizes() + list(sizes))
        no = 1 + len(out0)
        inv = [perm.index(a) for a in range(len(perm))]
        J = J.permute(list(range(no)) + [no + a for a in inv])
        labels = [sdim] + out0 + list(node.dim_labels)
        for lab in out0:                       # J[s, c, (c', ...)] = delta(c, c') * rest: drop the node-side copy
            if lab in node.dim_labels:
                idx = len(labels) - 1 - labels[::-1].index(lab)
                J = J.sum(dim=idx)
                del labels[idx]
        outs_all = list(self.output_labels or (sdim,))
        Jn = TensorNode(J.to(node.tensor.dtype).contiguous(), labels, name='J')
        Jn.permute_first(*[lab for lab in outs_all if lab not in node.dim_labels])
        letters = iter(string.ascii_letters)
        dl = {lab: next(letters) for lab in outs_all}
        olf = [lab for lab in outs_all if lab != sdim]
        for lab in olf:
            dl['_' + lab] = next(letters)
        for lab in Jn.dim_labels:
            if lab not in dl:
                dl[lab] = next(letters)
                dl['_' + lab] = next(letters)
        return {'J': Jn,
                'einsum': ''.join(dl[lab] for lab in Jn.dim_labels),
                'node_ein': ''.join(dl[lab] for lab in node.dim_labels),
                'dd_loss_ein': dl[sdim] + ''.join(dl[lab] for lab in olf) + ''.join(dl['_' + lab] for lab in olf),
                'd_loss_ein': ''.join(dl[lab] for lab in outs_all),
                'coeff_ein': dl[sdim] + ''.join(dl['_' + lab] for lab in olf)}

    def solve_system(self, node, A, b, method='exact', eps=0.0):
        """Scale by mean |diag|, ridge 2*eps on the scaled system with the unscaled
        weights in the right-hand side, x = -A^-1 b (tensor/network.py:293-327)."""
        m = method.lower()
        if m == 'gradient':
            return (-b).reshape(b.shape)
        half = A.ndim // 2
        A_f = A.flatten(0, half - 1).flatten(1, -1)
        b_f = b.flatten()
        scale = A_f.diag().abs().mean()
        scale = torch.where(scale == 0, torch.ones_like(scale), scale)
        A_f = A_f / scale
        b_f = b_f / scale
        if m in ('ridge_exact',) or m.startswith('ridge_cholesky'):
            A_f = A_f + (2 * eps) * torch.eye(A_f.shape[-1], dtype=A_f.dtype, device=A_f.device)
            b_f = b_f + (2 * eps) * node.tensor.flatten()
        if m in ('exact', 'ridge_exact'):
            x = torch.linalg.solve(A_f, -b_f)
        elif m == 'cholesky' or m.startswith('ridge_cholesky'):
            x = _cholesky_solve(A_f, -b_f)
        else:
            raise ValueError(f"Unknown method: {method}")
        return x.reshape(b.shape)

    # ------------------------------------------------------------------ one node update on the full data
    def _state_sig(self):
        """Identity + version of every parameter tensor: what a local problem depends on."""
        inputs = set(self.input_nodes)
        return tuple((id(n.tensor), n.tensor._version) for n in self.nodes if n not in inputs)

    def _site_front(self, d, node, y, loss_fn, batch_size, want_loss):
        """Everything of a node update that does not move a node: local problem, loss, gradient,
        right-hand side and Gram matrix (into the workspace), all queued on the stream."""
        self._user_guard(d, heavy=True)
        lp = self._engine.local(d, node)
        y_pred = self._engine.output_at(d, node, lp)  # the output is linear in every single node
        loss, grad, hess = loss_fn.forward(self._shape_output(y_pred).tensor, y)
        g, H = self._norm_gh(grad, hess, d.S, lp.C)
        total = self._host_scalar(self._batch_mean_loss(d, loss, batch_size)) if want_loss else None
        b = lp.rhs(g)
        front = {"lp": lp, "total": total, "b": b, "ws": None, "A_f": None, "gen": None, "mark": self._mark_pick()}
        # (the closed-form losses are symmetric by construction: no host read, no collective for them)
        sym = closed_form_symmetric_hessian(loss_fn) or self._hessian_symmetric(H, sharded=d.sharded)
        if lp.compressed_gram(sym):
            # compressed symmetric Gram in the workspace, own Cholesky on it
            self._mark("front", front["mark"])
            ws = self._engine.gram_ws(lp, H, sym, shard=self._shard if d.sharded else None)
            self._mark("gram", front["mark"])
            if d.sharded:   # one all-reduce of [compressed Gram | RHS] per core update
                self._shard.allreduce_gram_(ws, b)
                self._mark("comm", front["mark"])
            front["ws"] = ws
            front["gen"] = ws.generation   # the workspace is shared: a later Gram into it outdates this front
        else:
            # several Khatri-Rao terms (conv bond) or a conv node: explicit rows and the dense Gram kernel
            if d.sharded:   # [dense Gram | RHS] laid out as one buffer: one all-reduce per core update here too
                msg = torch.zeros(lp.D * lp.D + b.numel(), dtype=b.dtype, device=b.device)
                A_f = lp.dense_gram(H, out=msg[:lp.D * lp.D].view(lp.D, lp.D))
                msg[lp.D * lp.D:].copy_(b.reshape(-1))
                self._shard.allreduce_sum_(msg)
                front["b"] = msg[lp.D * lp.D:].view(b.shape)
            else:
                A_f = lp.dense_gram(H)
            front["A_f"] = A_f
        return front

    def _speculate(self, d, node, y, loss_fn, batch_size, want_loss):
        """Queue the front half of the NEXT node's update before the host waits for anything of the
        current one (pivot flag, callbacks with their own host reads): the GPU then always has the
        next Gram matrix to work on.  Nothing here moves a node; the result is used only if no
        parameter tensor has changed by the time the node is visited, else it is dropped."""
        key = (id(d), id(node), id(y), id(loss_fn), batch_size, want_loss, self._state_sig())
        alive = [n.tensor for n in self.nodes]   # held so that no id in the key can be reused meanwhile
        self._spec = (key, self._site_front(d, node, y, loss_fn, batch_size, want_loss), alive)

    def _site_update(self, d, node, y, loss_fn, batch_size, method, eps_, lr, adaptive_step, min_norm, max_norm,
                     want_loss):
        """Gram + RHS over all samples, solve, update the node.  Returns the
        mean-of-batch-means loss (float, or None if not requested)."""
        m = method.lower()
        if m not in ('exact', 'ridge_exact', 'cholesky') and not m.startswith('ridge_cholesky'):
            raise ValueError(f"Unknown method: {method}")
        spec, self._spec = self._spec, None
        front = None
        if spec is not None and spec[0] == (id(d), id(node), id(y), id(loss_fn), batch_size, want_loss, self._state_sig()):
            front = spec[1]
            if front["ws"] is not None and front["ws"].generation != front["gen"]:
                front = None     # a callback computed another Gram matrix into the shared workspace
        if front is None:
            front = self._site_front(d, node, y, loss_fn, batch_size, want_loss)
        lp, total, b = front["lp"], front["total"], front["b"]
        core = lp.core
        if eps_ == 0 and m == 'ridge_exact':
            m = 'exact'
        ridged = m == 'ridge_exact' or m.startswith('ridge_cholesky')
        ridge = 2 * eps_ if ridged else 0.0
        lu = m in ('exact', 'ridge_exact')
        if front["ws"] is not None:
            ws = front["ws"]
            scale = ops.gram_scale(ws)
            self._mark("pre", front["mark"])
            x, info = ops.chol_solve_gram(ws, b, core if ridged else None, scale, ridge)
            self._mark("solve", front["mark"])
            if not lu:
                self._pend_pivot_check(info, node)
            elif int(info.item()) != 0:
                # the reference's LU with partial pivoting (torch.linalg.solve, tensor/network.py:304-310) also takes
                # what is not positive definite; a matrix that is goes through the Cholesky kernels like the others
                A_f = ops.gram_expand(ws, scale=scale, ridge=ridge, out=self._engine.dense_buffer(lp.D, b.device))
                x = torch.linalg.solve(A_f, ops.solve_rhs(b, core if ridged else None, scale, ridge))
        else:
            A_f = front["A_f"]
            # scaling, ridge shift and the blocked Cholesky of the dense matrix, all on the device
            x, info = ops.chol_solve_dense(A_f, b.reshape(-1).contiguous(),
                                           core.reshape(-1).contiguous() if ridged else None, ridge)
            if not lu:
                self._pend_pivot_check(info, node)
            elif int(info.item()) != 0:
                scale = A_f.diagonal().abs().mean()
                scale = torch.where(scale == 0, torch.ones_like(scale), scale)
                A_f.div_(scale)
                if ridge:
                    A_f.diagonal().add_(ridge)
                rhs = ops.solve_rhs(b, core if ridged else None, scale, ridge).reshape(-1)
                x = torch.linalg.solve(A_f, rhs)
        node.update_node(lp.back(x), lr=lr, adaptive_step=adaptive_step, min_norm=min_norm, max_norm=max_norm)
        return total

    @staticmethod
    def _host_scalar(t):
        """A device scalar on its way to pinned host memory, with the event that says it has arrived:
        reading it later waits for this copy only, not for what has been queued behind it."""
        if not (isinstance(t, torch.Tensor) and t.is_cuda):
            return t
        host = torch.empty((), dtype=t.dtype, pin_memory=True)
        host.copy_(t, non_blocking=True)
        ready = torch.cuda.Event()
        ready.record()
        return (host, ready)

    def _mark_pick(self):
        """Phase marks for benchmarking: with `self._marks` set to a list, device events are recorded at the phase
        boundaries of every `self._mark_every`-th core update (an event costs about as much as a small kernel; no
        effect, no cost otherwise).  Returns the index of the update whose front is being queued if it is one of
        those, else None."""
        if getattr(self, "_marks", None) is None:
            return None
        k = self._mark_count = getattr(self, "_mark_count", -1) + 1
        return k if k % max(1, getattr(self, "_mark_every", 1)) == 0 else None

    def _mark(self, name, index):
        if index is not None and getattr(self, "_marks", None) is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._marks.append((name, ev, index))

    def _pend_pivot_check(self, info, node):
        """The pivot check of a Cholesky needs the host; it is settled after the gauge move has been
        queued (see _settle), so that nothing waits for the factorisation to drain first."""
        if info.is_cuda:   # copied to pinned memory now, so that reading it does not wait for later kernels
            host = torch.empty((), dtype=info.dtype, pin_memory=True)
            host.copy_(info, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record()
            info = (host, ready)
        self._pending = (info, [(n, n.tensor, list(n.dim_labels), list(n.left_labels), list(n.right_labels))
                                for n in dict.fromkeys(list(self.main_nodes) + [node])])

    def _settle(self):
        """Host-side check of the last Cholesky: on a non-positive pivot the nodes are put back as they
        were before the update (the reference raises inside solve_system, before any update) and
        torch.linalg.LinAlgError is raised."""
        pending, self._pending = self._pending, None
        if pending is None:
            return
        info, snapshot = pending
        if isinstance(info, tuple):
            info[1].synchronize()
            info = info[0]
        bad = int(info.item())
        if bad != 0:
            for n, t, labels, left, right in snapshot:
                n.tensor, n.dim_labels, n.left_labels, n.right_labels = t, labels, left, right
            if bad < 0:
                raise _FactorisationAborted()
            raise torch.linalg.LinAlgError(
                "linalg.cholesky: The factorization could not be completed because the input is not "
                f"positive-definite (the leading minor of order {bad} is not positive-definite).")

    def _batch_mean_loss(self, d, loss, batch_size):
        """Mean over the reference's mini-batches of per-batch mean losses (device scalar);
        with sharded samples the ranks exchange their per-batch partial sums."""
        v = ops.prep(loss.reshape(d.S, -1))
        if d.sharded:
            return self._shard.batch_mean(v, d.S_total, batch_size)
        return ops.batch_mean(v, batch_size)

    def _gradient_pass(self, d, node, y, loss_fn, batch_size, lr, adaptive_step, min_norm, max_norm, per_batch_update):
        """method='gradient' (tensor/network.py:458-470): the node moves after every
        mini-batch in the first direction, so batches are visited one by one.  The Jacobian
        of a node does not depend on the node itself, so it is set up once per visit."""
        lp = self._engine.local(d, node)
        total, b_out, batches = 0.0, None, 0
        for b0 in range(0, d.S, batch_size):
            lo, hi = b0, min(d.S, b0 + batch_size)
            sub = lp.slice(lo, hi)
            y_pred = self._engine.output_at(d, node, sub, self._engine.to_canon(node, node.tensor), rows=(lo, hi))
            loss, grad, _ = loss_fn.forward(self._shape_output(y_pred).tensor, y[lo:hi])
            bv = sub.rhs(ops.prep(grad.reshape(hi - lo, lp.C)))
            b_out = bv if b_out is None else b_out + bv
            if per_batch_update:
                node.update_node(lp.back(bv), lr=lr, adaptive_step=adaptive_step, min_norm=min_norm, max_norm=max_norm)
            total += loss.mean().item()
            batches += 1
        if not per_batch_update:
            node.update_node(lp.back(-b_out), lr=lr, adaptive_step=adaptive_step, min_norm=min_norm, max_norm=max_norm)
        return total / batches

    def accumulating_swipe(self, x, y_true, loss_fn, node_order=None, batch_size=-1, num_swipes=1, lr=1.0,
                           method='exact', eps=1e-12, eps_decay=None, convergence_criterion=None, orthonormalize=False,
                           verbose=False, skip_second=False, blocks_input=False, timeout=None, data_device=None,
                           model_device=None, disable_tqdm=None, block_callback=None, loss_callback=None,
                           direction='l2r', update_or_reset_stack='reset', adaptive_step=False, min_norm=None,
                           max_norm=None, eps_per_node=False, **legacy):
        """DMRG sweeps with a Gauss-Newton solve per core (tensor/network.py:379-608).
        Same schedule, ridge schedule, skip rule, callbacks and return value as the
        reference; the mini-batch loop is replaced by one pass over the resident data.
        `update_or_reset_stack` is accepted and has no effect: the cached environments are always the ones of
        the current nodes.  (The reference's 'update' mode, which none of its callers selects, refreshes one
        environment per update and leaves the prediction, which is contracted at the first node, on stale ones:
        tensor/network.py:115-131 with 489-492.)"""
        for k_ in legacy:
            if k_ not in _LEGACY_KWARGS:
                raise TypeError(f"accumulating_swipe() got an unexpected keyword argument '{k_}'")
        self._require_chain()
        self._pending = None
        self._spec = None
        d = self._data_for(x, shard=self._shard)
        self._current = d
        device = d.xs[0].device
        y = y_true if self._shard is None else self._shard.rows(y_true)
        y = y.to(device) if y.device != device else y
        if d.sharded and method == 'gradient':
            raise NotImplementedError("method='gradient' visits mini-batches sequentially and is not sharded")
        S = d.S_total
        if batch_size <= 0 or blocks_input:
            batch_size = S
        start = time.time() if timeout is not None else None
        want_loss = loss_callback is not None or (verbose and verbose > 1)
        last = {"fwd": None, "bwd": None}

        def eps_for(NS):
            e = eps[NS] if isinstance(eps, list) else eps
            return e * eps_decay ** NS if eps_decay is not None else e

        def run_pass(order, which, other, NS, eps_, reverse_gauge):
            for i, node in enumerate(order):
                if eps_per_node:
                    eps_ = eps[i if not reverse_gauge_index[which] else len(order) - 1 - i] if isinstance(eps, list) else eps
                last[which] = node     # bound before the skip test, like the reference's loop variable
                o = last[other]
                if node in self.node_indices and o in self.node_indices and self.node_indices[node] == self.node_indices[o]:
                    continue
                if self._timed_out(d, start, timeout):
                    print(f"Timeout reached ({timeout} seconds). Stopping accumulating_swipe.")
                    return False
                try:
                    for attempt in (0, 1):
                        if method == 'gradient':
                            loss = self._gradient_pass(d, node, y, loss_fn, batch_size, lr, adaptive_step, min_norm,
                                                       max_norm, per_batch_update=(which == "fwd"))
                        else:
                            loss = self._site_update(d, node, y, loss_fn, batch_size, method, eps_, lr, adaptive_step,
                                                     min_norm, max_norm, want_loss)
                        if orthonormalize:
                            (self.node_orthonormalize_right if reverse_gauge else self.node_orthonormalize_left)(node)
                        if self._streams is not None:   # what the callbacks of this update may look at is queued
                            self._nodes_ready = torch.cuda.Event()
                            self._nodes_ready.record()
                        if i + 1 < len(order) and method != 'gradient':
                            # the next node's Gram matrix is queued before the host waits for the pivot flag
                            # and before the callbacks take the host (see _speculate)
                            nxt = order[i + 1]
                            if not (nxt in self.node_indices and self.node_indices[nxt] == self.node_indices.get(last[other])):
                                self._speculate(d, nxt, y, loss_fn, batch_size, want_loss)
                        try:
                            self._settle()
                            break
                        except _FactorisationAborted:
                            # The single-launch factorisation gave up (its workgroups wait for each other, which needs
                            # most of the device: a compute-unit mask or a partitioned device can starve it).  The update
                            # has been undone: from here on this process factorises with one launch per panel, and the
                            # node is done again.
                            self._spec = None
                            # (sample-sharded: the abort is this rank's alone and a second attempt would issue collectives
                            # the other ranks do not: the rank fails instead of leaving them waiting)
                            if attempt or d.sharded or ops.chol_mode(0) == 0:
                                raise RuntimeError("ttk_chol_solve_f64: the factorisation was aborted (a workgroup waited "
                                                   "for a tile beyond its limit); the update has been undone") from None
                except torch.linalg.LinAlgError:
                    self._spec = None
                    if verbose and verbose > 0:
                        print(f"Singular system for node {node.name}")
                    return False
                if isinstance(loss, tuple):
                    loss[1].synchronize()
                    loss = loss[0].item()
                elif isinstance(loss, torch.Tensor):
                    loss = loss.item()
                if verbose and verbose > 1:
                    print(f"NS: {NS}, {'Left' if which == 'fwd' else 'Right'} loss ({node.name}):", loss, f" (eps: {eps_})")
                if loss_callback is not None:
                    self._user(loss_callback, NS, node, loss)
                if convergence_criterion is not None and self._user(convergence_criterion):
                    if verbose and verbose > 0:
                        print('Converged (left pass)' if which == "fwd" else 'Converged (right pass)')
                    if block_callback is not None:
                        self._user(block_callback, NS, node)
                    return True
                if block_callback is not None:
                    self._user(block_callback, NS, node)
            return None

        # index used for per-node eps lists: position in the un-reversed order (network.py:429,531)
        reverse_gauge_index = {"fwd": direction != 'l2r', "bwd": direction != 'r2l'}

        def swipes():
            NS = 0
            for _ in range(num_swipes):
                if node_order is not None:
                    first = node_order[0] if isinstance(node_order, tuple) else node_order
                else:
                    first = self.train_nodes
                first = list(first) if direction == 'l2r' else list(reversed(list(first)))
                r = run_pass(first, "fwd", "bwd", NS, eps_for(NS), reverse_gauge=False)
                if r is not None:
                    return r
                NS += 1
                if skip_second:
                    continue
                if node_order is not None:
                    second = node_order[1] if isinstance(node_order, tuple) else list(reversed(list(node_order)))
                else:
                    second = self.train_nodes
                second = list(second) if direction == 'r2l' else list(reversed(list(second)))
                r = run_pass(second, "bwd", "fwd", NS, eps_for(NS), reverse_gauge=True)
                if r is not None:
                    return r
                NS += 1
            return True

        streams = self._two_streams(device) if method != 'gradient' else None
        if streams is None:
            return swipes()
        user, pipe = streams
        pipe.wait_stream(user)
        self._streams, self._nodes_ready, self._sweep_d = streams, None, d
        try:
            with torch.cuda.stream(pipe):
                return swipes()
        finally:
            # a Gram queued ahead for a node the sweep never came to (it converged, timed out, ...) is called off: the
            # caller's stream is not queued behind it
            spec, self._spec = self._spec, None
            ws = spec[1].get("ws") if spec is not None else None
            if ws is not None and spec[1]["gen"] == ws.generation:
                with torch.cuda.stream(user):
                    ops.gram_cancel(ws)
            self._streams = self._nodes_ready = self._sweep_d = None
            user.wait_stream(pipe)

    def gradient_swipe(self, x, y_true, loss_fn, *args, **kwargs):
        """Name used by one caller of the reference that no reference class defines
        (image_convolution_lanczos_CIFAR10.py:87 passes batch_size, num_swipes, lr, max_iter,
        verbose, block_callback: the keywords of lanczos_swipe, tensor/network.py:709).  That
        call runs lanczos_swipe; a call that names a keyword only accumulating_swipe has
        (method, eps, node_order, ...) runs accumulating_swipe."""
        acc_only = ('node_order', 'method', 'eps', 'eps_decay', 'convergence_criterion', 'orthonormalize',
                    'skip_second', 'blocks_input', 'direction', 'update_or_reset_stack', 'adaptive_step',
                    'min_norm', 'max_norm', 'eps_per_node') + _LEGACY_KWARGS
        if args or any(k_ in kwargs for k_ in acc_only):
            return self.accumulating_swipe(x, y_true, loss_fn, *args, **kwargs)
        return self.lanczos_swipe(x, y_true, loss_fn, **kwargs)

    # ------------------------------------------------------------------ QR gauge (tensor/network.py:611-707)
    def orthonormalize_left(self):
        for n in self.main_nodes:
            self.node_orthonormalize_left(n)

    def orthonormalize_right(self):
        for n in self.main_nodes:
            self.node_orthonormalize_right(n)

    @staticmethod
    def _bond_product(Rm, t, first):
        """The triangular factor of a gauge move into the neighbouring core, over its first axis
        (sum_j Rm[i,j] t[j,...]) or its last one (sum_j t[...,j] Rm[j,i]): the row GEMM kernel on the
        device, an einsum for host tensors and other dtypes."""
        on_dev = t.is_cuda and t.dtype == torch.float64 and Rm.dtype == torch.float64 and Rm.shape[0] <= ops.DENSE_MAX_P \
            and Rm.shape[1] <= ops.DENSE_MAX_P
        if first:
            if on_dev:
                flat = t.reshape(t.shape[0], -1).contiguous()
                return ops.gemm_rows(Rm.contiguous(), flat, flat.shape[1]).reshape((Rm.shape[0],) + tuple(t.shape[1:]))
            return torch.einsum('ij,j...->i...', Rm, t)
        if on_dev:
            flat = t.reshape(-1, t.shape[-1]).contiguous()
            return ops.gemm_rows(flat, Rm.contiguous(), Rm.shape[1]).reshape(tuple(t.shape[:-1]) + (Rm.shape[1],))
        return torch.einsum('ji,...j->...i', Rm, t)

    def node_orthonormalize_left(self, node):
        """core -> Q with orthonormal columns over (left, phys); R moves into the next core."""
        i = self.node_indices[node]
        if i >= len(self.main_nodes) - 1:
            return
        bond = [node.dim_labels.index(lab) for lab in node.right_labels]
        rest = [a for a in range(node.tensor.ndim) if a not in bond]
        perm = rest + bond
        P = node.tensor.permute(perm)
        Q, Rm = ops.qr(P.flatten(0, len(rest) - 1))
        Q = Q.reshape(tuple(P.shape[:len(rest)]) + (Q.shape[-1],))
        node.tensor = Q.permute(*sorted(range(len(perm)), key=lambda a: perm[a]))
        nxt = self.main_nodes[i + 1]
        nxt.permute_first(*node.get_connecting_labels(nxt))
        nxt.tensor = self._bond_product(Rm, nxt.tensor, first=True)

    def node_orthonormalize_right(self, node):
        """Mirror image: RQ through a doubly flipped QR; R moves into the previous core."""
        i = self.node_indices[node]
        if i <= 0:
            return
        bond = [node.dim_labels.index(lab) for lab in node.left_labels]
        rest = [a for a in range(node.tensor.ndim) if a not in bond]
        T2 = node.tensor.reshape(node.tensor.shape[0], -1) if bond == [0] and node.tensor.is_contiguous() else None
        if T2 is not None and ops.qr_right_supported(T2):
            # the left bond is the leading axis of a dense core: one kernel reads it in place and writes the new core
            # and the bond factor as they are used
            Qt, Rm = ops.qr_right(T2)
            node.tensor = Qt.reshape((Qt.shape[0],) + tuple(node.tensor.shape[1:]))
        else:
            perm = rest + bond
            P = node.tensor.permute(perm)
            Qr, Rr = ops.qr(torch.flip(P.flatten(0, len(rest) - 1), dims=[0, 1]))
            Rm = torch.flip(Rr.T, dims=[0, 1])
            Q = torch.flip(Qr, dims=[0, 1]).reshape(tuple(P.shape[:len(rest)]) + (Qr.shape[-1],))
            node.tensor = Q.permute(*sorted(range(len(perm)), key=lambda a: perm[a]))
        prev = self.main_nodes[i - 1]
        prev.permute_last(*node.get_connecting_labels(prev))
        prev.tensor = self._bond_product(Rm, prev.tensor, first=False)

    # ------------------------------------------------------------------ matrix-free sweeps
    def _krylov_setup(self, d, node, y, loss_fn, batch_size, want_loss):
        """RHS phase shared by lanczos_swipe / scipy_swipe (tensor/network.py:746-767)."""
        self._user_guard(d, heavy=True)
        lp = self._engine.local(d, node)
        y_pred = self._engine.output_at(d, node, lp)
        loss, grad, hess = loss_fn.forward(self._shape_output(y_pred).tensor, y)
        g, H = self._norm_gh(grad, hess, d.S, lp.C)
        b = lp.rhs(g)
        if d.sharded:
            self._shard.allreduce_sum_(b)
        lossv = self._batch_mean_loss(d, loss, batch_size).item() if want_loss else None
        return lp, H, b, lossv

    def _matvec_fn(self, d, lp, H):
        """v -> (J^T H J) v over all samples; sharded runs add one all-reduce of the
        D-vector per product."""
        mv = lp.matvec_fn(H)
        if d.sharded:
            return lambda v: self._shard.allreduce_sum_(mv(v))
        return mv

    def _timed_out(self, d, start, timeout):
        """The timeout test of the sweeps (tensor/network.py:435,537,741,865).  Sample-sharded runs agree on it (one
        tiny all-reduce per node, only when a timeout is set): a rank that left by its own clock would leave the
        others waiting in the next collective."""
        if timeout is None:
            return False
        out = (time.time() - start) > timeout
        if d.sharded and self._shard is not None:
            out = not self._shard.all_true(not out, d.xs[0].device)
        return out

    def _sweep_data(self, x, y_true, batch_size):
        self._require_chain()
        d = self._data_for(x, shard=self._shard)
        self._current = d
        device = d.xs[0].device
        y = y_true if self._shard is None else self._shard.rows(y_true)
        y = y.to(device) if y.device != device else y
        return d, y, (d.S_total if batch_size <= 0 else batch_size)

    def lanczos_swipe(self, x, y_true, loss_fn, batch_size=1, num_swipes=1, lr=1.0, max_iter=50, tol=1e-6,
                      verbose=False, timeout=None, data_device=None, model_device=None, disable_tqdm=None,
                      block_callback=None, loss_callback=None):
        """Matrix-free Lanczos-Galerkin step per node (tensor/network.py:709-832).
        The start vector is drawn from the CPU generator (one randn per visited
        node, in visiting order) so that a seeded run reproduces the reference's CPU path."""
        d, y, batch_size = self._sweep_data(x, y_true, batch_size)
        device = d.xs[0].device
        start = time.time() if timeout is not None else None
        for NS in range(num_swipes):
            for node in (self.train_nodes if NS % 2 == 0 else reversed(self.train_nodes)):
                if self._timed_out(d, start, timeout):
                    print(f"Timeout reached ({timeout} seconds). Stopping lanczos_swipe.")
                    return False
                lp, H, b, lossv = self._krylov_setup(d, node, y, loss_fn, batch_size, loss_callback is not None)
                if loss_callback is not None:
                    loss_callback(lossv)
                x0n = torch.randn(tuple(node.tensor.shape), dtype=node.tensor.dtype)
                x0 = self._engine.to_canon(node, x0n.to(device)).reshape(lp.shape)
                if d.sharded:
                    self._shard.broadcast_(x0)
                fused = None if d.sharded else lp.lanczos_fused(H)   # (a sharded product needs an all-reduce in between)
                step = lanczos_solve(self._matvec_fn(d, lp, H), -b, x0, max_iter, tol, lockstep=d.sharded, fused=fused)
                node.update_node(lp.back(step), lr=lr)
                if block_callback is not None:
                    block_callback(NS, node)
        return True

    def scipy_swipe(self, x, y_true, loss_fn, solver, batch_size=1, num_swipes=1, lr=1.0, max_iter=50, tol=1e-6,
                    verbose=False, timeout=None, data_device=None, model_device=None, disable_tqdm=None,
                    block_callback=None, loss_callback=None):
        """CG / MINRES step per node (tensor/network.py:834-932).  scipy.sparse.linalg.cg
        and .minres are recognised and run as device-resident float64 iterations with
        SciPy's recurrences and stopping rules; any other callable receives a SciPy
        LinearOperator around the device mat-vec."""
        d, y, batch_size = self._sweep_data(x, y_true, batch_size)
        start = time.time() if timeout is not None else None
        warm = {}
        kind = _scipy_kind(solver)
        for NS in range(num_swipes):
            for node in (self.train_nodes if NS % 2 == 0 else reversed(self.train_nodes)):
                if self._timed_out(d, start, timeout):
                    print(f"Timeout reached ({timeout} seconds). Stopping lanczos_swipe.")
                    return False
                lp, H, b, lossv = self._krylov_setup(d, node, y, loss_fn, batch_size, loss_callback is not None)
                if loss_callback is not None:
                    loss_callback(lossv)
                mv = self._matvec_fn(d, lp, H)
                if kind == 'cg':
                    fused = None if d.sharded else lp.cg_fused(H)   # (a sharded product needs an all-reduce in between)
                    if fused is not None:
                        sol, _ = cg_solve_fused(fused, mv, -b, warm.get(node), max_iter, tol)
                    else:
                        sol, _ = cg_solve(mv, -b, warm.get(node), max_iter, tol)
                elif kind == 'minres':
                    sol, _ = minres_solve(mv, -b, warm.get(node), max_iter, tol)
                else:
                    sol = _host_solver(solver, mv, -b, warm.get(node), max_iter, tol)
                warm[node] = sol
                node.update_node(lp.back(sol), lr=lr)
                if block_callback is not None:
                    block_callback(NS, node)
        return True

    def disconnect(self, nodes):
        raise NotImplementedError("disconnect() is not part of the core-update path of this build")


class CPDNetwork(TensorNetwork):
    """Canonical polyadic model: one factor node per input, all sharing an unconnected rank
    label (tensor/network.py:935-986).  Per-factor tables replace the left/right environments."""

    def _make_engine(self):
        from .composite import CPDEngine
        return CPDEngine(self)

    def node_orthonormalize_left(self, node):
        """Factors carry no bonds; the gauge moves of the train do not apply."""

    def node_orthonormalize_right(self, node):
        """See node_orthonormalize_left."""


class SumOfNetworks(TensorNetwork):
    """Sum of the outputs of several networks over one shared input (tensor/network.py:988-1060).
    Nodes are renamed '<name>_n<i>'; train_operators=False trains the main nodes only.
    `only_bias_first` / `train_linear` of older call sites are accepted and ignored."""

    def __init__(self, networks, output_labels=('s',), sample_dim='s', train_operators=True, only_bias_first=None,
                 train_linear=None):
        input_nodes, main_nodes, train_nodes = [], [], []
        for i, net in enumerate(networks, 1):
            for n in list(net.input_nodes) + list(net.main_nodes):
                n.name = f"{n.name}_n{i}"
            input_nodes.extend(net.input_nodes)
            main_nodes.extend(net.main_nodes)
            train_nodes.extend(net.train_nodes if train_operators else net.main_nodes)
        self.networks = networks
        super().__init__(input_nodes, main_nodes, train_nodes, output_labels=output_labels, sample_dim=sample_dim)

    def _make_engine(self):
        from .composite import SumEngine
        return SumEngine(self, self.networks)

    def distribute(self, shard="auto"):
        super().distribute(shard)
        for net in self.networks:
            net._shard = self._shard
        return self

    def set_input(self, x):
        self._require_chain()
        d = self._data_for(x)
        changed = d is not self._current
        self._current = d
        return changed

    def orthonormalize_left(self):
        for net in self.networks:
            net.orthonormalize_left()

    def orthonormalize_right(self):
        for net in self.networks:
            net.orthonormalize_right()

    def node_orthonormalize_left(self, node):
        for net in self.networks:
            if node in net.main_nodes:
                net.node_orthonormalize_left(node)

    def node_orthonormalize_right(self, node):
        for net in self.networks:
            if node in net.main_nodes:
                net.node_orthonormalize_right(node)

    def left_update_stacks(self, node):
        raise NotImplementedError("left_update_stacks not implemented for SumOfNetworks")

    def right_update_stacks(self, node):
        raise NotImplementedError("right_update_stacks not implemented for SumOfNetworks")
