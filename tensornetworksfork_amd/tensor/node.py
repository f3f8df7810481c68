"""Labelled tensors and their graph edges (reference: tensor/node.py).

A node is a `torch.Tensor` plus one label per axis and a dict of edges to other
nodes.  The network (tensor/network.py of this package) reads the labels to
recognise the tensor-train chain and hands the raw device pointers to the HIP
kernels; the sweep methods never contract nodes pairwise.  `contract_with` /
`contract_vertically` are kept for callers that use the graph API directly.
"""
from collections import defaultdict

import torch


def _as_list(x):
    if x is None:
        return []
    return [x] if isinstance(x, str) else list(x)


def _pair_contract(A, la, B, lb, contract, out_labels):
    """sum over `contract` of A[la] B[lb] -> axes `out_labels`; labels in both operands that are not
    contracted are batch axes.  Brought to (batch, M, K) x (batch, K, N): one row GEMM without batch
    axes, one per-sample contraction kernel with them (float64 on the device, within the kernels'
    LDS limits); torch.matmul otherwise (host tensors, other dtypes)."""
    from .. import ops
    size = {}
    for t, labs in ((A, la), (B, lb)):
        for n, lab in zip(t.shape, labs):
            size[lab] = n
    for lab in contract:                      # a label only one operand carries is simply summed away
        if lab in la and lab not in lb:
            A = A.sum(la.index(lab))
            la = [x for x in la if x != lab]
        elif lab in lb and lab not in la:
            B = B.sum(lb.index(lab))
            lb = [x for x in lb if x != lab]
    K = [lab for lab in la if lab in lb and lab in contract]
    batch = [lab for lab in la if lab in lb and lab not in contract]
    M = [lab for lab in la if lab not in lb]
    N = [lab for lab in lb if lab not in la]
    prod = lambda labs: int(torch.tensor([size[x] for x in labs]).prod()) if labs else 1   # noqa: E731
    Bt, Mt, Kt, Nt = prod(batch), prod(M), prod(K), prod(N)
    A3 = A.permute([la.index(x) for x in batch + M + K]).reshape(Bt, Mt, Kt)
    B3 = B.permute([lb.index(x) for x in batch + K + N]).reshape(Bt, Kt, Nt)
    on_dev = A3.is_cuda and B3.is_cuda and A3.dtype == torch.float64 and B3.dtype == torch.float64
    if on_dev and Bt == 1 and Kt <= ops.DENSE_MAX_P:
        C3 = ops.gemm_rows(A3[0].contiguous(), B3[0].contiguous(), Nt).unsqueeze(0)
    elif on_dev and Bt > 1 and Mt * Kt * 8 <= 48 * 1024:
        C3 = ops.lcontract(A3.contiguous(), B3.contiguous())
    else:
        C3 = torch.matmul(A3, B3)
    cur = batch + M + N
    C = C3.reshape([size[x] for x in cur])
    return C.permute([cur.index(x) for x in out_labels]) if cur != list(out_labels) else C


class TensorNode:
    def __init__(self, tensor_or_shape, dim_labels, l=None, r=None, name=None, dtype=None):
        """A shape draws randn(shape)/||.||_2 from the global generator, as the
        reference does (tensor/node.py:9-11); a tensor is adopted as is."""
        if isinstance(tensor_or_shape, (tuple, list)):
            t = torch.randn(tuple(tensor_or_shape), dtype=dtype)
            self.tensor = t / torch.norm(t)
        else:
            self.tensor = tensor_or_shape
        self.dim_labels = list(dim_labels)
        self.left_labels = _as_list(l)
        self.right_labels = _as_list(r)
        self.name = name or ''
        self.connections = {}
        self.connection_priority = defaultdict(float)
        self.contracted = set()

    # ---- graph -----------------------------------------------------------
    def reset_connections(self):
        self.connections = {}
        self.connection_priority = defaultdict(float)
        self.contracted = set()

    def connect(self, other_node, labels, priority=float("-inf")):
        """Edge over `labels` in both directions; an existing edge keeps the
        larger priority (tensor/node.py:86-101)."""
        for label in _as_list(labels):
            for a, b in ((self, other_node), (other_node, self)):
                if label in a.connections:
                    a.connection_priority[label] = max(a.connection_priority[label], priority)
                else:
                    a.connection_priority[label] = priority
                a.connections[label] = b

    def get_connecting_labels(self, other_node, horizontal=True):
        """Labels of the edges between the two nodes; a node that is the result of contractions
        stands for the nodes it was made of (tensor/node.py:103-113)."""
        def side(a, b):
            bonds = a.left_labels + a.right_labels
            return [lab for lab, n in a.connections.items() if n is b and (horizontal or lab not in bonds)]
        found = set()
        for a in (self.contracted or {self}):
            for b in other_node.contracted | {other_node}:
                found.update(side(a, b) + side(b, a))
        return list(found)

    def is_horizontal_bond(self, label):
        return label in self.left_labels or label in self.right_labels

    def contract_with(self, other_node, contract_labels=None):
        """Contracts self with other_node over the given labels (default: the labels that connect
        them); labels both nodes carry and that are not contracted act as batch axes
        (tensor/node.py:28-74).  The new node's axes are self's labels then other's, in order,
        without the contracted ones (the reference orders them by set iteration, i.e. arbitrarily).
        On the device the product is one row GEMM (no batch label) or one per-sample contraction
        kernel (batch labels such as the sample axis)."""
        if self is other_node:
            return self
        if contract_labels is None:
            contract_labels = self.get_connecting_labels(other_node)
        contract_labels = _as_list(contract_labels)
        out_labels = [lab for lab in dict.fromkeys(self.dim_labels + other_node.dim_labels) if lab not in contract_labels]
        tensor = _pair_contract(self.tensor, self.dim_labels, other_node.tensor, other_node.dim_labels, contract_labels,
                                out_labels)
        left = [lab for lab in self.left_labels + other_node.left_labels if lab not in contract_labels]
        right = [lab for lab in self.right_labels + other_node.right_labels if lab not in contract_labels]
        node = TensorNode(tensor, out_labels, l=left, r=right,
                          name=f"<{self.name}-{','.join(contract_labels)}-{other_node.name}>")
        node.contracted = self.contracted | other_node.contracted
        if not self.contracted:
            node.contracted.add(self)
        if not other_node.contracted:
            node.contracted.add(other_node)
        for src in (self, other_node):      # edges to the outside survive, with the larger priority
            for label, connected in src.connections.items():
                if connected not in node.contracted:
                    if label in node.connections:
                        node.connection_priority[label] = max(node.connection_priority[label], src.connection_priority[label])
                    else:
                        node.connection_priority[label] = src.connection_priority[label]
                    node.connections[label] = connected
        return node

    def contract_vertically(self, exclude=set()):
        """Contracts every non-bond edge reachable from this node, edges of higher priority first
        (tensor/node.py:149-167)."""
        contracted = self
        queue = [self]
        while queue:
            current = queue.pop(0)
            for label, nxt in sorted(current.connections.items(), key=lambda it: current.connection_priority[it[0]],
                                     reverse=True):
                if nxt in exclude:
                    continue
                if label not in current.left_labels + current.right_labels:
                    contracted = current.contract_with(nxt, nxt.get_connecting_labels(current, horizontal=False))
                    queue.append(contracted)
        return contracted

    def get_vertical_nodes(self, exclude=set()):
        """The nodes contract_vertically would visit, in its order (tensor/node.py:255-278)."""
        order, visited, queue = [], set(), [self]
        while queue:
            current = queue.pop(0)
            if current in visited:
                continue
            visited.add(current)
            order.append(current)
            for label, nxt in sorted(current.connections.items(), key=lambda it: current.connection_priority[it[0]],
                                     reverse=True):
                if nxt in exclude or nxt in visited:
                    continue
                if label not in current.left_labels + current.right_labels:
                    queue.append(nxt)
        return order

    def get_transposed_node(self, exclude=set()):
        """The same tensor under primed labels (tensor/node.py:169-171)."""
        ren = lambda labs: [lab if lab in exclude else f'_{lab}' for lab in labs]   # noqa: E731
        return TensorNode(self.tensor, ren(self.dim_labels), l=ren(self.left_labels), r=ren(self.right_labels),
                          name='_' + self.name)

    # ---- tensor access ---------------------------------------------------
    @property
    def shape(self):
        return self.tensor.shape

    def dim_size(self, label):
        return self.tensor.shape[self.dim_labels.index(label)]

    def cuda(self):
        self.tensor = self.tensor.cuda()
        return self

    def to(self, device=None, dtype=None):
        self.tensor = self.tensor.to(device=device, dtype=dtype)
        return self

    def set_tensor(self, tensor):
        self.tensor = tensor
        return self

    def sum_labels(self, labels):
        axes = [self.dim_labels.index(lab) for lab in _as_list(labels) if lab in self.dim_labels]
        return self.tensor.sum(axes)

    def copy(self):
        """Same tensor, copied label lists, no edges."""
        return TensorNode(self.tensor, list(self.dim_labels), l=list(self.left_labels), r=list(self.right_labels),
                          name=self.name + '_c')

    def squeeze(self, exclude=set()):
        """Drops size-1 axes that are neither excluded nor connected (tensor/node.py:135-147)."""
        drop = [i for i, (n, lab) in enumerate(zip(self.shape, self.dim_labels))
                if n <= 1 and lab not in exclude and lab not in self.connections]
        if drop:
            gone = {self.dim_labels[i] for i in drop}
            self.tensor = self.tensor.squeeze(*drop)
            self.dim_labels = [lab for lab in self.dim_labels if lab not in gone]
            self.left_labels = [lab for lab in self.left_labels if lab not in gone]
            self.right_labels = [lab for lab in self.right_labels if lab not in gone]
        return self

    def _reorder(self, new_labels, inserted):
        have = [lab for lab in new_labels if lab in self.dim_labels]
        perm = [self.dim_labels.index(lab) for lab in have]
        if perm:
            self.tensor = self.tensor.permute(*perm)
        for lab in inserted:
            if lab not in self.dim_labels:
                self.tensor = self.tensor.unsqueeze(new_labels.index(lab))
        self.dim_labels = list(new_labels)
        return self

    def permute_first(self, *labels, expand=True):
        """Given labels first (missing ones become size-1 axes when expand)."""
        order = list(labels) + [lab for lab in self.dim_labels if lab not in labels]
        if not expand:
            order = [lab for lab in order if lab in self.dim_labels]
            return self._reorder(order, [])
        return self._reorder(order, labels)

    def permute_last(self, *labels):
        order = [lab for lab in self.dim_labels if lab not in labels] + list(labels)
        return self._reorder(order, labels)

    def permute(self, *labels):
        self.tensor = self.tensor.permute(*[self.dim_labels.index(lab) for lab in labels])
        self.dim_labels = list(labels)
        return self

    def expand_labels(self, labels, size):
        """Appends missing labels as broadcast axes of the given sizes (tensor/node.py:243-253)."""
        labels = list(labels)
        for lab in labels:
            if lab not in self.dim_labels:
                self.tensor = self.tensor.unsqueeze(-1)
                self.dim_labels = self.dim_labels + [lab]
        self.tensor = self.tensor.expand(*[size[labels.index(lab)] if lab in labels else -1 for lab in self.dim_labels])
        return self

    # ---- core update (tensor/node.py:178-203) ------------------------------
    def update_node(self, step, lr=1.0, adaptive_step=False, min_norm=None, max_norm=None):
        if adaptive_step:
            step_norm, weight_norm = torch.norm(step), torch.norm(self.tensor)
            if step_norm > weight_norm:
                step = step * (weight_norm / step_norm)
        unit = isinstance(lr, (int, float)) and lr == 1.0                              # 1.0 * step is step: one kernel less
        updated = self.tensor + step if unit else self.tensor + lr * step
        if max_norm is not None:
            n = torch.norm(updated)
            if n > max_norm:
                updated = updated * (max_norm / n)
        self.tensor = updated
        return self

    def __repr__(self):
        return f"TensorNode(name={self.name}, shape={tuple(self.shape)}, labels={self.dim_labels})"


class CPDTensorNode(TensorNode):
    """Factor whose bond axes stay diagonal (tensor/node.py:284-298)."""

    def update_node(self, step, lr=1):
        if self.left_labels and self.right_labels:
            n = self.dim_size(self.left_labels[0])
            mask = torch.eye(n, device=self.tensor.device, dtype=self.tensor.dtype)
            for lab in self.dim_labels:
                if not self.is_horizontal_bond(lab):
                    mask = mask.unsqueeze(self.dim_labels.index(lab))
            step = step * mask.expand_as(step)
        self.tensor = self.tensor + lr * step
        return self
