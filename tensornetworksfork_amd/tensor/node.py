"""Labelled tensors and their graph edges (reference: tensor/node.py).

A node is a `torch.Tensor` plus one label per axis and a dict of edges to other
nodes.  The network (tensor/network.py of this package) reads the labels to
recognise the tensor-train chain and hands the raw device pointers to the HIP
kernels; no contraction is performed by the node itself.
"""
from collections import defaultdict

import torch


def _as_list(x):
    if x is None:
        return []
    return [x] if isinstance(x, str) else list(x)


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
        def side(a, b):
            bonds = a.left_labels + a.right_labels
            return [lab for lab, n in a.connections.items() if n is b and (horizontal or lab not in bonds)]
        return list(set(side(self, other_node) + side(other_node, self)))

    def is_horizontal_bond(self, label):
        return label in self.left_labels or label in self.right_labels

    def contract_with(self, other_node, contract_labels=None):
        raise NotImplementedError("generic pairwise contraction is not part of this build; "
                                  "TensorNetwork contracts recognised chains with HIP kernels")

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
        updated = self.tensor + lr * step
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
