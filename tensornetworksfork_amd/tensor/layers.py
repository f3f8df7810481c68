"""Topology builders with the reference's constructors (reference: tensor/layers.py).

Only the shape / initialisation contract lives here: which nodes exist, their
labels, bond sizes and the order in which random cores are drawn, so that a
seeded layer starts from the same numbers as the reference.
"""
import torch
import torch.nn as nn

from .network import TensorNetwork
from .node import TensorNode


def _bond_pairs(N, r, f, constrict_bond):
    """(left, right) bond per site.  With constrict_bond a bond never exceeds
    what the sites on its smaller side can span: the schedule grows alternately
    from both ends as min(r, previous*f) (tensor/layers.py:17-30,58-75)."""
    if N == 1:
        return [(1, 1)]

    def grow(prev):
        return min(r, prev * f) if constrict_bond else r

    lefts = [(1, grow(1))]
    rights = [(grow(1), 1)]
    pairs = [lefts[0], rights[0]]
    for i in range(N - 2):
        open_left, open_right = lefts[-1][1], rights[0][0]
        if i == N - 3:
            pairs = lefts + [(open_left, open_right)] + rights
        if i % 2 == 0:
            lefts.append((open_left, grow(open_left)))
        else:
            rights.insert(0, (grow(open_right), open_right))
    return pairs


def _perturb_block(rl, f, rr, dtype):
    """Identity-like block acting on the last physical index (tensor/layers.py:32-39)."""
    core = torch.ones(rl, rr, dtype=dtype) if (rl == 1 or rr == 1) else torch.eye(rl, rr, dtype=dtype)
    return torch.cat((torch.zeros(rl, f - 1, rr), core.unsqueeze(1)), dim=1).unsqueeze(1)


class MainNodeLayer(nn.Module):
    def __init__(self, N, r, f, output_shape=tuple(), down_label='p', horizontal_label='r{0}', constrict_bond=True,
                 perturb=False, dtype=None):
        """Cores A1..AN labelled [r_i, c_i|c, p, r_{i+1}] (tensor/layers.py:9-97)."""
        super().__init__()
        output_shape = output_shape if isinstance(output_shape, tuple) else (output_shape,)
        if N == 1:
            r = 1
        self.labels = ['s']
        if perturb:
            first = _perturb_block(1, f, min(r, f) if constrict_bond else r, dtype)
            first = first * (1 + 0.02 * torch.randn_like(first))
            last = _perturb_block(r, f, 1, dtype)
            last = last * (1 + 0.02 * torch.randn_like(last))
            lefts, specs = [first], [first, last]
            for i in range(N - 2):
                rl, rr = lefts[-1].shape[-1], last.shape[0]
                if i == N - 3:
                    specs = lefts + [_perturb_block(rl, f, rr, dtype), last]
                else:
                    lefts.append(_perturb_block(rl, f, min(r, rl * f) if constrict_bond else r, dtype))
            if N == 1:
                specs = [first]
        else:
            specs = _bond_pairs(N, r, f, constrict_bond)
        self.nodes = []
        for i in range(1, N + 1):
            if i - 1 < len(output_shape):
                up, up_label = output_shape[i - 1], f'c{i}'
                self.labels.append(up_label)
            else:
                up, up_label = 1, 'c'
            left_label, right_label = horizontal_label.format(i), horizontal_label.format(i + 1)
            spec = specs[i - 1]
            if not perturb:
                spec = (spec[0], up, f, spec[1])
            self.nodes.append(TensorNode(spec, [left_label, up_label, down_label.format(i), right_label],
                                         l=left_label, r=right_label, name=f"A{i}", dtype=dtype))


class NodeLayer(nn.Module):
    def __init__(self, N, size, labels, name='L{0}', dtype=None):
        super().__init__()
        self.nodes = [TensorNode(size, [lab.format(i) for lab in labels], name=name.format(i), dtype=dtype)
                      for i in range(1, N + 1)]


class InputNodeLayer(NodeLayer):
    def __init__(self, N, f, label='p', dtype=None):
        """Placeholder inputs X1..XN of shape (1, f); they consume random draws like the reference's."""
        super().__init__(N, (1, f), ['s', label], name='X{0}', dtype=dtype)


class TensorNetworkLayer(nn.Module):
    def __init__(self, tensor_network: TensorNetwork = None):
        super().__init__()
        self.set_tensor_network(tensor_network)

    def set_tensor_network(self, tensor_network: TensorNetwork = None):
        self.tensor_network = tensor_network
        self.labels = tensor_network.output_labels if tensor_network is not None else None
        self.parametrized = False
        self.nodes = tensor_network.train_nodes if tensor_network is not None else []

    def node_states(self, detach=True):
        """{"tensor_param_i": core i}; cloned unless detach is False (tensor/layers.py:127-135)."""
        return {f"tensor_param_{i}": (n.tensor.detach().clone() if detach else n.tensor)
                for i, n in enumerate(self.tensor_network.train_nodes)}

    def load_node_states(self, tensor_params, set_value=False):
        for i, n in enumerate(self.tensor_network.train_nodes):
            key = f"tensor_param_{i}"
            if key not in tensor_params:
                raise ValueError(f"Missing parameter: {key}")
            if set_value:
                n.tensor = tensor_params[key]
            else:
                n.tensor.data.copy_(tensor_params[key].detach().clone())
        self.tensor_network.reset_stacks()

    def cuda(self, *args, **kwargs):
        self.tensor_network.cuda()
        return super().cuda(*args, **kwargs)

    def to(self, *args, **kwargs):
        self.tensor_network.to(*args, **kwargs)
        return super().to(*args, **kwargs)

    def cpu(self, *args, **kwargs):
        self.tensor_network.to('cpu')
        return super().cpu(*args, **kwargs)

    def forward(self, x, to_tensor=True):
        out = self.tensor_network.forward(x)
        if self.labels is not None:
            out.permute_first(*self.labels)
        return out.tensor if to_tensor else out

    def num_parameters(self):
        return sum(n.tensor.numel() for n in self.tensor_network.train_nodes)

    def zip_connect(self, nodes1, nodes2, label='p', priority=-1):
        if len(nodes1) != len(nodes2):
            raise ValueError("The number of nodes in both lists must be the same.")
        for i, (a, b) in enumerate(zip(nodes1, nodes2), 1):
            a.connect(b, label.format(i), priority=priority)

    def horizontal_connect(self, nodes):
        for a, b in zip(nodes[:-1], nodes[1:]):
            if a.right_labels and b.left_labels and a.right_labels[0] != b.left_labels[0]:
                raise ValueError(f"Right label of the first node does not match left label of the second node. "
                                 f"Nodes: {a.name}, {b.name}")
            a.connect(b, a.right_labels[0], priority=1)


class TensorTrainLayer(TensorNetworkLayer):
    def __init__(self, num_carriages, bond_dim, input_features, output_shape=tuple(), squeeze=True, constrict_bond=True,
                 perturb=False, dtype=None, seed=None):
        """Plain tensor train: cores first, then placeholder inputs, under one seed (tensor/layers.py:194-221)."""
        super().__init__()
        self.num_carriages = num_carriages
        self.bond_dim = bond_dim
        self.input_features = input_features
        self.output_shape = output_shape if isinstance(output_shape, tuple) else (output_shape,)
        if seed is not None:
            torch.manual_seed(seed)
            if torch.cuda.is_available():
                torch.cuda.manual_seed(seed)
        self.main_node_layer = MainNodeLayer(num_carriages, bond_dim, input_features, output_shape=output_shape,
                                             down_label='p{0}', constrict_bond=constrict_bond, perturb=perturb,
                                             dtype=dtype)
        self.horizontal_connect(self.main_node_layer.nodes)
        self.input_node_layer = InputNodeLayer(num_carriages, input_features, label='p{0}', dtype=dtype)
        self.zip_connect(self.input_node_layer.nodes, self.main_node_layer.nodes, label='p{0}')
        if squeeze:
            for n in self.main_node_layer.nodes:
                n.squeeze(self.main_node_layer.labels)
        self.set_tensor_network(TensorNetwork(self.input_node_layer.nodes, self.main_node_layer.nodes,
                                              output_labels=self.main_node_layer.labels))


class TensorConvolutionTrainLayer(TensorNetworkLayer):
    def __init__(self, num_carriages, bond_dim, num_patches, patch_pixels, output_shape, ring=False,
                 convolution_bond=-1, dtype=None, constrict_bond=True, perturb=False):
        """Tensor train over patches whose per-patch feature is a learned pixel filter
        (tensor/layers.py:791-890).  Column i: X_i (s, patches, patch_pixels) -- C_i (patch_pixels) or
        (CB_i, patch_pixels, CB_{i+1}) on 'patch_pixels', X_i -- A_i (r_i, c_i|c, patches, r_{i+1}) on
        'patches'; A's chained on r, C's on CB.  Random tensors are drawn column by column in the
        order X_i, C_i, A_i from the global generator; size-1 legs are squeezed away.
        main_nodes = A's, train_nodes = [C1, A1, C2, A2, ...]."""
        if ring:
            raise NotImplementedError("Ring structure is not implemented for TensorConvolutionTrainLayer.")
        N, CB = num_carriages, convolution_bond
        self.num_carriages, self.bond_dim = N, bond_dim
        self.num_patches, self.patch_pixels = num_patches, patch_pixels
        self.output_shape = output_shape if isinstance(output_shape, tuple) else (output_shape,)
        self.ring, self.convolution_bond = ring, CB
        n_out = len(self.output_shape)
        if perturb:
            last = _perturb_block(bond_dim, num_patches, 1, dtype)
            first = torch.randn((1, num_patches, bond_dim), dtype=dtype).unsqueeze(1)
            lefts, specs = [first], [first, last]
            for i in range(N - 2):
                rl, rr = lefts[-1].shape[-1], last.shape[0]
                if i == N - 3:
                    specs = lefts + [_perturb_block(rl, num_patches, rr, dtype), last]
                lefts.append(_perturb_block(rl, num_patches, bond_dim, dtype))
        else:
            specs = [(bond_dim if i > 1 else 1, self.output_shape[i - 1] if i <= n_out else 1, num_patches,
                      bond_dim if i < N else 1) for i in range(1, N + 1)]
        labels = ('s',) + tuple(f'c{i}' for i in range(1, min(N, n_out) + 1))
        xs, convs, trains = [], [], []
        for i in range(1, N + 1):
            xs.append(TensorNode((1, num_patches, patch_pixels), ['s', 'patches', 'patch_pixels'], name=f"X{i}"))
            if CB > 0:
                convs.append(TensorNode((CB if i > 1 else 1, patch_pixels, CB if i < N else 1),
                                        [f'CB{i}', 'patch_pixels', f'CB{i + 1}'], l=f'CB{i}', r=f'CB{i + 1}',
                                        name=f"C{i}"))
            else:
                convs.append(TensorNode((patch_pixels,), ['patch_pixels'], name=f"C{i}"))
            trains.append(TensorNode(specs[i - 1], [f'r{i}', f'c{i}' if i <= n_out else 'c', 'patches', f'r{i + 1}'],
                                     l=f'r{i}', r=f'r{i + 1}', name=f"A{i}"))
        order = []
        for xn, cn, an in zip(xs, convs, trains):
            xn.connect(an, 'patches')
            cn.connect(xn, 'patch_pixels')
            order += [cn, an]
        for i in range(1, N):
            trains[i - 1].connect(trains[i], f'r{i + 1}')
            if CB > 0:
                convs[i - 1].connect(convs[i], f'CB{i + 1}')
        for n in trains + convs:
            n.squeeze()
        self.x_nodes, self.conv_blocks, self.train_blocks = xs, convs, trains
        super().__init__(TensorNetwork(xs, trains, order, output_labels=labels))
        self.nodes = order
        self.input_nodes, self.main_nodes = xs, trains
        self.train_nodes = trains + convs


def _seed_all(seed):
    if seed is not None:
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(seed)


class TensorTrainLinearLayer(TensorNetworkLayer):
    def __init__(self, num_carriages, bond_dim, input_features, linear_dim, output_shape=tuple(), squeeze=True,
                 constrict_bond=True, perturb=False, dtype=None, seed=None):
        """Tensor train whose site i sees a learned linear map L_i (lin_i, p_i) of its input
        (tensor/layers.py:308-343): A_i --lin_i-- L_i --p_i-- X_i.  Draw order: cores, maps, inputs;
        train_nodes = [A1, L1, A2, L2, ...]."""
        super().__init__()
        self.num_carriages, self.bond_dim, self.input_features = num_carriages, bond_dim, input_features
        self.output_shape = output_shape if isinstance(output_shape, tuple) else (output_shape,)
        self.linear_dim = linear_dim
        _seed_all(seed)
        self.main_node_layer = MainNodeLayer(num_carriages, bond_dim, linear_dim, output_shape=output_shape,
                                             down_label='lin{0}', constrict_bond=constrict_bond, perturb=perturb,
                                             dtype=dtype)
        self.horizontal_connect(self.main_node_layer.nodes)
        self.linear_layer = NodeLayer(num_carriages, (linear_dim, input_features), labels=('lin{0}', 'p{0}'), dtype=dtype)
        self.zip_connect(self.main_node_layer.nodes, self.linear_layer.nodes, label='lin{0}', priority=2)
        self.input_node_layer = InputNodeLayer(num_carriages, input_features, label='p{0}', dtype=dtype)
        self.zip_connect(self.linear_layer.nodes, self.input_node_layer.nodes, label='p{0}', priority=1)
        if squeeze:
            for n in self.main_node_layer.nodes:
                n.squeeze(self.main_node_layer.labels)
        train = [n for pair in zip(self.main_node_layer.nodes, self.linear_layer.nodes) for n in pair]
        self.set_tensor_network(TensorNetwork(self.input_node_layer.nodes, main_nodes=self.main_node_layer.nodes,
                                              train_nodes=train, output_labels=self.main_node_layer.labels))


class OperatorNodeLayer(nn.Module):
    def __init__(self, N, operator, bond_label='b{0}', up_label='u{0}', down_label='d{0}', ring=False, left=None,
                 right=None, dtype=None):
        """Fixed operator nodes O1..ON labelled [b_{i-1}, up_i, down_i, b_i] forming a matrix-product
        operator (tensor/layers.py:368-406): `operator` per site (list) or shared, with optional
        distinct first / last tensors."""
        super().__init__()
        self.nodes = []
        for i in range(N):
            if isinstance(operator, (list, tuple)):
                t = operator[i]
            elif not ring and i == 0 and left is not None:
                t = left
            elif not ring and i == N - 1 and right is not None:
                t = right
            else:
                t = operator
            lb = bond_label if ring and i == 0 else bond_label.format(i)
            rb = bond_label if ring and i == N - 1 else bond_label.format(i + 1)
            self.nodes.append(TensorNode(t, [lb, up_label.format(i + 1), down_label.format(i + 1), rb], l=lb, r=rb,
                                         name=f"O{i + 1}", dtype=dtype))
        self.labels = []


def get_cum_sum_operator(n, num_carriages, input_features, dtype=None):
    """Site n of the cumulative-sum operator (tensor/layers.py:408-423):
    O[a, k, l, m] = [a <= k] [k == l] [m == k]  (m collapses to 0 on the last site, a to 0 on the first)."""
    f = input_features
    left = 1 if n == 0 else f
    right = 1 if n == num_carriages - 1 else f
    op = torch.zeros((left, f, f, right), dtype=dtype)
    for k in range(f):
        op[:min(left, k + 1), k, k, 0 if right == 1 else k] = 1
    return op


class CumSumLayer(TensorNetworkLayer):
    def __init__(self, num_carriages, bond_dim, input_features, output_shape=tuple(), squeeze=True, constrict_bond=True,
                 perturb=False, dtype=None, seed=None):
        """Tensor train reading its inputs through the fixed cumulative-sum operator
        (tensor/layers.py:425-477): A_i --p_i-- O_i --d_i-- X_i, O's chained on b.  The constructor
        takes `seed` but, like the reference, does not seed the generator."""
        super().__init__()
        self.num_carriages, self.input_features = num_carriages, input_features
        self.main_node_layer = MainNodeLayer(num_carriages, bond_dim, input_features, output_shape=output_shape,
                                             down_label='p{0}', constrict_bond=constrict_bond, perturb=perturb,
                                             dtype=dtype)
        self.horizontal_connect(self.main_node_layer.nodes)
        if num_carriages > 1:
            ops_ = [get_cum_sum_operator(n, num_carriages, input_features, dtype=dtype)
                    for n in (0, 1, num_carriages - 1)]
            self.operator_layer = OperatorNodeLayer(num_carriages, operator=ops_[1], left=ops_[0], right=ops_[2],
                                                    up_label='p{0}', down_label='d{0}', dtype=dtype)
            self.input_node_layer = InputNodeLayer(num_carriages, input_features, label='d{0}', dtype=dtype)
            self.horizontal_connect(self.operator_layer.nodes)
            self.zip_connect(self.main_node_layer.nodes, self.operator_layer.nodes, label='p{0}', priority=2)
            self.zip_connect(self.operator_layer.nodes, self.input_node_layer.nodes, label='d{0}', priority=1)
            for n in self.operator_layer.nodes:
                n.squeeze()
        else:
            self.input_node_layer = InputNodeLayer(num_carriages, input_features, label='p{0}', dtype=dtype)
            self.zip_connect(self.input_node_layer.nodes, self.main_node_layer.nodes, label='p{0}', priority=1)
        if squeeze:
            for n in self.main_node_layer.nodes:
                n.squeeze(self.main_node_layer.labels)
        self.set_tensor_network(TensorNetwork(input_nodes=self.input_node_layer.nodes,
                                              main_nodes=self.main_node_layer.nodes,
                                              train_nodes=self.main_node_layer.nodes,
                                              output_labels=self.main_node_layer.labels))


class CPDLayer(TensorNetworkLayer):
    def __init__(self, num_factors, rank, input_features, output_shape=tuple(), perturb=False, seed=None):
        """Canonical polyadic model (tensor/layers.py:1549-1625): inputs X_i (s, p) are drawn first,
        then factors A_1 (b, p, o) -- (p, o) when it is the only factor -- and A_i (b, p); every
        X_i--A_i edge is labelled 'p' and the rank label 'b' is shared without edges.  `perturb`
        starts factors 2..N as selectors of the last feature (the last one with 2 % noise)."""
        from .network import CPDNetwork
        self.num_factors, self.rank, self.input_features = num_factors, rank, input_features
        self.output_shape = output_shape if isinstance(output_shape, tuple) else (output_shape,)
        _seed_all(seed)
        self.x_nodes = [TensorNode((1, input_features), ['s', 'p'], name=f"X{i}") for i in range(1, num_factors + 1)]
        nodes = []
        for i in range(1, num_factors + 1):
            if i == 1:
                out_dim = self.output_shape[0] if self.output_shape else 1
                if num_factors == 1:
                    nodes.append(TensorNode((input_features, out_dim), ['p', 'o'], name="A1"))
                else:
                    nodes.append(TensorNode((rank, input_features, out_dim), ['b', 'p', 'o'], name="A1"))
                continue
            spec = (rank, input_features)
            if perturb:
                last = torch.ones(rank, 1)
                if i == num_factors:
                    last = last + 0.02 * torch.randn(rank, 1)
                spec = torch.cat((torch.zeros(rank, input_features - 1), last), dim=1)
            nodes.append(TensorNode(spec, ['b', 'p'], name=f"A{i}"))
        for x, a in zip(self.x_nodes, nodes):
            x.connect(a, 'p')
        super().__init__(CPDNetwork(self.x_nodes, nodes, output_labels=('s', 'o'), sample_dim='s'))
        self.nodes = nodes
        self.labels = ['s', 'o']
