"""CPU-side checks: library exports, layer construction, sweep bookkeeping."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests import _golden as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from tensornetworksfork_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libttk.so missing: run __graft_entry__.build()"
    header = open(os.path.join(ROOT, "include", "ttk.h")).read()
    declared = set(re.findall(r"\b(ttk_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.ttk_version.restype = ctypes.c_int
    assert lib.ttk_version() >= 100


def test_cpu_tensors_are_rejected_loudly():
    from tensornetworksfork_amd import ops, _lib
    t = torch.zeros(4, 1, 3, dtype=torch.float64)
    with pytest.raises(_lib.TTKError):
        ops.env_step_left(t, torch.zeros(4, 2, dtype=torch.float64), torch.zeros(3, 2, 3, dtype=torch.float64))


@pytest.mark.parametrize("name", G.names("tt_core_update_") + G.names("swipe_trace_"))
def test_seeded_layers_match_reference_cores(name):
    from tensornetworksfork_amd.tensor.layers import TensorTrainLayer
    torch.set_default_dtype(torch.float64)
    d = G.load(name)
    cb = bool(int(d["constrict"])) if "constrict" in d else True
    layer = TensorTrainLayer(int(d["N"]), int(d["r"]), int(d["p"]), output_shape=int(d["C"]), constrict_bond=cb,
                             seed=int(d["seed"]))
    tn = layer.tensor_network
    assert tn._engine.cols is not None, tn._engine.error
    for n, c in zip(tn.main_nodes, G.cores(d)):
        assert tuple(n.tensor.shape) == tuple(c.shape) and torch.equal(n.tensor, c)
    assert list(tn.output_labels) == ['s', 'c1']
    assert [n.name for n in tn.train_nodes] == [f"A{i + 1}" for i in range(int(d["N"]))]


def test_headline_bond_schedule():
    from tensornetworksfork_amd.tensor.layers import TensorTrainLayer
    layer = TensorTrainLayer(16, 32, 6, output_shape=1, constrict_bond=True, seed=42)
    shapes = [tuple(n.tensor.shape) for n in layer.tensor_network.main_nodes]
    assert shapes == [(1, 6, 6), (6, 6, 32)] + [(32, 6, 32)] * 12 + [(32, 6, 6), (6, 6)]
    assert layer.num_parameters() == 76104


def test_node_states_roundtrip_and_permutes():
    from tensornetworksfork_amd.tensor.layers import TensorTrainLayer
    layer = TensorTrainLayer(3, 3, 4, output_shape=2, seed=1)
    st = layer.node_states()
    assert sorted(st) == ["tensor_param_0", "tensor_param_1", "tensor_param_2"]
    layer.tensor_network.train_nodes[1].tensor = torch.zeros_like(st["tensor_param_1"])
    layer.load_node_states(st, set_value=True)
    assert torch.equal(layer.tensor_network.train_nodes[1].tensor, st["tensor_param_1"])
    n = layer.tensor_network.main_nodes[1]
    labels = list(n.dim_labels)
    n.permute_last(labels[0])
    assert n.dim_labels == labels[1:] + labels[:1]
    n.permute_first(labels[0])
    assert n.dim_labels == labels


def test_unrecognised_topology_fails_loudly():
    from tensornetworksfork_amd.tensor.node import TensorNode
    from tensornetworksfork_amd.tensor.network import TensorNetwork
    a = TensorNode((2, 3, 2), ['r1', 'p1', 'r2'], l='r1', r='r2', name='A1')
    x = TensorNode((1, 3), ['s', 'p1'], name='X1')
    extra = TensorNode((3,), ['q'], name='C1')
    a.connect(x, 'p1')
    x.connect(extra, 'q')
    tn = TensorNetwork([x], [a])
    with pytest.raises(NotImplementedError):
        tn.forward(torch.zeros(2, 3))


def test_early_stopping_patience():
    from tensornetworksfork_amd.models._common import EarlyStopping
    seq = iter([1.0, 0.5, 0.49999, 0.6, 0.7])
    es = EarlyStopping(None, None, model_predict=lambda X: None, get_model_weights=lambda: {"w": 0},
                       loss_fn=lambda y, p: next(seq), abs_err=1e-2, rel_err=1e-2, early_stopping=3)
    assert [es.convergence_criterion() for _ in range(5)] == [False, False, False, False, True]
    assert es.best_val_loss == 0.49999 and es.epoch == 5
    es0 = EarlyStopping(None, None, model_predict=lambda X: None, get_model_weights=lambda: None,
                        loss_fn=lambda y, p: 1.0, early_stopping=0)
    assert es0.convergence_criterion() is True


def test_feature_maps_match_oracle():
    from tensornetworksfork_amd.models.tnml import fbasis, polynomial_basis
    from oracle import tt_oracle as O
    X = torch.rand(17, 3, dtype=torch.float64) * 2 - 1
    for a, b in zip(polynomial_basis(X, 5), O.polynomial_basis(X, 5)):
        assert torch.equal(a, b)
    for a, b in zip(fbasis(X), O.fbasis(X)):
        assert torch.equal(a, b)


def test_split_matches_reference_split():
    from tensornetworksfork_amd.models._common import split_train_val
    d = G.load("tnml_fit_poly_r6")
    X = torch.tensor(d["X"])
    y = torch.tensor(d["y"])
    Xt, yt, Xv, yv = split_train_val(X, y, 42, 0.1)
    idx = np.arange(X.shape[0])
    np.random.RandomState(42).shuffle(idx)
    assert torch.equal(Xt, X[idx[:360]]) and torch.equal(Xv, X[idx[360:]])


def test_feature_pair_classes_detection():
    from tensornetworksfork_amd.tensor.engine import pair_classes
    torch.manual_seed(0)
    x = torch.rand(500, dtype=torch.float64) * 2 - 1
    X = torch.stack([x ** d for d in range(6)], dim=-1)
    rep = pair_classes(X)
    iu = torch.triu_indices(6, 6)
    sums = (iu[0] + iu[1]).tolist()
    assert rep is not None and len(set(rep)) == 11
    assert all(sums[b] == sums[rep[b]] and rep[b] <= b and rep[rep[b]] == rep[b] for b in range(21))
    assert pair_classes(torch.rand(500, 5, dtype=torch.float64)) is None
    sc = torch.stack([torch.cos(x), torch.sin(x)], dim=-1)
    assert pair_classes(sc) is None
    assert pair_classes(torch.rand(50, 12, dtype=torch.float64)) is None      # above the class limit: not attempted


def check_pairwise_contraction(device):
    """contract_with / contract_vertically against explicit einsums: a plain column (core with its
    input), a sample-batched pair, a conv column (core - input - filter) and the bookkeeping of
    labels, edges and the `contracted` set (tensor/node.py:28-74,149-167)."""
    import torch
    from tensornetworksfork_amd.tensor.layers import TensorConvolutionTrainLayer, TensorTrainLayer
    from tensornetworksfork_amd.tensor.node import TensorNode
    torch.set_default_dtype(torch.float64)
    layer = TensorTrainLayer(3, 4, 5, output_shape=2, seed=1).to(device)
    tn = layer.tensor_network
    g = torch.Generator().manual_seed(2)
    x = torch.rand(30, 5, generator=g).to(device)
    tn.set_input(x)
    A2, X2 = tn.main_nodes[1], tn.input_nodes[1]
    col = A2.contract_with(X2)
    plab = [lab for lab in A2.dim_labels if lab in X2.dim_labels]
    assert len(plab) == 1 and plab[0] not in col.dim_labels and set(col.dim_labels) == (set(A2.dim_labels) | set(X2.dim_labels)) - set(plab)
    ref = torch.einsum('lpr,sp->lrs', A2.tensor, X2.tensor)
    got = col.tensor.permute([col.dim_labels.index(lab) for lab in [A2.dim_labels[0], A2.dim_labels[2], 's']])
    assert torch.allclose(got, ref, rtol=1e-12, atol=1e-14)
    assert col.contracted == {A2, X2} and col.left_labels == A2.left_labels and col.right_labels == A2.right_labels
    assert set(col.connections.values()) == {tn.main_nodes[0], tn.main_nodes[2]}
    assert A2.contract_vertically().tensor.shape == col.tensor.shape
    # batch label: both operands carry 's'
    E = TensorNode(torch.rand(30, 4, generator=g).to(device), ['s', A2.dim_labels[0]], name='E')
    out = E.contract_with(col, [A2.dim_labels[0]])
    assert out.dim_labels == ['s', A2.dim_labels[2]]
    assert torch.allclose(out.tensor, torch.einsum('sl,lpr,sp->sr', E.tensor, A2.tensor, X2.tensor), rtol=1e-12, atol=1e-14)
    # conv column: core - input - filter
    torch.manual_seed(3)
    conv = TensorConvolutionTrainLayer(2, 3, 6, 4, output_shape=2, convolution_bond=2).to(device).tensor_network
    xc = torch.rand(20, 6, 4, generator=g).to(device)
    conv.set_input(xc)
    A1 = conv.main_nodes[0]
    full = A1.contract_vertically()
    C1 = [n for n in conv.train_nodes if n.name.startswith('C')][0]
    X1 = conv.input_nodes[0]
    ein = torch.einsum('abc,sbx,xd->sacd', A1.tensor.permute([A1.dim_labels.index(lab) for lab in A1.dim_labels]),
                       X1.tensor, C1.tensor) if A1.tensor.ndim == 3 and C1.tensor.ndim == 2 else None
    assert ein is not None and set(full.dim_labels) == {'s', A1.dim_labels[0], A1.dim_labels[2], C1.dim_labels[1]}
    perm = [full.dim_labels.index(lab) for lab in ['s', A1.dim_labels[0], A1.dim_labels[2], C1.dim_labels[1]]]
    assert torch.allclose(full.tensor.permute(perm), ein, rtol=1e-12, atol=1e-14)
    assert [n.name for n in A1.get_vertical_nodes()] == [A1.name, X1.name, C1.name]
    T = A1.get_transposed_node(exclude={'s'})
    assert T.dim_labels == ['_' + lab for lab in A1.dim_labels] and T.tensor is A1.tensor


def test_pairwise_contraction_host():
    check_pairwise_contraction("cpu")


@pytest.mark.gpu
def test_pairwise_contraction_device():
    check_pairwise_contraction("cuda:0")


def test_rhs_route_for_many_feature_sites():
    """Which right-hand-side kernel a Khatri-Rao term goes to: the features-first one only where the site has
    many features and an output wide enough to earn back its fixed cost (the shapes the choice was measured on),
    and only inside the extents its chunk images fit."""
    from tensornetworksfork_amd.tensor.engine import _rhs_features_first
    assert _rhs_features_first(48, 65, 48)          # CIFAR carriage
    assert _rhs_features_first(24, 50, 24)          # MNIST carriage
    assert _rhs_features_first(1, 65, 48)           # edge site with a wide right bond
    assert not _rhs_features_first(48, 65, 1)       # right edge: one pass of the three-factor kernel
    assert not _rhs_features_first(12, 30, 12)
    assert not _rhs_features_first(32, 6, 32)       # few features: never
    assert not _rhs_features_first(48, 300, 48)     # beyond the kernel's feature limit
    assert not _rhs_features_first(48, 200, 100)    # two chunk images would not fit the LDS
    assert not _rhs_features_first(4, 64, 200)      # bond beyond the staged width
