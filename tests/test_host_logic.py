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
