"""TNMLRegressor: one TT site per input feature with a local feature map
(reference: models/tnml.py)."""
import math

import numpy as np
import torch
from sklearn.base import BaseEstimator, RegressorMixin
from sklearn.metrics import accuracy_score, r2_score

from .. import ops
from ..tensor.bregman import SquareBregFunction
from ..tensor.engine import PAIR_CLASS_MAX_P
from ..tensor.layers import TensorTrainLayer
from ._common import (EarlyStopping, error_rate_torch, root_mean_squared_error_torch, split_train_val, to_tensor64,
                      unexplained_variance)

__all__ = ["TNMLRegressor", "EarlyStopping", "fbasis", "polynomial_basis", "root_mean_squared_error_torch",
           "error_rate_torch", "unexplained_variance"]


def _on_device(X):
    return isinstance(X, torch.Tensor) and X.is_cuda and X.dtype == torch.float64 and X.ndim == 2 and X.shape[0] > 0


def fbasis(X):
    """[cos(pi x/2), sin(pi x/2)] per feature: list of (S, 2) (models/tnml.py:11-16).  A float64 device
    input goes through one kernel for all features (ttk_feature_map_f64)."""
    if _on_device(X):
        return list(ops.feature_map(X.contiguous(), "sincos", 2).unbind(0))
    h = 0.5 * math.pi
    return [torch.stack([torch.cos(h * X[:, i]), torch.sin(h * X[:, i])], dim=-1) for i in range(X.shape[-1])]


def polynomial_basis(X, degree=3):
    """[1, x, ..., x^degree] per feature: list of (S, degree+1) (models/tnml.py:18-23).  A float64 device
    input goes through one kernel for all features; the products x^i x^j of such a basis depend on i + j
    only, which the tensors carry along (`ttk_pair_rep`) so that the Gram kernel's feature-pair classes
    need not be detected from the data."""
    if _on_device(X):
        p = degree + 1
        feats = list(ops.feature_map(X.contiguous(), "poly", p).unbind(0))
        if 2 <= p <= PAIR_CLASS_MAX_P:
            first, rep = {}, []
            for i in range(p):
                for j in range(i, p):
                    rep.append(first.setdefault(i + j, len(rep)))
            for t in feats:
                if rep != list(range(len(rep))):
                    t.ttk_pair_rep = (rep, t._version)
        return feats
    return [torch.stack([X[:, i] ** d for d in range(degree + 1)], dim=-1) for i in range(X.shape[-1])]


class TNMLRegressor(BaseEstimator, RegressorMixin):
    def __init__(self, r=8, output_dim=1, seed=42, device='cuda', bf=None, lr=1.0, eps_start=1.0, eps_decay=0.5,
                 abs_err=1e-6, rel_err=1e-4, batch_size=512, method='ridge_cholesky', num_swipes=30, model_type='tt',
                 task='regression', train_operator=False, early_stopping=0, basis='sin-cos', degree=3, verbose=0,
                 constrict_bond=True):
        self.r = r
        self.input_dim = degree + 1 if basis == 'polynomial' else 2
        self.output_dim = output_dim
        self.constrict_bond = constrict_bond
        self.perturb = False
        self.seed = seed
        self.device = device
        self.bf = bf if bf is not None else SquareBregFunction()
        self.lr = lr
        self.eps = eps_start
        self.eps_decay = eps_decay
        self.abs_err = abs_err
        self.rel_err = rel_err
        self.batch_size = batch_size
        self.method = method
        self.num_swipes = num_swipes
        self.model_type = model_type
        self.task = task
        self.train_operator = train_operator
        self.early_stopping = early_stopping
        self.basis = basis
        self.degree = degree
        self.verbose = verbose
        self._model = None

    def _initialize_model(self):
        self._model = TensorTrainLayer(self.N, self.r, self.input_dim, output_shape=self.output_dim,
                                       constrict_bond=self.constrict_bond, perturb=self.perturb,
                                       seed=self.seed).to(self.device)
        if self.verbose > 2:
            print("Number of parameters:", self._model.num_parameters())

    def _features(self, X):
        return polynomial_basis(X, degree=self.degree) if self.basis == 'polynomial' else fbasis(X)

    def _predict_tensor(self, feats):
        y = self._model.tensor_network.forward_batch(feats, self.batch_size)
        if self.task == 'classification':
            y = torch.cat([y, torch.zeros_like(y[..., :1])], dim=-1)
        return y

    def fit(self, X, y, X_val=None, y_val=None, validation_split=0.1, split_train=True):
        """Feature map, left-orthonormal start, sweeps with QR gauge and early stopping
        (models/tnml.py:157-240)."""
        X, y = to_tensor64(X, self.device), to_tensor64(y, self.device)
        if self._model is None:
            self.N = X.shape[1]
            self._initialize_model()
        if self.verbose > 0:
            print("Number of parameters:", self._model.num_parameters())
        if X_val is None or y_val is None:
            if split_train:
                X_train, y_train, X_val, y_val = split_train_val(X, y, self.seed, validation_split)
            else:
                X_train, y_train, X_val, y_val = X, y, X, y
        else:
            X_val, y_val = to_tensor64(X_val, self.device), to_tensor64(y_val, self.device)
            X_train, y_train = X, y
        F_train, F_val = self._features(X_train), self._features(X_val)
        self._early_stopper = EarlyStopping(
            F_val, y_val, model_predict=self._predict_tensor, get_model_weights=self._model.node_states,
            loss_fn=root_mean_squared_error_torch if self.task == 'regression' else error_rate_torch,
            abs_err=self.abs_err, rel_err=self.rel_err, early_stopping=self.early_stopping, verbose=self.verbose)
        tn = self._model.tensor_network
        tn.orthonormalize_left()
        tn.accumulating_swipe(F_train, y_train, self.bf, batch_size=self.batch_size, lr=self.lr, eps=self.eps,
                              eps_decay=self.eps_decay, convergence_criterion=self._early_stopper.convergence_criterion,
                              orthonormalize=True, method=self.method, verbose=self.verbose,
                              num_swipes=self.num_swipes, skip_second=False, direction='l2r',
                              disable_tqdm=self.verbose < 3)
        if self._early_stopper.best_state_dict is not None:
            self._model.load_node_states(self._early_stopper.best_state_dict, set_value=True)
        return self

    def predict(self, X):
        X = to_tensor64(X, self.device)
        return self._predict_tensor(self._features(X)).detach().cpu().numpy()

    def score(self, X, y_true):
        if not isinstance(y_true, np.ndarray):
            y_true = y_true.cpu().numpy()
        y_pred = self.predict(X)
        if self.task == 'regression':
            return r2_score(y_true, y_pred)
        return accuracy_score(y_true, np.argmax(y_pred, axis=1))
