"""TensorTrainRegressor: N-site TT over the same bias-augmented feature vector
(reference: models/tensor_train.py)."""
import numpy as np
import torch
from sklearn.base import BaseEstimator, RegressorMixin
from sklearn.metrics import accuracy_score, r2_score

from ..tensor.bregman import SquareBregFunction
from ..tensor.layers import (CPDLayer, CumSumLayer, TensorNetworkLayer, TensorTrainLayer,
                             TensorTrainLinearLayer)
from ..tensor.network import SumOfNetworks
from ._common import (EarlyStopping, error_rate_torch, root_mean_squared_error_torch, split_train_val, to_tensor64,
                      unexplained_variance)

__all__ = ["TensorTrainRegressor", "EarlyStopping", "root_mean_squared_error_torch", "error_rate_torch",
           "unexplained_variance"]


class TensorTrainRegressor(BaseEstimator, RegressorMixin):
    def __init__(self, N=3, r=8, output_dim=1, linear_dim=None, constrict_bond=False, perturb=False, seed=42,
                 device='cuda', bf=None, lr=1.0, eps_start=1.0, eps_decay=0.5, abs_err=1e-4, rel_err=1e-3,
                 batch_size=512, method='ridge_cholesky', num_swipes=30, model_type='tt', task='regression',
                 train_operator=False, cum_sum=False, early_stopping=0, verbose=0):
        self.N = N
        self.r = r
        self.output_dim = output_dim
        self.linear_dim = linear_dim if linear_dim is not None and linear_dim > 0 else None
        self.constrict_bond = constrict_bond
        self.perturb = perturb
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
        self.cum_sum = cum_sum
        self.early_stopping = early_stopping
        self.verbose = verbose
        self._model = None
        if self.perturb and self.output_dim > 1:
            raise ValueError("perturb not supported for output dim > 1")

    def _initialize_model(self):
        """model_type -> topology (models/tensor_train.py:133-208): 'tt*' / 'cpd*' pick the tensor
        train or the CPD model, '*type1*' / '*typeI*' the sum of models of order 1..N (the
        higher orders without the bias feature, seeds seed+i), cum_sum the cumulative-sum
        operator, linear_dim < input_dim a learned linear map per site."""
        if self.input_dim is None:
            raise ValueError("input_dim must be set")
        if isinstance(self.linear_dim, float) and 0 < self.linear_dim < 1:
            self.linear_dim = max(int(self.linear_dim * self.input_dim), 1)
        mt = self.model_type
        type1 = 'type1' in mt or 'typeI' in mt
        tt_kw = dict(output_shape=self.output_dim, constrict_bond=self.constrict_bond, perturb=self.perturb)

        def make(order, features, seed):
            if mt.startswith('cpd'):
                return CPDLayer(order, self.r, features, output_shape=self.output_dim, perturb=self.perturb, seed=seed)
            if self.cum_sum:
                return CumSumLayer(order, self.r, features, seed=seed, **tt_kw)
            if self.linear_dim is None or self.linear_dim >= self.input_dim:
                return TensorTrainLayer(order, self.r, features, seed=seed, **tt_kw)
            return TensorTrainLinearLayer(order, self.r, features, self.linear_dim, seed=seed, **tt_kw)

        if not (mt.startswith('cpd') or mt.startswith('tt')):
            return
        if type1:
            nets = [make(i, self.input_dim - 1 if i != 1 else self.input_dim, self.seed + i).tensor_network
                    for i in range(1, self.N + 1)]
            self._model = TensorNetworkLayer(SumOfNetworks(nets, output_labels=nets[0].output_labels,
                                                           train_operators=self.train_operator)).to(self.device)
        else:
            self._model = make(self.N, self.input_dim, self.seed).to(self.device)
        if self.verbose > 2:
            print("Number of parameters:", self._model.num_parameters())

    def _with_bias(self, X):
        return torch.cat((X, torch.ones((X.shape[0], 1), dtype=torch.float64, device=X.device)), dim=1)

    def _predict_tensor(self, Xb):
        y = self._model.tensor_network.forward_batch(Xb, self.batch_size)
        if self.task == 'classification':
            y = torch.cat([y, torch.zeros_like(y[..., :1])], dim=-1)
        return y

    def fit(self, X, y, X_val=None, y_val=None, validation_split=0.1, split_train=True):
        """Bias column, split, sweeps without gauge fixing, best-weights restore
        (models/tensor_train.py:212-296)."""
        X, y = to_tensor64(X, self.device), to_tensor64(y, self.device)
        X = self._with_bias(X)
        if self._model is None:
            self.input_dim = X.shape[1]
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
            if X_val.shape[1] != X_train.shape[1]:
                X_val = self._with_bias(X_val)
        self._early_stopper = EarlyStopping(
            X_val, y_val, model_predict=self._predict_tensor, get_model_weights=self._model.node_states,
            loss_fn=root_mean_squared_error_torch if self.task == 'regression' else error_rate_torch,
            abs_err=self.abs_err, rel_err=self.rel_err, early_stopping=self.early_stopping, verbose=self.verbose)
        self._model.tensor_network.accumulating_swipe(
            X_train, y_train, self.bf, batch_size=self.batch_size, lr=self.lr, eps=self.eps, eps_decay=self.eps_decay,
            convergence_criterion=self._early_stopper.convergence_criterion, orthonormalize=False, method=self.method,
            verbose=self.verbose, num_swipes=self.num_swipes, skip_second=False, direction='l2r',
            disable_tqdm=self.verbose < 3)
        if self._early_stopper.best_state_dict is not None:
            self._model.load_node_states(self._early_stopper.best_state_dict, set_value=True)
        return self

    def predict(self, X):
        X = to_tensor64(X, self.device)
        return self._predict_tensor(self._with_bias(X)).detach().cpu().numpy()

    def score(self, X, y_true):
        if not isinstance(y_true, np.ndarray):
            y_true = y_true.cpu().numpy()
        y_pred = self.predict(X)
        if self.task == 'regression':
            return r2_score(y_true, y_pred)
        return accuracy_score(y_true, np.argmax(y_pred, axis=1))
