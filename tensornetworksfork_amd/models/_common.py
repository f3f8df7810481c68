"""Shared pieces of the two estimators (reference: models/tensor_train.py:11-89, models/tnml.py:25-101)."""
from time import time

import numpy as np
import torch


def root_mean_squared_error_torch(y_true, y_pred):
    """sklearn's root_mean_squared_error (per-output RMSE, uniform average), evaluated
    on the device; one scalar crosses to the host."""
    yt = y_true.reshape(y_true.shape[0], -1).to(torch.float64)
    yp = y_pred.reshape(y_pred.shape[0], -1).to(torch.float64)
    return torch.sqrt(torch.mean((yt - yp) ** 2, dim=0)).mean().item()


def unexplained_variance(y_true, y_pred):
    y_mean = torch.mean(y_true, dim=0, keepdim=True)
    ss_tot = torch.sum((y_true - y_mean) ** 2, dim=1, keepdim=True)
    ss_res = torch.sum((y_true - y_pred) ** 2, dim=1, keepdim=True)
    return (ss_res / ss_tot).mean().item()


def error_rate_torch(y_true, y_pred):
    """1 - accuracy of argmax predictions; y_true one-hot or class indices."""
    pred = torch.argmax(y_pred, dim=1)
    true = torch.argmax(y_true, dim=1) if (y_true.ndim > 1 and y_true.shape[1] > 1) else y_true.reshape(-1).to(pred.dtype)
    return 1.0 - (pred == true).double().mean().item()


class EarlyStopping:
    """Validation loss after every core update, best-weights snapshot and patience
    (models/tensor_train.py:29-89).  Patience is reset only by an improvement of at
    least abs_err or rel_err*|best|; any improvement refreshes the snapshot."""

    def __init__(self, X_val, y_val, model_predict, get_model_weights=None, loss_fn=None, abs_err=0.0, rel_err=0.0,
                 early_stopping=5, verbose=0):
        self.X_val = X_val
        self.y_val = y_val
        self.model_predict = model_predict
        self.get_model_weights = get_model_weights
        self.loss_fn = loss_fn
        self.abs_err = abs_err
        self.rel_err = rel_err
        self.early_stopping = early_stopping
        self.verbose = verbose
        self.early_stop_count = 0
        self.best_val_loss = np.inf
        self.val_history = {}
        self.best_state_dict = self.get_model_weights()
        self.start_time = time()
        self.time_history = {}
        self.epoch = 0

    def convergence_criterion(self):
        elapsed = time() - self.start_time
        self.epoch += 1
        val_loss = self.loss_fn(self.y_val, self.model_predict(self.X_val))
        self.val_history[self.epoch] = val_loss
        self.time_history[self.epoch] = elapsed
        gain = self.best_val_loss - val_loss
        if gain > 0:
            significant = gain >= self.abs_err or gain >= self.rel_err * abs(self.best_val_loss)
            self.best_val_loss = val_loss
            if self.get_model_weights is not None:
                self.best_state_dict = self.get_model_weights()
            self.early_stop_count = 0 if significant else self.early_stop_count + 1
            if self.verbose > 0:
                print(f"Epoch {self.epoch}: Val loss {val_loss:.4f} (diff: {gain:.4f}).")
        else:
            self.early_stop_count += 1
        if self.early_stop_count >= self.early_stopping:
            if self.verbose > 0:
                print(f"Converged with best loss: {self.best_val_loss:.4f}")
            return True
        return False


def to_tensor64(a, device):
    return torch.tensor(a, dtype=torch.float64, device=device) if isinstance(a, np.ndarray) else a


def split_train_val(X, y, seed, validation_split):
    """Shuffled split with np.random.RandomState(seed) (models/tnml.py:174-181)."""
    n = X.shape[0]
    idx = np.arange(n)
    np.random.RandomState(seed).shuffle(idx)
    cut = int(n * (1 - validation_split))
    tr, va = idx[:cut], idx[cut:]
    return X[tr], y[tr], X[va], y[va]
