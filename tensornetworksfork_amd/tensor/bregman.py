"""Loss objects returning (loss, d_loss, dd_loss) per sample (reference: tensor/bregman.py).

Square, element-wise MSE and softmax cross-entropy are evaluated in closed form
by one HIP kernel (ttk_loss_f64); the class names, constructor arguments and
output shapes are the reference's.  Custom callables wrapped in AutogradLoss /
AutogradBregman keep the reference's autograd evaluation, which is independent
of the device.
"""
import torch
from torch import nn
from torch.nn import functional as F

from .. import ops


def _flat2(t):
    return t.flatten(start_dim=1) if t.ndim > 1 else t.unsqueeze(-1)


class BregFunction(nn.Module):
    """Bregman divergence of psi: loss = psi(x) - psi(y) - <d psi(y), x - y> (tensor/bregman.py:5-32)."""

    def transform_forward(self, x, y):
        return x, y

    def forward(self, x, y):
        x, y = self.transform_forward(x, y)
        loss = self.psi(x) - self.psi(y) - self.prod(self.d(y), x - y)
        return loss, self.grad(x, y), self.hess(x, y)

    def grad(self, x, y):
        return self.d(x) - self.d(y)

    def hess(self, x, y):
        return self.dsq(x)

    def prod(self, x, y):
        return torch.sum(x * y, dim=-1)

    def psi(self, x):
        raise NotImplementedError

    def d(self, x):
        raise NotImplementedError

    def dsq(self, x):
        raise NotImplementedError


class SquareBregFunction(BregFunction):
    """psi(x) = sum x^2: loss (S,), gradient 2(x-y) (S,C), Hessian 2 stored as (S,C,1)
    exactly like the reference (tensor/bregman.py:34-52); get_A_b broadcasts the
    trailing axis over the second output index."""

    def forward(self, x, y):
        x2, y2 = _flat2(x), _flat2(y)
        loss, g, H = ops.loss("square", ops.prep(x2), ops.prep(y2))
        return loss, g.reshape(x2.shape), H[..., :1]


class AutogradLoss(nn.Module):
    """Element-wise loss with gradient and Hessian w.r.t. the model output
    (tensor/bregman.py:266-292).  The default nn.MSELoss(reduction='none') has
    the closed form loss (S,C), 2(x-y), diag(2) and runs as a HIP kernel; any
    other callable is differentiated with autograd like the reference."""

    def __init__(self, loss_func=None):
        super().__init__()
        self._closed_form = loss_func is None or (isinstance(loss_func, nn.MSELoss) and loss_func.reduction == 'none')
        self.loss_func = nn.MSELoss(reduction='none') if loss_func is None else loss_func

    def forward(self, model_out, y_true, only_loss=False):
        if self._closed_form and model_out.ndim == 2 and not only_loss:
            return ops.loss("mse", ops.prep(model_out), ops.prep(y_true))
        model_out = model_out.requires_grad_(True)
        loss = self.loss_func(model_out, y_true)
        if only_loss:
            return loss
        return _autograd_grad_hess(loss, model_out)


def _autograd_grad_hess(loss, x):
    d_loss = torch.autograd.grad(loss.sum(), x, create_graph=True, retain_graph=True)[0]
    rows = [torch.autograd.grad(d_loss[..., i].sum(), x, retain_graph=True, create_graph=True)[0].unsqueeze(-2)
            for i in range(d_loss.shape[-1])]
    return loss.detach(), d_loss.detach(), torch.cat(rows, dim=-2).detach()


class XEAutogradBregman(BregFunction):
    """Cross entropy on logits [w*x, 0] against the label argmax(y), y over C+1 classes
    (tensor/bregman.py:189-216).  The reference differentiates cross_entropy(z, argmax y) twice
    with autograd; the closed form pi = softmax, g = w(pi - onehot(argmax y))[:C],
    H = w^2 (diag(pi) - pi pi^T)[:C,:C] is what that evaluates to, soft labels included."""

    _kind = "xe_hard"

    def __init__(self, w=1.0):
        super().__init__()
        self.w = w

    def forward(self, x, y, only_loss=False):
        if only_loss:
            z = torch.cat((self.w * x, torch.zeros_like(x[..., :1])), dim=-1)
            return F.cross_entropy(z, y.argmax(dim=-1), reduction='none')
        return ops.loss(self._kind, ops.prep(x), ops.prep(y), w=self.w)


class KLDivBregman(XEAutogradBregman):
    """The reference's closed form (tensor/bregman.py:100-146): as above but with the gradient
    against the target probabilities themselves, g = w(pi - y)[:C]."""

    _kind = "xe"

    def __init__(self, w=1.0, grad_clip=1e3):
        super().__init__(w)
        self.grad_clip = grad_clip


class AutogradBregman(BregFunction):
    """Bregman divergence of a user potential, differentiated by autograd
    (tensor/bregman.py:218-263)."""

    def __init__(self, phi_func, forward_transform=None, d_phi_x_func=None):
        super().__init__()
        self.phi_func = phi_func
        self._transform_forward = forward_transform
        self._d_phi_x_func = d_phi_x_func

    def transform_forward(self, x, y):
        if self._transform_forward is not None:
            x, y = self._transform_forward(x, y)
        return x, y

    def forward(self, x, y, only_loss=False):
        x = x.requires_grad_(True)
        x, y = self.transform_forward(x, y)
        phi_x, phi_y = self.phi_func(x), self.phi_func(y)
        if self._d_phi_x_func is not None:
            d_phi_x = self._d_phi_x_func(x)
        else:
            d_phi_x = torch.autograd.grad(phi_x.sum(), x, create_graph=True, retain_graph=True)[0]
        loss = phi_y - phi_x - (d_phi_x * (y - x)).sum(-1, keepdim=True)
        if only_loss:
            return loss
        return _autograd_grad_hess(loss, x)


def closed_form_symmetric_hessian(loss_fn):
    """True when `loss_fn` is one of the closed-form losses above evaluated by ttk_loss_f64 (their per-sample
    Hessians are symmetric by construction: 2 * ones, diag(2), w^2 (diag(pi) - pi pi^T)); False for anything
    differentiated by autograd or overridden by a caller, whose Hessian has to be looked at."""
    for cls in (SquareBregFunction, XEAutogradBregman):
        if isinstance(loss_fn, cls) and type(loss_fn).forward in (SquareBregFunction.forward, XEAutogradBregman.forward):
            return True
    return isinstance(loss_fn, AutogradLoss) and type(loss_fn).forward is AutogradLoss.forward and loss_fn._closed_form
