"""Device front-ends that produce the per-site input tensors of the models from raw data: the patch rows of
the image scripts and the local feature maps of TNMLRegressor (reference: image_convolution_lanczos_MNIST.py:
29-32, CIFAR_lanczos_swipe.py:35-38, models/tnml.py:11-23)."""
from .. import ops
from ..models.tnml import fbasis, polynomial_basis  # noqa: F401


def patchify(images, kernel_size=4, stride=4, dtype=None):
    """(S, Cin, H, W) device images -> (S, patches + 1, Cin*k*k + 1): what
        x = F.unfold(images, kernel_size=(k, k), stride=(s, s), padding=0).transpose(-2, -1)
        x = torch.cat((x, zeros(S, 1, x.shape[2])), dim=-2); x = torch.cat((x, zeros(S, x.shape[1], 1)), dim=-1)
        x[..., -1, -1] = 1.0
    builds in the image scripts, in one pass over the images (ttk_patchify).  dtype: the output's (default: the
    images'; float32 images may be widened to float64 on the way)."""
    return ops.patchify(images.contiguous(), kernel_size, stride, dtype)
