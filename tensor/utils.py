"""`tensor.utils` as the reference's callers import it; the implementation lives in tensornetworksfork_amd.tensor.utils."""
from tensornetworksfork_amd.tensor.utils import *  # noqa: F401,F403
from tensornetworksfork_amd.tensor import utils as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
