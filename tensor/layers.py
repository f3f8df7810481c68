"""`tensor.layers` as the reference's callers import it; the implementation lives in tensornetworksfork_amd.tensor.layers."""
from tensornetworksfork_amd.tensor.layers import *  # noqa: F401,F403
from tensornetworksfork_amd.tensor import layers as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
