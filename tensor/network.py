"""`tensor.network` as the reference's callers import it; the implementation lives in tensornetworksfork_amd.tensor.network."""
from tensornetworksfork_amd.tensor.network import *  # noqa: F401,F403
from tensornetworksfork_amd.tensor import network as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
