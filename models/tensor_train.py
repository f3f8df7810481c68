"""`models.tensor_train` as the reference's callers import it; the implementation lives in tensornetworksfork_amd.models.tensor_train."""
from tensornetworksfork_amd.models import tensor_train as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
