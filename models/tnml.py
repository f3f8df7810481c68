"""`models.tnml` as the reference's callers import it; the implementation lives in tensornetworksfork_amd.models.tnml."""
from tensornetworksfork_amd.models import tnml as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
