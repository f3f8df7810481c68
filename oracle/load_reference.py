"""Import the reference implementation IN THIS CONTAINER ONLY (test infrastructure).

The reference (/root/reference, read-only, Python >= 3.11 syntax) is loaded
from its source text into module objects; nothing of it is copied into this
repository.  Only `oracle/gen_golden.py` uses this loader, to generate the
small input/output fixtures under tests/golden/.  The GPU box has no
/root/reference, so nothing that runs there may import this file.

One expression at /root/reference/tensor/network.py:1012 uses a starred
subscript (PEP 646) that Python 3.10 rejects; it is rewritten in memory to the
equivalent tuple form before compilation.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("TTK_REFERENCE_ROOT", "/root/reference")

_PEP646_OLD = "x[..., *[slice(0, b.shape[i]) for i in range(1, b.tensor.ndim)]]"
_PEP646_NEW = "x[(Ellipsis, *[slice(0, b.shape[i]) for i in range(1, b.tensor.ndim)])]"


def load_reference(root=REF_ROOT):
    """Returns {'tensor': pkg, 'network': mod, 'node': mod, 'layers': mod,
    'bregman': mod, 'tnml': mod, 'tensor_train': mod}."""
    if not os.path.isdir(root):
        raise FileNotFoundError(f"reference tree not found at {root}")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    # The build's own drop-in packages are also called `tensor` / `models`;
    # make sure the names resolve to the reference while this loader is used.
    for name in [m for m in sys.modules if m == "tensor" or m.startswith("tensor.")
                 or m == "models" or m.startswith("models.")]:
        del sys.modules[name]
    pkg = types.ModuleType("tensor")
    pkg.__path__ = [os.path.join(root, "tensor")]
    sys.modules["tensor"] = pkg
    mods = {"tensor": pkg}
    for name in ["utils", "node", "network", "data_compression", "bregman", "layers"]:
        path = os.path.join(root, "tensor", name + ".py")
        with open(path) as f:
            src = f.read()
        if name == "network":
            assert _PEP646_OLD in src, "reference network.py changed; patch no longer applies"
            src = src.replace(_PEP646_OLD, _PEP646_NEW)
        mod = types.ModuleType("tensor." + name)
        mod.__file__ = path
        mod.__package__ = "tensor"
        sys.modules["tensor." + name] = mod
        exec(compile(src, path, "exec"), mod.__dict__)
        setattr(pkg, name, mod)
        mods[name] = mod
    mpkg = types.ModuleType("models")
    mpkg.__path__ = [os.path.join(root, "models")]
    sys.modules["models"] = mpkg
    import importlib
    mods["tnml"] = importlib.import_module("models.tnml")
    mods["tensor_train"] = importlib.import_module("models.tensor_train")
    return mods


def unload_reference():
    for name in [m for m in sys.modules if m == "tensor" or m.startswith("tensor.")
                 or m == "models" or m.startswith("models.")]:
        del sys.modules[name]
