"""MI355X-native DMRG core-update engine behind the reference's tensor/ and models/ call surface."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
