"""Label helpers (reference: tensor/utils.py:112-120).  The reference's
networkx drawing helper is not part of the core-update path; a stub keeps
`from tensor.utils import visualize_tensornetwork` importable."""
import string


class EinsumLabeler:
    """Hands out one einsum letter per distinct label, in first-use order."""

    def __init__(self):
        self._pool = iter(string.ascii_letters)
        self.mapping = {}

    def __getitem__(self, label):
        letter = self.mapping.get(label)
        if letter is None:
            letter = next(self._pool)
            self.mapping[label] = letter
        return letter


def visualize_tensornetwork(*args, **kwargs):
    raise NotImplementedError("drawing is outside the core-update path of this build")
