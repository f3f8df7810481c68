"""Drop-in counterpart of the reference's `tensor` package (node, network, layers, bregman, utils)."""
