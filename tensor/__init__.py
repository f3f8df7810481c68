"""Import shim: the reference exposes its engine as the top-level package `tensor`."""
