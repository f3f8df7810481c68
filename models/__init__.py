"""Import shim: the reference exposes its estimators as the top-level package `models`."""
