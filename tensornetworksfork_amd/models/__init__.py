"""Drop-in counterpart of the reference's `models` package (tensor_train, tnml)."""
