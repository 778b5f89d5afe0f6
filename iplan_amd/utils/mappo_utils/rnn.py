from .blocks import RNNLayer  # noqa: F401
