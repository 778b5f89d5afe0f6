from .blocks import ACTLayer  # noqa: F401
