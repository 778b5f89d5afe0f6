from .blocks import Categorical  # noqa: F401
