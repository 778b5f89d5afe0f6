from .blocks import PopArt  # noqa: F401
