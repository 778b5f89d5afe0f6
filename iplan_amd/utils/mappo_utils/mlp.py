from .blocks import MLPBase, MLPLayer  # noqa: F401  (reference import path utils/mappo_utils/mlp.py)
