"""Operator plugin layer; same names as the reference's models/stylegan2/op/__init__.py:1-2."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu  # noqa: F401
from .upfirdn2d import upfirdn2d  # noqa: F401
