"""Shared plumbing of the encoder mirrors: parameter containers are plain torch.nn layers
(so that state-dict keys equal the reference's), the arithmetic goes through the C ABI
with inference-mode BatchNorm folded into per-channel affines once per checkpoint."""
import torch
from torch import nn

from .. import _marshal as M
from .._runtime import lib, require_gpu, stream


class FrozenPlanMixin:
    """Derived tensors (re-laid-out conv weights, folded BatchNorm) are cached per module
    and dropped whenever the module is moved / cast / re-loaded.  Parameters are frozen
    in HairFast (inference only); after an in-place edit call `invalidate()`."""

    def invalidate(self):
        for m in self.modules():
            if hasattr(m, "_plan"):
                m._plan = None

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.invalidate()
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self.invalidate()
        return out


def fold_bn(bn, conv_bias=None):
    """BatchNorm2d (running statistics) -> (scale, shift) device vectors."""
    require_gpu(bn.weight)
    return M.bn_fold(lib(), stream(), bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps,
                     None if conv_bias is None else conv_bias.detach())


def prep_conv(conv):
    require_gpu(conv.weight)
    return M.conv_prepare(lib(), stream(), conv.weight.detach())


def conv(x, wt, k, stride=1, **kw):
    return M.conv2d(lib(), stream(), x, wt, k, stride, **kw)
