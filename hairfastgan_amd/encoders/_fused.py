"""Shared plumbing of the encoder mirrors: parameter containers are plain torch.nn layers
(so that state-dict keys equal the reference's), the arithmetic goes through the C ABI
with inference-mode BatchNorm folded into per-channel affines once per checkpoint."""
import torch
from torch import nn

from .. import _marshal as M
from .._runtime import lib, require_gpu, stream


class FrozenPlanMixin:
    """Derived tensors (re-laid-out conv weights, folded BatchNorm) are cached per module
    and dropped whenever the module is moved / cast / re-loaded - also when that happens through
    a PARENT container: nn.Module.load_state_dict recurses through _load_from_state_dict, not
    through the children's load_state_dict, but it does run every visited module's
    load-state-dict post-hooks, so each planned module registers one.  Parameters are frozen
    in HairFast (inference only); after an in-place edit of a parameter call `invalidate()`."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.register_load_state_dict_post_hook(lambda module, _incompatible: module.invalidate())

    def invalidate(self):
        for m in self.modules():
            if hasattr(m, "_plan"):
                m._plan = None

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
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
