"""upfirdn2d on the HIP kernel hf_upfirdn2d_f32, through the registered operator
`torch.ops.hairfast.upfirdn2d` (hairfastgan_amd/ops.py).

Interface of the reference's models/stylegan2/op/upfirdn2d.py:145-156:
`upfirdn2d(input [N,C,H,W], kernel [kh,kw], up=1, down=1, pad=(pad0, pad1))`, the
same pad pair on both axes.  Forward only.
"""
import torch

from ... import ops  # noqa: F401  (registers torch.ops.hairfast.*)
from ..._runtime import require_gpu


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    require_gpu(input, kernel)
    return torch.ops.hairfast.upfirdn2d(input, kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
