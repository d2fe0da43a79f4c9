"""fused bias + leaky ReLU on the HIP kernel hf_fused_bias_act_f32, through the registered operator
`torch.ops.hairfast.fused_bias_act` (hairfastgan_amd/ops.py).

Interface of the reference's models/stylegan2/op/fused_act.py:73-96 (module
`FusedLeakyReLU(channel)` with parameter `.bias [C]`; function
`fused_leaky_relu(input, bias, negative_slope=0.2, scale=sqrt(2))`).  Forward only
(the pipeline runs under inference_mode); unlike the reference's CPU branch, which
hard-codes 0.2 (fused_act.py:90), `negative_slope` is honoured as in its CUDA branch.
"""
import torch
from torch import nn

from ... import ops  # noqa: F401  (registers torch.ops.hairfast.*)
from ..._runtime import require_gpu


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    require_gpu(input, bias)
    return torch.ops.hairfast.fused_bias_act(input, None if bias is None else bias.detach(), float(negative_slope), float(scale))


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
