"""CLIP ViT-B/32 image tower on the MI355X kernels - SURVEY.md section 8 row f4.

What `ClipBlendingModel.get_image_embed` calls (models/Encoders.py:75-94): `clip_model.encode_image`, the
`VisionTransformer` of the reference's un-vendored dependency `clip @ git+https://github.com/openai/CLIP@a1d0717`
(requirements.txt:6; clip/model.py).  That source is not part of the reference tree, so this module follows the
PUBLISHED architecture (ViT-B/32: 224^2 input, 32^2 patches, width 768, 12 layers, 12 heads, projection to 512) with
the state-dict keys of the OpenAI model's `visual.*` entries (`visual.conv1.weight`, `visual.class_embedding`,
`visual.transformer.resblocks.3.attn.in_proj_weight`, `visual.ln_post.bias`, `visual.proj`, ...) - parity UNPINNED
against the reference (see oracle/ref_clip.py), tested against a torch restatement that is itself checked against
torch's nn.MultiheadAttention / nn.LayerNorm.

Layout: activations are FEATURE-MAJOR, x[feature][token] with token = image * 50 + position - an NCHW tensor
[1, C, images, 50] whose pixels are the tokens - so that every Linear is a 1x1-conv GEMM on the fp16 matrix cores
(hf_conv1x1_f16_f32, csrc/gemm_h.hip, in the process-wide operand mode: f16x3 = fp32-class; weights streamed once per
call, split-K when the token count is small, bias / residual in its epilogue) and nothing is transposed between layers; LayerNorm,
the 12-head attention core and QuickGELU are kernels on the same layout (csrc/vit.hip).  The patch embedding (a
32x32 / stride-32 convolution) is the same GEMM on unfolded patches.  fp32 tensors and accumulation (the reference runs
this tower in fp16).
"""
import torch
from torch import nn

from . import _marshal as M
from ._runtime import lib, require_gpu, stream
from .encoders._fused import FrozenPlanMixin, PreparedConv, conv


class _Attention(nn.Module):  # nn.MultiheadAttention's parameter layout
    def __init__(self, width):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width).normal_(0, width ** -0.5))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = nn.Linear(width, width)


class ResidualAttentionBlock(nn.Module):  # clip/model.py ResidualAttentionBlock
    def __init__(self, width, heads):
        super().__init__()
        self.heads = heads
        self.attn = _Attention(width)
        self.ln_1 = nn.LayerNorm(width)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(width, 4 * width))
        self.mlp.add_module("gelu", nn.Identity())  # QuickGELU: no parameters
        self.mlp.add_module("c_proj", nn.Linear(4 * width, width))
        self.ln_2 = nn.LayerNorm(width)
        self._plan = None

    def forward(self, x, images, seq):
        """x [1, width, images, seq] feature-major."""
        L, st = lib(), stream()
        if self._plan is None:
            prep = lambda w: PreparedConv(M.conv_prepare(L, st, w.detach().reshape(w.shape[0], w.shape[1], 1, 1).contiguous()), 1)  # noqa: E731
            self._plan = {"qkv": prep(self.attn.in_proj_weight), "out": prep(self.attn.out_proj.weight),
                          "fc": prep(self.mlp.c_fc.weight), "proj": prep(self.mlp.c_proj.weight)}
        p = self._plan
        h = M.channel_layernorm(L, st, x, self.ln_1.weight.detach(), self.ln_1.bias.detach(), self.ln_1.eps)
        qkv = conv(h, p["qkv"], 1, 1, bias=self.attn.in_proj_bias.detach())
        a = M.mha_small(L, st, qkv, images, seq, self.heads)
        x = conv(a, p["out"], 1, 1, bias=self.attn.out_proj.bias.detach(), residual=x)
        h = M.channel_layernorm(L, st, x, self.ln_2.weight.detach(), self.ln_2.bias.detach(), self.ln_2.eps)
        h = conv(h, p["fc"], 1, 1, bias=self.mlp.c_fc.bias.detach(), act=M.ACT_QGELU)  # QuickGELU in the GEMM's epilogue
        return conv(h, p["proj"], 1, 1, bias=self.mlp.c_proj.bias.detach(), residual=x)


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])


class VisionTransformer(FrozenPlanMixin, nn.Module):  # clip/model.py VisionTransformer
    def __init__(self, input_resolution=224, patch_size=32, width=768, layers=12, heads=12, output_dim=512):
        super().__init__()
        self.input_resolution, self.patch_size, self.width, self.output_dim = input_resolution, patch_size, width, output_dim
        scale = width ** -0.5
        n_tok = (input_resolution // patch_size) ** 2 + 1
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(n_tok, width))
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self.conv1 = nn.Conv2d(3, width, patch_size, patch_size, bias=False)
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = _Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self._plan = None

    @torch.inference_mode()
    def forward(self, image, taps=None):
        """image [B,3,R,R], CLIP-normalised -> [B, output_dim]."""
        require_gpu(image)
        L, st = lib(), stream()
        b, _, r, _ = image.shape
        ps, wd = self.patch_size, self.width
        if r != self.input_resolution:
            raise ValueError(f"the tower takes {self.input_resolution}^2 images")
        if self._plan is None:
            w = self.conv1.weight.detach().reshape(wd, 3 * ps * ps, 1, 1).contiguous()
            self._plan = {"embed": PreparedConv(M.conv_prepare(L, st, w), 1),
                          "proj": PreparedConv(M.conv_prepare(L, st, self.proj.detach().t().reshape(self.output_dim, wd, 1, 1).contiguous()), 1)}
        p = self._plan
        g = r // ps
        seq = g * g + 1
        # patches, feature-major: [1, 3*ps*ps, B, g*g] (glue: a re-layout of the input image)
        cols = image.float().reshape(b, 3, g, ps, g, ps).permute(1, 3, 5, 0, 2, 4).reshape(1, 3 * ps * ps, b, g * g).contiguous()
        emb = conv(cols, p["embed"], 1, 1)                                               # [1, width, B, g*g]
        x = emb.new_empty((1, wd, b, seq))
        x[0, :, :, 0] = self.class_embedding.detach()[:, None]
        x[0, :, :, 1:] = emb[0]
        x = x + self.positional_embedding.detach().t()[None, :, None, :]
        x = M.channel_layernorm(L, st, x, self.ln_pre.weight.detach(), self.ln_pre.bias.detach(), self.ln_pre.eps)
        for i, blk in enumerate(self.transformer.resblocks):
            x = blk(x, b, seq)
            if taps is not None:
                taps[i] = x
        cls = x[:, :, :, 0:1].contiguous()                                               # [1, width, B, 1]
        cls = M.channel_layernorm(L, st, cls, self.ln_post.weight.detach(), self.ln_post.bias.detach(), self.ln_post.eps)
        out = conv(cls, p["proj"], 1, 1)                                                 # [1, out_dim, B, 1]
        return out[0, :, :, 0].t().contiguous()


class ClipImageTower(nn.Module):
    """`clip_model` as far as HairFast uses it: `.visual` and `encode_image` (clip/model.py CLIP.encode_image)."""

    def __init__(self, **sizes):
        super().__init__()
        self.visual = VisionTransformer(**sizes)

    def load_clip_state_dict(self, state):
        """Takes the OpenAI model's state dict (or ClipBlendingModel's, keys `clip_model.visual.*`): the `visual.*`
        entries are loaded, the text tower's are ignored."""
        own = {}
        for k, v in state.items():
            k = k[len("clip_model."):] if k.startswith("clip_model.") else k
            if k.startswith("visual."):
                own[k] = v.float()
        return self.load_state_dict(own)

    def encode_image(self, image):
        return self.visual(image)

    __call__ = encode_image
