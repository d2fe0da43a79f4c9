"""FeatureStyleEncoder on the MI355X kernels - host-side mirror of
models/FeatureStyleEncoder/nets/feature_style_encoder.py:12-65 (`fs_encoder_v2`),
arcface/iresnet.py:28-57 (`IBasicBlock`) and the part of trainer.py:273-297, 357-365
(`Trainer.test(img=..., return_latent=True)`) that Embedding.py:74-76 consumes.

Same parameter names as the reference (517 state-dict entries).  An IBasicBlock is 2-3
launches of the fused conv kernel (BN before the conv on the staged activations, BN after
+ PReLU + residual add in the epilogue); the 18 style heads are one weight-streaming GEMV
batch over the concatenated [18*512, 8640] matrix.  Inference only.

The reference's Trainer.test additionally runs its private StyleGAN2 copy on the encoded
latents and the caller throws the image away (trainer.py:295, Embedding.py:75-76; 148.5
GFLOP per image whose only effect is advancing the RNG).  `FSEncoder.test` keeps the
return convention `[x_1, x_1_recon, w_recon, fea_1]` but leaves `x_1_recon` as None unless
a generator is attached with `run_discarded_generator=True`.
"""
import torch
from torch import nn

from .. import _marshal as M
from .._runtime import lib, reference_rng_walk, require_gpu, stream
from ._fused import FrozenPlanMixin, chain_takes_split, conv, conv_pair, fold_bn, prep_conv

_IRESNET50 = [(64, 3), (128, 4), (256, 14), (512, 3)]  # (planes, blocks) per layer, arcface/iresnet.py iresnet50


class IBasicBlock(FrozenPlanMixin, nn.Module):  # iresnet.py:28-57
    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(inplanes, eps=1e-05)
        self.conv1 = nn.Conv2d(inplanes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, eps=1e-05)
        self.prelu = nn.PReLU(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes, eps=1e-05)
        self.downsample = downsample
        self.stride = stride
        self._plan = None

    def _prepared(self):
        if self._plan is None:
            p = {"bn1": fold_bn(self.bn1), "w1": prep_conv(self.conv1), "bn2": fold_bn(self.bn2),
                 "slope": self.prelu.weight.detach(), "w2": prep_conv(self.conv2), "bn3": fold_bn(self.bn3)}
            if self.downsample is not None:
                p["wd"] = prep_conv(self.downsample[0])
                p["bnd"] = fold_bn(self.downsample[1])
            self._plan = p
        return self._plan

    def forward(self, x):
        return self.forward_chain(x, None, None)[0]

    def takes_split(self, h, wd):
        """Would this block's first conv accept its input pre-split (handed over by the previous block's second conv)?"""
        p = self._prepared()
        return chain_takes_split(p["w1"], h, wd, out_scale=p["bn2"][0], bias=p["bn2"][1], act=M.ACT_PRELU, slope=p["slope"])

    def forward_chain(self, x, xs, nxt):
        """The block on x; xs: x in the first conv's pre-split layout with THIS block's bn1 applied (the previous block's
        hand-over) or None; nxt: the next block when its first conv takes such a hand-over.  -> (out, SplitActivation of the
        next block's bn1(out) | None): the second conv's epilogue (bn3 + residual) writes both.  Same bits as block after block."""
        require_gpu(x)
        p = self._prepared()
        identity = x
        if "wd" in p:
            identity = conv(x, p["wd"], 1, self.stride, out_scale=p["bnd"][0], bias=p["bnd"][1])
        kw1 = dict(out_scale=p["bn2"][0], bias=p["bn2"][1], act=M.ACT_PRELU, slope=p["slope"])
        if xs is None:
            kw1.update(in_scale=p["bn1"][0], in_shift=p["bn1"][1])
        kw2 = dict(out_scale=p["bn3"][0], bias=p["bn3"][1], residual=identity)
        if nxt is None:
            return conv_pair(xs if xs is not None else x, p["w1"], kw1, p["w2"], self.stride, kw2), None
        n_bn = nxt._prepared()["bn1"]
        split, out = conv_pair(xs if xs is not None else x, p["w1"], kw1, p["w2"], self.stride, kw2,
                               out_split=dict(next_scale=n_bn[0], next_shift=n_bn[1]))
        return out, split


def run_block_chain(blocks, x, xs=None, after=None):
    """`for b in blocks: x = b(x)` with the unit -> unit hand-off: every block's second conv also writes the next block's
    first-conv input pre-split.  after: the block that follows the chain (its first conv receives the last hand-over).
    -> (x, hand-over for `after` | None)."""
    blocks = list(blocks)
    for i, blk in enumerate(blocks):
        nxt = blocks[i + 1] if i + 1 < len(blocks) else after
        oh, ow = (x.shape[2] - 1) // blk.stride + 1, (x.shape[3] - 1) // blk.stride + 1
        x, xs = blk.forward_chain(x, xs, nxt if (nxt is not None and nxt.takes_split(oh, ow)) else None)
    return x, xs


def _make_layer(inplanes, planes, blocks):
    down = nn.Sequential(nn.Conv2d(inplanes, planes, 1, 2, bias=False), nn.BatchNorm2d(planes, eps=1e-05))
    layers = [IBasicBlock(inplanes, planes, 2, down)]
    layers += [IBasicBlock(planes, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


class fs_encoder_v2(FrozenPlanMixin, nn.Module):  # feature_style_encoder.py:12-65
    def __init__(self, n_styles=18, opts=None, residual=False, use_coeff=False, resnet_layer=None,
                 video_input=False, f_maps=512, stride=(1, 1)):
        super().__init__()
        if video_input:
            raise NotImplementedError("video_input is not used by HairFast")
        # The reference seeds the trunk from an ArcFace iresnet50 checkpoint
        # (opts.arcface_model_path) and then overwrites it with the FS-encoder weights
        # (FSencoder.py:31-40); here the state dict is loaded directly.
        self.conv = nn.Sequential(nn.Conv2d(3, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64, eps=1e-05), nn.PReLU(64))
        inpl = 64
        for li, (planes, blocks) in enumerate(_IRESNET50):
            setattr(self, f"block_{li + 1}", _make_layer(inpl, planes, blocks))
            inpl = planes
        self.content_layer = nn.Sequential(
            nn.BatchNorm2d(256, eps=1e-05), nn.Conv2d(256, 512, 3, 1, 1, bias=False), nn.BatchNorm2d(512, eps=1e-05),
            nn.PReLU(num_parameters=512), nn.Conv2d(512, 512, 3, stride, 1, bias=False), nn.BatchNorm2d(512, eps=1e-05))
        self.content_stride = stride[0] if isinstance(stride, (tuple, list)) else stride
        self.avg_pool = nn.AdaptiveAvgPool2d((3, 3))
        self.styles = nn.ModuleList([nn.Linear(960 * 9, 512) for _ in range(n_styles)])
        self._plan = None

    def forward(self, x):
        """x [B,3,256,256] -> (latents [B,n_styles,512], content [B,512,16,16])."""
        require_gpu(x)
        L, st = lib(), stream()
        if self._plan is None:
            c, cl = self.conv, self.content_layer
            self._plan = {
                "w_in": prep_conv(c[0], pad=True), "bn_in": fold_bn(c[1]), "slope_in": c[2].weight.detach(),
                "c_bn0": fold_bn(cl[0]), "c_w1": prep_conv(cl[1]), "c_bn2": fold_bn(cl[2]),
                "c_slope": cl[3].weight.detach(), "c_w4": prep_conv(cl[4]), "c_bn5": fold_bn(cl[5]),
                # the 18 heads share their input: one [18*512, 8640] GEMV batch
                "head_w": torch.cat([m.weight.detach() for m in self.styles], 0).contiguous(),
                "head_b": torch.cat([m.bias.detach() for m in self.styles], 0).contiguous()}
        p = self._plan
        x = conv(x, p["w_in"], 3, 1, out_scale=p["bn_in"][0], bias=p["bn_in"][1], act=M.ACT_PRELU, slope=p["slope_in"])
        b = x.shape[0]
        pooled = x.new_empty((b, 960, 3, 3))
        c_off, content = 0, None
        xs = None
        for li in range(4):
            nxt_layer = getattr(self, f"block_{li + 2}", None) if li < 3 else None
            x, xs = run_block_chain(getattr(self, f"block_{li + 1}"), x, xs, after=None if nxt_layer is None else nxt_layer[0])
            if li == 2:
                content = conv_pair(x, p["c_w1"], dict(in_scale=p["c_bn0"][0], in_shift=p["c_bn0"][1], out_scale=p["c_bn2"][0],
                                                       bias=p["c_bn2"][1], act=M.ACT_PRELU, slope=p["c_slope"]),
                                    p["c_w4"], self.content_stride, dict(out_scale=p["c_bn5"][0], bias=p["c_bn5"][1]))
            M.adaptive_avgpool_into(L, st, pooled, x, c_off)
            c_off += x.shape[1]
        out = M.linear(L, st, pooled.reshape(b, -1), p["head_w"], p["head_b"], 1.0)
        return out.reshape(b, len(self.styles), 512), content


class FSEncoder(nn.Module):
    """The slice of `Trainer` (trainer.py) that HairFast's Embedding stage uses:
    `.test(img=normalised_1024_image, return_latent=True)`."""

    def __init__(self, n_styles=18, fs_stride=2, scale=2, generator=None, run_discarded_generator=False):
        super().__init__()
        self.enc = fs_encoder_v2(n_styles=n_styles, stride=(fs_stride, fs_stride))
        self.register_buffer("dlatent_avg", torch.zeros(n_styles, 512))
        self.scale = scale
        self.generator = generator
        # False (default): the reference's discarded generator forward (trainer.py:295) is skipped.
        # Besides 148.5 GFLOP per image it draws 17 noise maps from torch's device RNG, so with it
        # skipped the RNG stream DIVERGES from the reference's: a seeded run whose later stages use
        # randomize_noise / noise=None no longer reproduces the reference's random draws (outputs stay
        # correct in distribution; HairFast's own later generator calls pass explicit layer ranges
        # and the default noise=None, so only bit-reproduction of a seeded reference run is affected).
        # Set True to keep the RNG stream - and the cost - identical to the reference.
        self.run_discarded_generator = run_discarded_generator

    @torch.inference_mode()
    def test(self, w=None, img=None, noise=None, zero_noise_input=True, return_latent=False, training_mode=False):
        require_gpu(img)
        L, st = lib(), stream()
        x = img
        for _ in range(self.scale):  # downscale(x, 2, 'bilinear'), trainer.py:61-64
            x = M.downscale2x(L, st, x)
        w_recon, fea = self.enc(x)
        w_recon = M.add_bcast(L, st, w_recon, self.dlatent_avg)  # trainer.py:289
        x_recon = None
        if (self.run_discarded_generator or reference_rng_walk()) and self.generator is not None:
            x_recon, _ = self.generator([w_recon], input_is_latent=True)  # trainer.py:295 (discarded by the caller)
        output = [img[:, :3], x_recon]
        if return_latent:
            output += [w_recon, fea]
        return output
