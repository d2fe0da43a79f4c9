"""BiSeNet face parsing + `get_segmentation` on the MI355X kernels - host-side
mirror of models/CtrlHair/external_code/face_parsing/model.py:230-253 (`BiSeNet`), resnet.py:19-77
(`Resnet18`, `BasicBlock`), my_parsing_util.py:72-95 (`FaceParsing_tensor`) and models/Net.py:108-115
(`get_segmentation`).

It is the direct consumer of the generator's pixels (Alignment.py:65-67 parses the rotated 1024^2 image,
Embedding.py:81 the three 512^2 inputs: 5 calls per triple) and the subject of north_star's "bit-exact
segmentation-mask indices": the mask is an argmax, so it is exactly reproducible wherever the top-1 / top-2
logit margin exceeds the fp32 tolerance of the logits; tests/test_gpu_parsing.py counts the index flips against
the reference-pinned oracle and shows they all sit below that margin.

Same parameter names as the reference (`cp.resnet.*`, `cp.arm16.*`, `ffm.*`, `conv_out.*`, ...).  Execution:
every Conv2d + BatchNorm2d (+ ReLU / residual) is one launch of the fused conv kernels (fp16 matrix cores where
the shape qualifies, encoders/_fused.conv); the attention gates, the nearest up-samplings and the max-pool are
small HIP kernels; the tail - bilinear up-sampling of the 19 logit planes, argmax, label permutation, nearest
resize - is ONE kernel that evaluates the interpolation only at the pixels the resized mask keeps.  The two
auxiliary heads (`conv_out16` / `conv_out32`) exist as parameters but are not evaluated: HairFast only uses the
first output.  Inference only.
"""
import torch
from torch import nn

from . import _marshal as M
from ._runtime import conv_precision, lib, require_gpu, stream
from .encoders._fused import FrozenPlanMixin, chain_takes_split, conv, conv_pair, fold_bn, prep_conv, takes_f16_conv
from .encoders import _fused

USE_CHAIN_STRIDED = True  # hand-over into a stride-2 first conv (pre-split stride-2 form)

# FaceParsing_tensor.label_list order -> index in PARSING_LABEL_LIST (global_value_utils.py:49-51); 13 = hair
_BISENET_LABELS = ["background", "skin_other", "l_brow", "r_brow", "l_eye", "r_eye", "eye_g", "l_ear", "r_ear", "ear_r",
                   "nose", "mouth", "u_lip", "l_lip", "neck", "neck_l", "cloth", "hair", "hat"]
_CELEBA_LABELS = ["background", "skin_other", "nose", "eye_g", "l_eye", "r_eye", "l_brow", "r_brow", "l_ear", "r_ear",
                  "mouth", "u_lip", "l_lip", "hair", "hat", "ear_r", "neck_l", "neck", "cloth"]
LABEL_REMAP = [_CELEBA_LABELS.index(n) for n in _BISENET_LABELS]

RELU = dict(act=M.ACT_LRELU, alpha=0.0)  # ReLU = leaky ReLU with slope 0 in the conv epilogue


class ConvBNReLU(nn.Module):  # model.py:13-36
    def __init__(self, in_chan, out_chan, ks=3, stride=1, padding=1):
        super().__init__()
        self.conv = nn.Conv2d(in_chan, out_chan, kernel_size=ks, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(out_chan)
        self.ks, self.stride = ks, stride

    def plan(self):
        return prep_conv(self.conv), fold_bn(self.bn)

    @staticmethod
    def run(p, x, ks, stride=1):
        w, (s, t) = p
        return conv(x, w, ks, stride, out_scale=s, bias=t, **RELU)


class BiSeNetOutput(nn.Module):  # model.py:38-48
    def __init__(self, in_chan, mid_chan, n_classes):
        super().__init__()
        self.conv = ConvBNReLU(in_chan, mid_chan, ks=3, stride=1, padding=1)
        self.conv_out = nn.Conv2d(mid_chan, n_classes, kernel_size=1, bias=False)


class AttentionRefinementModule(nn.Module):  # model.py:68-89
    def __init__(self, in_chan, out_chan):
        super().__init__()
        self.conv = ConvBNReLU(in_chan, out_chan, ks=3, stride=1, padding=1)
        self.conv_atten = nn.Conv2d(out_chan, out_chan, kernel_size=1, bias=False)
        self.bn_atten = nn.BatchNorm2d(out_chan)
        self.sigmoid_atten = nn.Sigmoid()


class BasicBlock(nn.Module):  # resnet.py:19-46
    def __init__(self, in_chan, out_chan, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_chan, out_chan, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(out_chan)
        self.conv2 = nn.Conv2d(out_chan, out_chan, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(out_chan)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if in_chan != out_chan or stride != 1:
            self.downsample = nn.Sequential(nn.Conv2d(in_chan, out_chan, 1, stride, bias=False), nn.BatchNorm2d(out_chan))
        self.stride = stride

    def plan(self):
        p = {"w1": prep_conv(self.conv1), "bn1": fold_bn(self.bn1), "w2": prep_conv(self.conv2), "bn2": fold_bn(self.bn2)}
        if self.downsample is not None:
            p["wd"], p["bnd"] = prep_conv(self.downsample[0]), fold_bn(self.downsample[1])
        return p

    def run(self, p, x):
        return self.run_chain(p, x, None, False)[0]

    def takes_split(self, p, h, wd):
        """Would this block's first conv accept its input pre-split (handed over by the previous block's second conv)?"""
        return chain_takes_split(p["w1"], h, wd, out_scale=p["bn1"][0], bias=p["bn1"][1], **RELU) if self.stride == 1 else \
            (USE_CHAIN_STRIDED and _fused.USE_CHAIN and takes_f16_conv(p["w1"], h, wd, 3, self.stride, out_scale=p["bn1"][0], bias=p["bn1"][1], **RELU))

    def run_chain(self, p, x, xs, hand_over):
        """xs: x in conv1's pre-split layout (the previous block's hand-over) or None; hand_over: also return the result
        pre-split for the next block's conv1 (no affine: a BasicBlock's conv1 takes its input as it is).  -> (out, split | None)."""
        sc = x if "wd" not in p else conv(x, p["wd"], 1, self.stride, out_scale=p["bnd"][0], bias=p["bnd"][1])
        # relu(shortcut + bn2(conv2(r))): the residual joins BEFORE the activation (resnet.py:41-45); conv1's result goes to
        # conv2 in its pre-split input layout when both run on the fp16 matrix cores (conv_pair)
        first = xs if xs is not None else x
        kw1 = dict(out_scale=p["bn1"][0], bias=p["bn1"][1], **RELU)
        kw2 = dict(out_scale=p["bn2"][0], bias=p["bn2"][1], act=M.ACT_LRELU | M.ACT_RESIDUAL_FIRST, alpha=0.0, residual=sc)
        if not hand_over:
            return conv_pair(first, p["w1"], kw1, p["w2"], 1, kw2, stride1=self.stride), None
        split, out = conv_pair(first, p["w1"], kw1, p["w2"], 1, kw2, stride1=self.stride, out_split={})
        return out, split


def _layer(in_chan, out_chan, stride):
    return nn.Sequential(BasicBlock(in_chan, out_chan, stride), BasicBlock(out_chan, out_chan, 1))


class Resnet18(nn.Module):  # resnet.py:56-77 (no download here: weights come with BiSeNet's checkpoint)
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1, self.layer2 = _layer(64, 64, 1), _layer(64, 128, 2)
        self.layer3, self.layer4 = _layer(128, 256, 2), _layer(256, 512, 2)


class ContextPath(nn.Module):  # model.py:92-127
    def __init__(self):
        super().__init__()
        self.resnet = Resnet18()
        self.arm16 = AttentionRefinementModule(256, 128)
        self.arm32 = AttentionRefinementModule(512, 128)
        self.conv_head32 = ConvBNReLU(128, 128, ks=3, stride=1, padding=1)
        self.conv_head16 = ConvBNReLU(128, 128, ks=3, stride=1, padding=1)
        self.conv_avg = ConvBNReLU(512, 128, ks=1, stride=1, padding=0)


class FeatureFusionModule(nn.Module):  # model.py:178-207
    def __init__(self, in_chan, out_chan):
        super().__init__()
        self.convblk = ConvBNReLU(in_chan, out_chan, ks=1, stride=1, padding=0)
        self.conv1 = nn.Conv2d(out_chan, out_chan // 4, kernel_size=1, bias=False)
        self.conv2 = nn.Conv2d(out_chan // 4, out_chan, kernel_size=1, bias=False)
        self.relu = nn.ReLU(inplace=True)
        self.sigmoid = nn.Sigmoid()


class BiSeNet(FrozenPlanMixin, nn.Module):  # model.py:230-253
    def __init__(self, n_classes=19):
        super().__init__()
        self.cp = ContextPath()
        self.ffm = FeatureFusionModule(256, 256)
        self.conv_out = BiSeNetOutput(256, 256, n_classes)
        self.conv_out16 = BiSeNetOutput(128, 64, n_classes)
        self.conv_out32 = BiSeNetOutput(128, 64, n_classes)
        self.n_classes = n_classes
        self._plan = None

    def _prepared(self):
        if self._plan is None:
            r, cp = self.cp.resnet, self.cp
            p = {"stem": (prep_conv(r.conv1), fold_bn(r.bn1)), "stem_f16": None}
            if r.conv1.out_channels % 64 == 0 and tuple(r.conv1.weight.shape[1:]) == (3, 7, 7):
                p["stem_f16"] = M.stem_prepare(r.conv1.weight)
            for li in (1, 2, 3, 4):
                for j in (0, 1):
                    p[f"l{li}.{j}"] = getattr(r, f"layer{li}")[j].plan()
            for name in ("arm16", "arm32"):
                a = getattr(cp, name)
                p[name] = (a.conv.plan(), prep_conv(a.conv_atten), fold_bn(a.bn_atten))
            for name in ("conv_head32", "conv_head16", "conv_avg"):
                p[name] = getattr(cp, name).plan()
            p["ffm"] = (self.ffm.convblk.plan(), prep_conv(self.ffm.conv1), prep_conv(self.ffm.conv2))
            p["out"] = (self.conv_out.conv.plan(), prep_conv(self.conv_out.conv_out, pad=True))  # 19 class filters padded to 64
            p["remap"] = torch.tensor(LABEL_REMAP, dtype=torch.int32, device=self.conv_out.conv_out.weight.device)
            self._plan = p
        return self._plan

    def logits_low(self, x):
        """The first head's class scores at 1/8 resolution [B, n_classes, H/8, W/8] (before model.py:249's up-sampling)."""
        require_gpu(x)
        L, st = lib(), stream()
        p = self._prepared()
        r = self.cp.resnet
        w, (s, t) = p["stem"]
        if p["stem_f16"] is not None and conv_precision() != "f32":
            # conv1 + bn1 + ReLU + the 3x3/2 max pool in one pass on the fp16 matrix cores (csrc/stem.hip)
            x = M.stem7x7s2(L, st, x, p["stem_f16"], out_scale=s, bias=t, alpha=0.0, pool=True)
        else:
            x = conv(x, w, 7, 2, out_scale=s, bias=t, **RELU)
            x = M.maxpool3x3s2(L, st, x)
        feats = []
        order = [(li, j) for li in (1, 2, 3, 4) for j in (0, 1)]
        xs = None
        for k, (li, j) in enumerate(order):  # block -> block hand-over of the pre-split conv1 input (encoders/_fused.py USE_CHAIN)
            blk = getattr(r, f"layer{li}")[j]
            nxt = order[k + 1] if k + 1 < len(order) else None
            oh, ow = (x.shape[2] - 1) // blk.stride + 1, (x.shape[3] - 1) // blk.stride + 1
            hand = nxt is not None and getattr(r, f"layer{nxt[0]}")[nxt[1]].takes_split(p[f"l{nxt[0]}.{nxt[1]}"], oh, ow)
            x, xs = blk.run_chain(p[f"l{li}.{j}"], x, xs, hand)
            if j == 1:
                feats.append(x)
        feat8, feat16, feat32 = feats[1], feats[2], feats[3]

        def arm(name, f):
            cbr, w_att, (s_att, t_att) = p[name]
            feat = ConvBNReLU.run(cbr, f, 3)
            pooled = M.plane_mean(L, st, feat)  # F.avg_pool2d over the whole plane
            logit = conv(pooled[:, :, None, None], w_att, 1, 1, out_scale=s_att, bias=t_att)  # bn_atten(conv_atten(.))
            return feat, logit.reshape(logit.shape[0], -1)

        b = x.shape[0]
        avg = M.plane_mean(L, st, feat32)[:, :, None, None]
        avg = ConvBNReLU.run(p["conv_avg"], avg, 1)                                   # [B,128,1,1]; nearest up = broadcast
        f32a, g32 = arm("arm32", feat32)
        feat32_sum = M.gate(L, st, f32a, g32, add_bcast=avg.reshape(b, -1))            # feat32_arm + avg_up
        feat32_up = M.upsample_nearest(L, st, feat32_sum, feat16.shape[2], feat16.shape[3])
        feat32_up = ConvBNReLU.run(p["conv_head32"], feat32_up, 3)
        f16a, g16 = arm("arm16", feat16)
        feat16_sum = M.gate(L, st, f16a, g16, add_plane=feat32_up)                     # feat16_arm + feat32_up
        feat16_up = M.upsample_nearest(L, st, feat16_sum, feat8.shape[2], feat8.shape[3])
        feat_cp8 = ConvBNReLU.run(p["conv_head16"], feat16_up, 3)
        # FeatureFusionModule(feat_res8, feat_cp8)
        blk, w1, w2 = p["ffm"]
        feat = ConvBNReLU.run(blk, torch.cat([feat8, feat_cp8], dim=1), 1)
        att = M.plane_mean(L, st, feat)[:, :, None, None]
        att = conv(conv(att, w1, 1, 1, **RELU), w2, 1, 1)
        fuse = M.gate(L, st, feat, att.reshape(b, -1), plus_one=1.0)                   # feat * sigmoid(atten) + feat
        cbr, w_out = p["out"]
        return conv(ConvBNReLU.run(cbr, fuse, 3), w_out, 1, 1)

    @torch.inference_mode()
    def parse(self, x, resize=True, remap=True):
        """[B,3,H,W] ImageNet-normalised -> int64 labels [B,1,256,256] (resize) or [B,1,H,W]: argmax of the bilinearly
        up-sampled class scores, CelebAMask label order when `remap`."""
        low = self.logits_low(x)
        H, W = x.shape[2], x.shape[3]
        return M.parsing_mask(lib(), stream(), low, self._prepared()["remap"] if remap else None, (H, W),
                              (256, 256) if resize else (H, W))

    def forward(self, x):
        """Reference return convention (feat_out, feat_out16, feat_out32) is not reproduced: HairFast only consumes the
        argmax of the first output - use parse() / get_segmentation(); this returns the 1/8-resolution class scores."""
        return self.logits_low(x)


def get_segmentation(net, img_rgb, resize=True):
    """models/Net.py:108-115 with the BiSeNet instance passed in (the reference keeps a class-level singleton):
    img_rgb [B,3,H,W] ImageNet-normalised -> long [B,1,256,256] (or [B,1,H,W]); the reference takes B = 1."""
    return net.parse(img_rgb, resize=resize, remap=True)
