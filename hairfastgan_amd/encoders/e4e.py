"""encoder4editing (e4e) encoder on the MI355X kernels - host-side mirror of
models/encoder4editing/models/encoders/psp_encoders.py:124-200 (`Encoder4Editing`),
helpers.py:57-120 (`SEModule`, `bottleneck_IR_SE`), psp_encoders.py:34-55
(`GradualStyleBlock`) and utils/model_utils.py:7-14 (`get_latents`).

Same module tree / parameter names as the reference (621 state-dict entries for
Encoder4Editing(50, 'ir_se')), so `pSp`'s `encoder.*` checkpoint keys load unchanged.
Execution: every Conv2d runs on the fp32-MFMA implicit-GEMM kernel with the neighbouring
BatchNorm2d (inference statistics), PReLU / LeakyReLU folded into its prologue /
epilogue; a bottleneck_IR_SE unit is 3 conv launches + squeeze-excite (mean, gate) + one
fused `res*gate + shortcut` pass instead of ~14 ATen calls.  Inference only.
"""
import math

import torch
from torch import nn

from .. import _marshal as M
from .._runtime import conv_precision, lib, require_gpu, stream
from ._fused import FrozenPlanMixin, PreparedConv, chain_takes_split, conv, conv_pair, fold_bn, prep_conv

# get_blocks(50): (in_channel, depth, stride) per unit (helpers.py:30-37)
_IR50 = ([(64, 64, 2)] + [(64, 64, 1)] * 2 + [(64, 128, 2)] + [(128, 128, 1)] * 3 +
         [(128, 256, 2)] + [(256, 256, 1)] * 13 + [(256, 512, 2)] + [(512, 512, 1)] * 2)
_IR_UNITS = {50: _IR50}


class SEModule(nn.Module):  # helpers.py:57-73 (parameter container; fused in bottleneck_IR_SE.forward)
    def __init__(self, channels, reduction):
        super().__init__()
        self.fc1 = nn.Conv2d(channels, channels // reduction, kernel_size=1, padding=0, bias=False)
        self.fc2 = nn.Conv2d(channels // reduction, channels, kernel_size=1, padding=0, bias=False)


class bottleneck_IR_SE(FrozenPlanMixin, nn.Module):  # helpers.py:93-120
    def __init__(self, in_channel, depth, stride):
        super().__init__()
        self.in_channel, self.depth, self.stride = in_channel, depth, stride
        if in_channel == depth:
            self.shortcut_layer = nn.MaxPool2d(1, stride)
        else:
            self.shortcut_layer = nn.Sequential(nn.Conv2d(in_channel, depth, (1, 1), stride, bias=False),
                                                nn.BatchNorm2d(depth))
        self.res_layer = nn.Sequential(
            nn.BatchNorm2d(in_channel), nn.Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False), nn.PReLU(depth),
            nn.Conv2d(depth, depth, (3, 3), stride, 1, bias=False), nn.BatchNorm2d(depth), SEModule(depth, 16))
        self._plan = None

    def _prepared(self):
        if self._plan is None:
            r = self.res_layer
            p = {"in": fold_bn(r[0]), "w1": prep_conv(r[1]), "slope": r[2].weight.detach(),
                 "w2": prep_conv(r[3]), "out": fold_bn(r[4]),
                 "fc1": r[5].fc1.weight.detach().reshape(self.depth // 16, self.depth),
                 "fc2": r[5].fc2.weight.detach().reshape(self.depth, self.depth // 16)}
            if self.in_channel != self.depth:
                p["wsc"] = prep_conv(self.shortcut_layer[0])
                p["sc"] = fold_bn(self.shortcut_layer[1])
            self._plan = p
        return self._plan

    def forward(self, x):
        return self.forward_chain(x, None, None)[0]

    def takes_split(self, h, wd):
        """Would this unit's first conv accept its input pre-split (handed over by the previous unit's tail)?"""
        p = self._prepared()
        return chain_takes_split(p["w1"], h, wd, act=M.ACT_PRELU, slope=p["slope"])

    def forward_chain(self, x, xs, nxt):
        """The unit on x [B,C,H,W]; xs: x already in the first conv's pre-split layout with THIS unit's BatchNorm applied
        (the previous unit's hand-over) or None; nxt: the next unit when its first conv takes such a hand-over, else None.
        -> (out, SplitActivation of the next unit's BatchNorm(out) | None).  Same bits as unit after unit."""
        require_gpu(x)
        p = self._prepared()
        if "wsc" in p:
            shortcut, sc_stride = conv(x, p["wsc"], 1, self.stride, out_scale=p["sc"][0], bias=p["sc"][1]), 1
        else:
            shortcut, sc_stride = x, self.stride  # MaxPool2d(1, stride) == strided identity
        kw1 = dict(act=M.ACT_PRELU, slope=p["slope"])
        if xs is None:
            kw1.update(in_scale=p["in"][0], in_shift=p["in"][1])
        r = conv_pair(xs if xs is not None else x, p["w1"], kw1, p["w2"], self.stride, dict(out_scale=p["out"][0], bias=p["out"][1]))
        gate = M.se_gate(lib(), stream(), M.plane_mean(lib(), stream(), r), p["fc1"], p["fc2"])
        if nxt is not None and r.shape[1] % 8 == 0:
            n_in = nxt._prepared()["in"]
            return M.scale_shortcut_add_split(lib(), stream(), r, gate, shortcut, sc_stride, n_in[0], n_in[1],
                                              want_lo=conv_precision() == "f16x3")
        return M.scale_shortcut_add(lib(), stream(), r, gate, shortcut, sc_stride), None


class EqualLinear(nn.Module):  # e4e's stylegan2 copy, model.py:128-157 (lr_mul = 1, no activation here)
    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, input):
        require_gpu(input)
        bias = None if self.bias is None else (self.bias.detach() if self.lr_mul == 1 else self.bias.detach() * self.lr_mul)
        return M.linear(lib(), stream(), input, self.weight.detach(), bias, self.scale)


class GradualStyleBlock(FrozenPlanMixin, nn.Module):  # psp_encoders.py:34-55
    def __init__(self, in_c, out_c, spatial):
        super().__init__()
        self.out_c, self.spatial = out_c, spatial
        mods = [nn.Conv2d(in_c, out_c, kernel_size=3, stride=2, padding=1), nn.LeakyReLU()]
        for _ in range(int(math.log2(spatial)) - 1):
            mods += [nn.Conv2d(out_c, out_c, kernel_size=3, stride=2, padding=1), nn.LeakyReLU()]
        self.convs = nn.Sequential(*mods)
        self.linear = EqualLinear(out_c, out_c, lr_mul=1)
        self._plan = None

    def forward(self, x):
        require_gpu(x)
        if self._plan is None:
            self._plan = [(prep_conv(m), m.bias.detach()) for m in self.convs if isinstance(m, nn.Conv2d)]
        for wt, b in self._plan:
            x = conv(x, wt, 3, 2, bias=b, act=M.ACT_LRELU, alpha=0.01)  # nn.LeakyReLU() default slope
        return self.linear(x.reshape(-1, self.out_c))


class Encoder4Editing(FrozenPlanMixin, nn.Module):  # psp_encoders.py:124-200
    def __init__(self, num_layers, mode="ir", opts=None):
        super().__init__()
        if num_layers not in _IR_UNITS or mode != "ir_se":
            raise NotImplementedError("HairFast uses Encoder4Editing(50, 'ir_se') (models/psp.py:25-27)")
        self.input_layer = nn.Sequential(nn.Conv2d(3, 64, (3, 3), 1, 1, bias=False), nn.BatchNorm2d(64), nn.PReLU(64))
        self.body = nn.Sequential(*[bottleneck_IR_SE(i, d, s) for (i, d, s) in _IR_UNITS[num_layers]])
        self.styles = nn.ModuleList()
        log_size = int(math.log(opts.stylegan_size, 2))
        self.style_count = 2 * log_size - 2
        self.coarse_ind, self.middle_ind = 3, 7
        for i in range(self.style_count):
            self.styles.append(GradualStyleBlock(512, 512, 16 if i < self.coarse_ind else (32 if i < self.middle_ind else 64)))
        self.latlayer1 = nn.Conv2d(256, 512, kernel_size=1, stride=1, padding=0)
        self.latlayer2 = nn.Conv2d(128, 512, kernel_size=1, stride=1, padding=0)
        self._plan = None

    def get_deltas_starting_dimensions(self):
        return list(range(self.style_count))

    def forward(self, x):
        """x [B,3,256,256] in [-1,1] -> W+ offsets [B,18,512] (inference stage: all deltas)."""
        require_gpu(x)
        if self._plan is None:
            il = self.input_layer
            self._plan = {"w_in": prep_conv(il[0], pad=True), "bn_in": fold_bn(il[1]), "slope_in": il[2].weight.detach(),
                          "lat1": prep_conv(self.latlayer1), "lat2": prep_conv(self.latlayer2)}
        p = self._plan
        x = conv(x, p["w_in"], 3, 1, out_scale=p["bn_in"][0], bias=p["bn_in"][1], act=M.ACT_PRELU, slope=p["slope_in"])
        taps = {}
        xs = None
        for i, unit in enumerate(self.body):
            nxt = self.body[i + 1] if i + 1 < len(self.body) else None
            oh, ow = (x.shape[2] - 1) // unit.stride + 1, (x.shape[3] - 1) // unit.stride + 1
            x, xs = unit.forward_chain(x, xs, nxt if (nxt is not None and nxt.takes_split(oh, ow)) else None)
            if i in (6, 20, 23):
                taps[i] = x
        c1, c2, c3 = taps[6], taps[20], taps[23]
        L, st = lib(), stream()
        p2 = M.upsample_bilinear_add(L, st, c3, conv(c2, p["lat1"], 1, 1, bias=self.latlayer1.bias.detach()))
        p1 = M.upsample_bilinear_add(L, st, p2, conv(c1, p["lat2"], 1, 1, bias=self.latlayer2.bias.detach()))
        # The 18 style heads in three families that share their input feature map (c3 / p2 / p1,
        # psp_encoders.py:187-199): each level of a family's conv chains is ONE grouped launch.
        deltas = []
        for (lo, hi), feats in (((0, self.coarse_ind), c3), ((self.coarse_ind, self.middle_ind), p2),
                                ((self.middle_ind, self.style_count), p1)):
            deltas += self._style_family(lo, hi, feats)
        w0 = deltas[0]
        rows = [w0] + [M.add_bcast(L, st, d, w0) for d in deltas[1:]]  # w[:, i] = w0 + delta_i
        return torch.stack(rows, dim=1)

    def _style_family(self, lo, hi, feats):
        """GradualStyleBlocks lo..hi-1 on the shared map `feats`: level-wise grouped convs, then the
        per-head EqualLinear.  Returns the list of [B,512] outputs."""
        L, st = lib(), stream()
        key = f"family_{lo}"
        if key not in self._plan:
            heads = [self.styles[i] for i in range(lo, hi)]
            convs = [[m for m in h.convs if isinstance(m, nn.Conv2d)] for h in heads]
            levels = []
            for lvl in range(len(convs[0])):
                wt = torch.stack([M.conv_prepare(L, st, c[lvl].weight.detach()) for c in convs]).contiguous()
                bias = torch.stack([c[lvl].bias.detach() for c in convs]).contiguous()
                levels.append((PreparedConv(wt, 3), bias))
            self._plan[key] = levels
        G = hi - lo
        x, shared = feats, True
        for wt, bias in self._plan[key]:
            x = conv(x, wt, 3, 2, presplit=True, bias=bias, act=M.ACT_LRELU, alpha=0.01, groups=G, x_shared=shared)
            shared = False
        return [self.styles[lo + g].linear(x[g].reshape(-1, 512)) for g in range(G)]


def get_latents(net, x):
    """model_utils.py:7-14: encoder(x) + latent_avg when `start_from_latent_avg`."""
    codes = net.encoder(x)
    if net.opts.start_from_latent_avg:
        avg = net.latent_avg
        if codes.ndim == 2:
            codes = M.add_bcast(lib(), stream(), codes, avg[0].contiguous())
        else:
            codes = M.add_bcast(lib(), stream(), codes, avg.contiguous())
    return codes
