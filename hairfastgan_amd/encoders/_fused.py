"""Shared plumbing of the encoder mirrors: parameter containers are plain torch.nn layers
(so that state-dict keys equal the reference's), the arithmetic goes through the C ABI
with inference-mode BatchNorm folded into per-channel affines once per checkpoint."""
import os

import torch
import torch.nn.functional as F
from torch import nn

from .. import _marshal as M
from .._runtime import conv_precision, lib, plan_batch, require_gpu, stream


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


class PreparedConv:
    """A Conv2d's weights in the layouts the kernels take: `wt` [k*k,cin,cout] (or [G,k*k,cin,cout]
    for a grouped launch) for the fp32-MFMA kernels and, derived on first use, the fp16 (hi, lo)
    split of it for the fp16 matrix-core kernel (csrc/convh_enc.hip)."""

    def __init__(self, wt, k):
        self.wt, self.k = wt, k
        self.cin, self.cout = wt.shape[-2], wt.shape[-1]
        self._f16 = None
        self._small = None
        self._patch = None
        self.cout_true, self.cin_true = self.cout, self.cin   # prep_conv(pad=True): the module's own channel counts
        self._padded = {}

    def pad_vec(self, v):
        """A per-output-channel epilogue vector (bias, BN scale, PReLU slope) extended with zeros to the padded cout; cached."""
        if v is None or self.cout_true == self.cout:
            return v
        key = (v.data_ptr(), None if v.is_inference() else v._version)  # inference tensors carry no version counter
        if key not in self._padded:
            self._padded[key] = torch.cat([v.detach().reshape(-1), v.new_zeros(self.cout - self.cout_true)]).contiguous()
        return self._padded[key]

    def small(self):
        """(hi, lo) of a 3x3 weight in the tap-GEMM layout of hf_modconv3x3_small_f16_f32 (the nine taps as rows of one GEMM)."""
        if self._small is None:
            self._small = M.split_weights_small(lib(), stream(), self.wt)
        return self._small

    def patch(self):
        """(hi, lo) of a 3x3 weight as ONE GEMM over patches (`patches`, tap-major): K = 9*cin ordered (tap, ci) - the
        prepared layout itself."""
        if self._patch is None:
            w4 = self.wt if self.wt.ndim == 4 else self.wt.unsqueeze(0)                       # [G, 9, cin, cout]
            g, taps, cin, cout = w4.shape
            w1 = w4.reshape(g, 1, taps * cin, cout).contiguous()
            self._patch = M.conv_split_weights_f16(lib(), stream(), w1 if self.wt.ndim == 4 else w1[0])
        return self._patch

    def f16(self):
        if self._f16 is None:
            self._f16 = M.conv_split_weights_f16(lib(), stream(), self.wt)
        return self._f16


def prep_conv(conv, pad=False):
    """pad: zero input planes / zero filters up to the multiples the fp16 matrix-core kernels take (cin % 16, cout % 64):
    a 3-channel input layer or a 3- / 19-channel output head then runs there instead of on the fp32 MFMA (2.6x the flops
    of a 19-filter 1x1 head, at 10x the rate).  conv() pads the input's channels and slices the output back; the
    per-channel epilogue vectors are padded by `PreparedConv.pad_vec`."""
    require_gpu(conv.weight)
    w = conv.weight.detach()
    cout, cin = w.shape[:2]
    if pad and conv_precision() != "f32":
        ci_pad, co_pad = -(-cin // 16) * 16, -(-cout // 64) * 64
        if ci_pad != cin:
            w = torch.cat([w, w.new_zeros(cout, ci_pad - cin, *w.shape[2:])], 1)
        if co_pad != cout:
            w = torch.cat([w, w.new_zeros(co_pad - cout, *w.shape[1:])], 0)
    pc = PreparedConv(M.conv_prepare(lib(), stream(), w.contiguous()), conv.kernel_size[0])
    pc.cout_true, pc.cin_true = cout, cin
    return pc


# Which convs hand the fp16-core kernel a PRE-SPLIT input (one hf_split_activation_f16 pass, then LDS-DMA
# staging) instead of converting while staging: "heads" = the e4e style-head levels, whose input feeds
# up to 88 (group, channel-tile) block columns; "all" = also every conv with >= 8 block columns per input tile
# (512+ output channels: the deep encoder stages, PostProcess's 512-1024 channel trunk); "none".  Tuning knob.
PRESPLIT = os.environ.get("HAIRFAST_ENC_PRESPLIT", "all")
# 1x1 convolutions / Linear layers on the fp16 matrix cores (csrc/gemm_h.hip) instead of the fp32-MFMA general kernel
USE_GEMM_H = os.environ.get("HAIRFAST_GEMM_H", "1") != "0"
# conv -> conv hand-off through the first conv's epilogue (conv_pair): tuning knob, "0" = fp32 tensor + split pass
USE_PAIR = os.environ.get("HAIRFAST_CONV_PAIR", "1") != "0"


def _small_plane_conv(x, w, mode, kw):
    """A dense 3x3 stride-1 conv on planes too small for the tiled fp16-core kernel (under 16 columns: the CtrlHair shape
    adaptor's 4^2 / 8^2 layers with 1024-2048 channels, 75-150 MB of weights each) as the tap GEMM of csrc/gemm_h.hip -
    weight streaming with (tap, channel tile, K split) spread over the chip; the fp32-MFMA kernel took 420-680 us per layer
    at batch 16.  None when the call has an epilogue the combine pass does not have."""
    act = kw.get("act", M.ACT_NONE)
    if (set(kw) - {"bias", "act", "alpha"}) or act not in (M.ACT_NONE, M.ACT_LRELU) or (act == M.ACT_LRELU and kw.get("bias") is None):
        return None
    alpha = float(kw.get("alpha", 0.0)) if act == M.ACT_LRELU else 1.0
    return M.modconv3x3_small(lib(), stream(), x, w.small(), 3 if mode == "f16x3" else 1, None, None, None, None, kw.get("bias"),
                              w.cout, alpha=alpha, scale=1.0)


def patches(x, k, stride, pad, tap_major=True):
    """im2col of x [..., cin, h, w] as k*k strided slices of the zero-padded tensor, concatenated along the channel axis:
    [..., k*k*cin, oh, ow] with K ordered (tap, ci) (tap_major) or (ci, tap) like F.unfold.  One pad and one cat kernel for
    the whole batch - F.unfold launches an im2col kernel PER SAMPLE (264 launches for one level of the e4e style heads)."""
    h, wd = x.shape[-2:]
    oh, ow = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
    xp = F.pad(x, (pad, pad, pad, pad))
    taps = [xp[..., ky:ky + stride * (oh - 1) + 1:stride, kx:kx + stride * (ow - 1) + 1:stride] for ky in range(k) for kx in range(k)]
    if tap_major:
        return torch.cat(taps, dim=-3)
    return torch.stack(taps, dim=-3).reshape(*x.shape[:-3], x.shape[-3] * k * k, oh, ow)


def _patch_gemm_conv(x, w, mode, stride, kw):
    """A 3x3 conv whose OUTPUT planes are at most 8x8 (the e4e style heads' stride-2 chains 16^2 -> 8^2 -> ... -> 1, eleven
    heads per grouped launch, 9.4 MB of weights per head and level) as a GEMM over unfolded patches on the fp16 matrix
    cores: K = cin*9 with exactly the needed flops, (group, channel tile, image pair) blocks over the chip.  The patches
    are a few MB (strided slices of the padded map, glue).  The fp32-MFMA kernel ran these levels at 0.3-1.3 ms per launch of a batched swap."""
    groups, shared = kw.get("groups", 1), kw.get("x_shared", True)
    b, cin, h, wd = x.shape[-4:]
    oh, ow = (h - 1) // stride + 1, (wd - 1) // stride + 1
    cols = patches(x, 3, stride, 1).reshape(*x.shape[:-3], 9 * cin, 1, oh * ow)              # K = (tap, ci)
    hi, lo = w.patch()
    rest = {k_: v for k_, v in kw.items() if k_ not in ("groups", "x_shared")}
    y = M.conv1x1_f16(lib(), stream(), cols, hi, lo, 3 if mode == "f16x3" else 1, w.cout, 1, groups=groups, x_shared=shared, **rest)
    return y.reshape(*y.shape[:-2], oh, ow)


def takes_f16_conv(w, h, wd, k, stride, **kw):
    """Does conv() run this call on the tiled fp16-core 3x3 kernel (hf_conv2d_f16_f32) - the one that accepts a pre-split input
    and can emit a split output?  (The same tests as conv()'s dispatch, in its order.)"""
    mode = conv_precision()
    if mode == "f32" or k != 3:
        return False
    if (USE_GEMM_H and h <= 16 and wd <= 16 and (h - 1) // stride < 8 and (wd - 1) // stride < 8 and w.cin % 32 == 0
            and w.cout % 64 == 0 and w.cin * w.cout >= 256 * 256 and stride == 2 and not ({"in_scale", "in_shift", "residual"} & set(kw))):
        return False  # patch GEMM
    return M.conv2d_f16_supported(w.cin, w.cout, h, wd, k, stride)


def conv(x, w, k, stride=1, presplit=False, split_out=None, **kw):
    """Conv2d + folded BN / activation / residual.  3x3 convs whose shape the fp16 matrix-core kernel
    takes run there in the process-wide operand mode (_runtime.conv_precision: f16x3 = fp32-class
    split operands, f16 = rounded operands); everything else, and mode f32, on the fp32 MFMA.
    presplit: this conv's input is shared by many block columns - convert it once (see PRESPLIT).
    x may be a SplitActivation when takes_f16_conv(...) holds for the call.
    split_out = {"next_scale", "next_shift", "want_f32"} (any subset): the result is wanted as the pre-split input of the
    next fp16-core conv; returns (SplitActivation | None, fp32 out | None) - the split form whenever the launch can emit it
    from its epilogue (hf_conv2d_f16_split_f32), else (None, out)."""
    if w.cout_true != w.cout or w.cin_true != w.cin:  # prep_conv(pad=True): zero planes in, zero filters sliced off
        if split_out is not None or kw.get("groups", 1) != 1 or kw.get("residual") is not None:
            raise ValueError("padded convs are plain single-group layers")
        if w.cin_true != w.cin:
            x = F.pad(x, (0, 0, 0, 0, 0, w.cin - w.cin_true))
            for k_ in ("in_scale", "in_shift"):
                if kw.get(k_) is not None:
                    raise ValueError("padded input channels carry no affine")
        kw = {k_: (w.pad_vec(v) if k_ in ("out_scale", "bias", "slope") else v) for k_, v in kw.items()}
        y = _conv_unpadded(x, w, k, stride, presplit, **kw)
        return y if w.cout_true == w.cout else y[:, :w.cout_true].contiguous()
    return _conv_unpadded(x, w, k, stride, presplit, split_out, **kw)


def _conv_unpadded(x, w, k, stride=1, presplit=False, split_out=None, **kw):
    mode = conv_precision()
    h, wd = x.shape[-2], x.shape[-1]
    if split_out is not None:
        b = x.shape[0]
        nterms = 3 if mode == "f16x3" else 1
        out_px = b * ((h - 1) // stride + 1) * ((wd - 1) // stride + 1)  # (bit-neutral: the real batch in every mode)
        to_split = (not isinstance(x, M.SplitActivation) and PRESPLIT == "all"
                    and (w.cout // 64 >= 8 or (w.cin >= 128 and out_px >= 12288)))
        if (kw.get("groups", 1) == 1 and takes_f16_conv(w, h, wd, k, stride, **kw)
                and M.conv2d_f16_split_supported(lib(), b, w.cin, w.cout, h, wd, stride, nterms,
                                                 pre=to_split or isinstance(x, M.SplitActivation))):
            hi, lo = w.f16()
            if to_split:
                x = M.split_activation_f16(lib(), stream(), x, kw.pop("in_scale", None), kw.pop("in_shift", None), want_lo=nterms == 3)
            return M.conv2d_f16_split(lib(), stream(), x, hi, lo, nterms, w.cout, stride, next_scale=split_out.get("next_scale"),
                                      next_shift=split_out.get("next_shift"), want_f32=split_out.get("want_f32", False), **kw)
        return None, _conv_unpadded(x, w, k, stride, presplit, **kw)
    if isinstance(x, M.SplitActivation):  # a hand-off from the producing conv: only the tiled fp16-core kernel reads it
        if k != 3 or mode == "f32":
            raise ValueError("a pre-split input goes to a 3x3 conv on the fp16 matrix cores (takes_f16_conv)")
        hi, lo = w.f16()
        return M.conv2d_f16(lib(), stream(), x, hi, lo, 3 if mode == "f16x3" else 1, w.cout, stride, **kw)
    if mode != "f32" and k == 1 and USE_GEMM_H and M.conv1x1_f16_supported(w.cin, w.cout):
        hi, lo = w.f16()
        nterms = 3 if mode == "f16x3" else 1
        # one input feeding >= 8 output-channel tiles over enough pixels: convert it once (hf_split_activation_f16), the
        # GEMM then stages it by LDS-DMA instead of converting it in every block column
        # (from 32 tiles already at 128 pixels: SEAN's table GEMM - 288 channel tiles over 304 label columns - spent 3/4 of its
        # 160 us converting the same 512 x 304 input in every block column)
        px = x.shape[0] * h * wd  # pre-split or register-staged input: equal results, so the whole launch decides in every mode
        if (PRESPLIT == "all" and stride == 1 and kw.get("groups", 1) == 1 and w.cin % 8 == 0
                and ((w.cout // 64 >= 8 and px >= 512) or (w.cout // 64 >= 32 and px >= 128))):
            x = M.split_activation_f16(lib(), stream(), x, kw.pop("in_scale", None), kw.pop("in_shift", None), want_lo=nterms == 3)
        return M.conv1x1_f16(lib(), stream(), x, hi, lo, nterms, w.cout, stride, **kw)
    if (mode != "f32" and k == 3 and USE_GEMM_H and h <= 16 and wd <= 16 and (h - 1) // stride < 8 and (wd - 1) // stride < 8
            and w.cin % 32 == 0 and w.cout % 64 == 0 and w.cin * w.cout >= 256 * 256 and stride == 2
            and not ({"in_scale", "in_shift", "residual"} & set(kw))):
        return _patch_gemm_conv(x, w, mode, stride, kw)
    if (mode != "f32" and k == 3 and stride == 1 and x.dim() == 4 and not M.conv2d_f16_supported(w.cin, w.cout, h, wd, k, stride)
            and M.conv3x3_small_supported(w.cin, w.cout, h, wd, x.shape[0])):
        y = _small_plane_conv(x, w, mode, kw)
        if y is not None:
            return y
    if mode != "f32" and M.conv2d_f16_supported(w.cin, w.cout, h, wd, k, stride):
        hi, lo = w.f16()
        nterms = 3 if mode == "f16x3" else 1
        groups = kw.get("groups", 1)
        # worth a separate pass when the same input tile is converted by many (group, 64-channel) block columns -
        # or, from 128 input channels, when a batched pass (HairFast.swap_batch) makes the extra launch negligible
        # (tools/bench_enc_layers.py, ENC_BATCH_MULT=8: 256->256 @32^2 128 -> 111 us, 512->512 stride 2 340 -> 160 us)
        out_px = x.shape[0] * ((h - 1) // stride + 1) * ((wd - 1) // stride + 1)  # (bit-neutral: the real batch)
        many = groups * (w.cout // 64) >= 8 or (w.cin >= 128 and out_px >= 12288)
        if not isinstance(x, M.SplitActivation) and ((PRESPLIT == "all" and many) or (presplit and PRESPLIT in ("all", "heads"))):
            x = M.split_activation_f16(lib(), stream(), x, kw.pop("in_scale", None), kw.pop("in_shift", None),
                                       want_lo=nterms == 3)
        return M.conv2d_f16(lib(), stream(), x, hi, lo, nterms, w.cout, stride, **kw)
    return M.conv2d(lib(), stream(), x, w.wt, k, stride, **kw)


# unit -> unit hand-off (round 5): the LAST conv / tail kernel of a residual unit also writes the next unit's first-conv input in
# its pre-split layout (with that unit's BatchNorm applied): no split pass at the unit boundary, and the next unit's first conv
# stages by LDS-DMA whatever its width.  "0" = every unit converts its own input.  Same bits either way.
USE_CHAIN = os.environ.get("HAIRFAST_UNIT_CHAIN", "1") != "0"


def conv_pair(x, w1, kw1, w2, stride2, kw2, stride1=1, out_split=None):
    """conv3x3(x, w1, stride1, **kw1) -> conv3x3(., w2, stride2, **kw2): the two convolutions of an IR-SE / IBasicBlock unit
    (helpers.py:99-115, iresnet.py:44-56; stride on the second) or of a ResNet BasicBlock (resnet.py:36-45; stride on the
    first).  When both run on the tiled fp16-core kernel the first one's epilogue writes its result straight in the second
    one's pre-split input layout (hf_conv2d_f16_split_f32): no fp32 tensor in between, no split pass, and the second conv
    stages by LDS-DMA.
    x may be a SplitActivation (kw1 then carries no input affine: it went into the split).
    out_split = {"next_scale", "next_shift"}: the second conv's result is ALSO wanted as the pre-split input of a following
    fp16-core conv -> returns (SplitActivation | None, fp32 result)."""
    h, wd = x.shape[-2], x.shape[-1]
    h1, w1d = (h - 1) // stride1 + 1, (wd - 1) // stride1 + 1
    if USE_PAIR and takes_f16_conv(w2, h1, w1d, 3, stride2, **kw2):
        mid, mid32 = conv(x, w1, 3, stride1, split_out={}, **kw1)
        mid = mid if mid is not None else mid32
    else:
        mid = conv(x, w1, 3, stride1, **kw1)
    if out_split is None:
        return conv(mid, w2, 3, stride2, **kw2)
    return conv(mid, w2, 3, stride2, split_out=dict(out_split, want_f32=True), **kw2)


def chain_takes_split(w1, h, wd, **kw1):
    """Will the first 3x3 conv of a unit accept a pre-split input handed over by its predecessor (the tiled fp16-core kernel
    runs it)?"""
    return USE_CHAIN and takes_f16_conv(w1, h, wd, 3, 1, **kw1)
